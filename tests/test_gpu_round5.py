"""GPU tests (-m gpu) added in round 5 outside the linear adjoint (tests/test_gpu_linadj.py): the advisor's round-4 findings."""
import warnings

import pytest
import torch

pytestmark = pytest.mark.gpu


def dev():
    return torch.device('cuda:0')


class _Neg(torch.nn.Module):
    """f = -y: its backward saves no tensor - the saved-tensor heuristic of the device-controlled engine sees nothing."""

    def forward(self, t, y):
        return -y


class _Shift(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.b = torch.nn.Parameter(torch.full((3,), 0.25, dtype=torch.float64))

    def forward(self, t, y):
        return y + self.b


@pytest.mark.parametrize('make', [_Neg, _Shift])
def test_backward_dynamics_with_autograd_inside_are_never_recorded(make):
    """(advisor, round 4, medium) odeint_adjoint's backward solve runs on the device-controlled engine, whose default records an attempt as a
    hipGraph once enough attempts remain - with torch.autograd.grad inside, that aborts the process.  The augmented dynamics are marked
    `_mi_no_capture`; a long backward interval (dozens of attempts) completes eagerly and the gradients are those of the exact flow."""
    from tfdiffeq_amd import odeint, odeint_adjoint
    func = make().to(dev())
    y0 = torch.tensor([[1.0, -2.0, 0.5], [0.3, 0.1, -0.7]], dtype=torch.float64, device=dev(), requires_grad=True)
    t = torch.tensor([0.0, 6.0], dtype=torch.float64)
    out = odeint_adjoint(func, y0, t, rtol=1e-10, atol=1e-12, method='dopri5')
    out[-1].sum().backward()
    st = odeint.last_stats                                        # (the backward odeint call was the last one)
    assert st.get('n_attempts', 0) >= 14                          # more than EAGER_FIRST + MIN_REMAINING: 'auto' would have recorded
    assert 'hipGraph' not in str(st.get('engine', '')) and st.get('replays', 0) == 0
    if make is _Neg:
        assert torch.allclose(y0.grad, torch.full_like(y0, float(torch.exp(torch.tensor(-6.0, dtype=torch.float64)))), rtol=1e-7)
    else:
        e6 = float(torch.exp(torch.tensor(6.0, dtype=torch.float64)))
        assert torch.allclose(y0.grad, torch.full_like(y0, e6), rtol=1e-7)
        assert torch.allclose(func.b.grad, torch.full((3,), 2 * (e6 - 1.0), dtype=torch.float64, device=dev()), rtol=1e-6)


def test_mlp_fixed_grid_with_stage_fusion_falls_through_to_the_step_loop():
    """(advisor, round 4, low) rhs.MLP has a one-launch fixed-grid kernel but no per-stage kernels: options={'fusion': 'stage'} used to raise
    NativeError from mi_ode_create; it takes the per-step loop again, same values as the one-launch kernel."""
    from tfdiffeq_amd import models, odeint
    torch.manual_seed(0)
    f = models.ODEFunc(6, 16, non_linearity='tanh').to(dev())
    r = f.device_rhs()
    y0 = torch.randn(40, 6, device=dev())
    t = torch.linspace(0.0, 1.0, 5)
    with torch.no_grad():
        a = odeint(r, y0, t, method='rk4')
        b = odeint(r, y0, t, method='rk4', options={'fusion': 'stage'})
    assert float((a - b).abs().max()) <= 2e-6 * max(1.0, float(a.abs().max()))


def test_plain_callable_gets_gradients_for_exactly_the_leaves_it_uses():
    """(advisor, round 4, medium) `odeint(lambda t, y: net(y) ...)` under grad mode: the tensors that receive gradients are the leaves of the
    callable's autograd graph - a module it merely names gets None (not zeros: an optimizer with weight decay would move it), a tensor behind
    an attribute chain is found."""
    from tfdiffeq_amd import odeint
    torch.manual_seed(1)
    used = torch.nn.Linear(4, 4).double().to(dev())
    unused = torch.nn.Linear(4, 4).double().to(dev())

    class Box(object):
        pass
    box = Box()
    box.inner = Box()
    box.inner.scale = torch.tensor(0.5, dtype=torch.float64, device=dev(), requires_grad=True)

    def f(t, y):
        _ = unused
        return torch.tanh(used(y)) * box.inner.scale
    y0 = torch.randn(16, 4, dtype=torch.float64, device=dev())
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        out = odeint(f, y0, torch.tensor([0.0, 1.0], dtype=torch.float64), rtol=1e-8, atol=1e-10, method='dopri5')
    out[-1].pow(2).sum().backward()
    assert used.weight.grad is not None and used.bias.grad is not None and box.inner.scale.grad is not None
    assert unused.weight.grad is None and unused.bias.grad is None
    # against autograd through a fine fixed-step integration of the same function (float64, 400 rk4 steps)
    for p in (used.weight, used.bias, box.inner.scale):
        p.grad_ref, p.grad = p.grad.clone(), None
    y = y0.clone()
    h = 1.0 / 400
    for _ in range(400):
        k1 = f(0, y); k2 = f(0, y + 0.5 * h * k1); k3 = f(0, y + 0.5 * h * k2); k4 = f(0, y + h * k3)
        y = y + (h / 6.0) * (k1 + 2 * k2 + 2 * k3 + k4)
    y.pow(2).sum().backward()
    for p in (used.weight, used.bias, box.inner.scale):
        assert float((p.grad - p.grad_ref).abs().max()) <= 1e-6 * max(1.0, float(p.grad.abs().max()))
