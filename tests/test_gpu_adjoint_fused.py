"""GPU tests of the fused backward segment of odeint_adjoint (csrc/mi_ode_adjoint.h, include/mi_ode.h section A').

Reference behaviour: tfdiffeq/adjoint.py:57-178 - the backward pass integrates the augmented tuple state
(y, adj_y, adj_t, adj_params) with `odeint` (dopri5 over a heterogeneous tuple: per-component error ratios, their max,
initial step over all components).  Checkers:
  * the augmented dynamics (adjoint.py:69-105) against torch.autograd.grad of the same network,
  * whole segments and whole gradients against the SAME algorithm run on the plane-kernel engine (the generic tuple path,
    which the round-1/2 tests pin against the oracle restatement) - fp32, so the band is a few ulp of the largest element
    times the number of steps,
  * gradients against autograd through the torch-CPU restatement of the reference's Dopri5 path (oracle/ode_torch_cpu.py).
"""
import copy

import numpy as np
import pytest
import torch

from tests.bands import assert_scalar

pytestmark = pytest.mark.gpu


def dev():
    return torch.device('cuda:0')


def _canon(func, tensors):
    """parameters()-ordered tensors ([out, in] weights) -> the kernel's canonical flat order ([in, out] weights)."""
    w1, b1, w2, b2, w3, b3 = tensors
    return torch.cat([w1.t().reshape(-1), b1, w2.t().reshape(-1), b2, w3.t().reshape(-1), b3])


def _func(dim, hidden, seed, non_linearity='tanh', time_dependent=False):
    from tfdiffeq_amd.models import ODEFunc
    torch.manual_seed(seed)
    return ODEFunc(dim, hidden, non_linearity=non_linearity, time_dependent=time_dependent).to(dev())


def _engine(batch, dim, hidden, tol=1e-3, max_num_steps=1000, time_dependent=False):
    from tfdiffeq_amd import adjoint as ADJ
    f32 = lambda v: float(np.float32(v))     # noqa: E731
    return ADJ._FusedAdjointEngine(batch, dim, hidden, tol, tol, f32(0.9), f32(10.0), f32(0.2), max_num_steps, dev(), time_dependent)


def _rel(a, b):
    return float((a - b).abs().max()) / max(float(b.abs().max()), 1e-30)


@pytest.mark.parametrize('batch,dim,hidden', [(1, 3, 5), (8, 4, 16), (33, 16, 16), (100, 10, 100), (257, 64, 16), (300, 64, 128), (5000, 48, 96)])
def test_augmented_dynamics_match_autograd(batch, dim, hidden):
    """mi_ode_adjoint_dynamics = (f, -a^T df/dy, -a^T df/dtheta) of adjoint.py:69-105; ragged last tiles, padded widths."""
    func = _func(dim, hidden, 1)
    g = torch.Generator(device='cpu').manual_seed(2)
    y = torch.randn(batch, dim, generator=g).to(dev())
    a = torch.randn(batch, dim, generator=g).to(dev())
    eng = _engine(batch, dim, hidden)
    try:
        f, vy, vp = eng.dynamics(func.device_rhs(), y, a)
    finally:
        eng.close()
    yr = y.clone().requires_grad_(True)
    fr = func(torch.tensor(0.), yr)
    grads = torch.autograd.grad(fr, (yr,) + tuple(func.parameters()), -a)
    assert _rel(f, fr.detach()) < 3e-6
    assert _rel(vy, grads[0]) < 3e-6
    assert _rel(vp, _canon(func, grads[1:])) < 1e-5          # sums over the batch in a different order than rocBLAS


def _plane_segment(func, y, a, adj_t, theta, t0, t1, tol, max_num_steps=1000):
    """The reference's call (adjoint.py:148-153) on the generic tuple engine, adj_params in canonical order."""
    import tfdiffeq_amd as T
    fp = tuple(func.parameters())

    def aug(tt, ya):
        with torch.enable_grad():
            t_ = tt.detach().requires_grad_(True)
            y_ = ya[0].detach().requires_grad_(True)
            fe = func(t_, y_)
            vj = torch.autograd.grad(fe, (t_, y_) + fp, -ya[1], allow_unused=True)
        vt = torch.zeros_like(ya[2]) if vj[0] is None else vj[0].to(ya[2].dtype).reshape(ya[2].shape)     # (None: time-independent network)
        return (fe.detach(), vj[1], vt, _canon(func, vj[2:]))
    with torch.no_grad():
        out = T.odeint(aug, (y, a, adj_t, theta), torch.tensor([t0, t1], dtype=torch.float64), rtol=tol, atol=tol, method='dopri5',
                       options={'max_num_steps': max_num_steps})
    return out, dict(T.odeint.last_stats)


@pytest.mark.parametrize('batch,dim,hidden,tol,t0,t1', [
    (8, 4, 16, 1e-3, 1.0, 0.0), (100, 10, 16, 1e-4, 1.0, 0.0), (300, 64, 128, 1e-3, 1.0, 0.0), (64, 8, 32, 1e-6, 1.0, 0.25),
    (64, 8, 32, 1e-4, 0.0, 2.0),           # increasing time: no reversal (misc.py:311-321 leaves f alone)
    (2000, 33, 70, 1e-5, 0.7, -0.4)])
def test_segment_matches_the_plane_kernel_engine(batch, dim, hidden, tol, t0, t1):
    """One backward interval: same attempts, same accepted steps, end values within fp32 roundoff of the generic tuple path."""
    func = _func(dim, hidden, 3)
    g = torch.Generator(device='cpu').manual_seed(4)
    y = torch.randn(batch, dim, generator=g).to(dev())
    a = (torch.randn(batch, dim, generator=g) / batch).to(dev())
    eng = _engine(batch, dim, hidden, tol)
    theta = (0.01 * torch.randn(eng.n_params, generator=g)).to(dev())
    adj_t = torch.tensor(0.3, device=dev())
    try:
        a1, t_1, p1 = eng.segment(func.device_rhs(), y, a, adj_t, theta, t0, t1)
        st = eng.stats.as_dict()
    finally:
        eng.close()
    ref, rs = _plane_segment(func, y, a, adj_t, theta, t0, t1, tol)
    assert st['status'] == 0 and st['n_launches'] == 1
    assert (st['n_attempts'], st['n_accepted']) == (rs['n_attempts'], rs['n_accepted'])
    nsteps = max(st['n_accepted'], 1)
    assert _rel(a1, ref[1][1]) < 4e-6 * nsteps
    assert _rel(p1, ref[3][1]) < 4e-6 * nsteps
    assert abs(float(t_1) - float(ref[2][1])) <= 1e-6 * abs(float(adj_t))


def test_zero_adjoint_time_and_zero_gradient_segment():
    """adj_t = 0 makes d0/d1 of that component 0/0 in misc.py:233 (python max() then skips the NaN); adj_y = 0 makes every
    derivative of the adjoint components vanish.  Same path as the generic engine in both cases."""
    func = _func(6, 16, 5)
    y = torch.randn(50, 6, generator=torch.Generator().manual_seed(6)).to(dev())
    for a_scale, at in ((1.0, 0.0), (0.0, 0.0), (0.0, 0.5)):
        a = (a_scale * torch.randn(50, 6, generator=torch.Generator().manual_seed(7)) / 50).to(dev())
        eng = _engine(50, 6, 16, 1e-4)
        theta = torch.zeros(eng.n_params, device=dev())
        adj_t = torch.tensor(at, device=dev())
        try:
            a1, t_1, p1 = eng.segment(func.device_rhs(), y, a, adj_t, theta, 1.0, 0.0)
            st = eng.stats.as_dict()
        finally:
            eng.close()
        ref, rs = _plane_segment(func, y, a, adj_t, theta, 1.0, 0.0, 1e-4)
        assert (st['n_attempts'], st['n_accepted']) == (rs['n_attempts'], rs['n_accepted']), (a_scale, at, st, rs)
        assert float((a1 - ref[1][1]).abs().max()) <= 4e-6 * max(float(ref[1][1].abs().max()), 1e-30)
        assert float((p1 - ref[3][1]).abs().max()) <= 4e-6 * max(float(ref[3][1].abs().max()), 1e-30)
        assert float(t_1) == float(ref[2][1])


def test_max_num_steps_raises_like_the_reference():
    func = _func(4, 16, 8)
    y = torch.randn(20, 4).to(dev())
    a = torch.randn(20, 4).to(dev())
    eng = _engine(20, 4, 16, 1e-9, max_num_steps=2)
    try:
        with pytest.raises(AssertionError, match='max_num_steps exceeded'):
            eng.segment(func.device_rhs(), y, a, torch.tensor(0.1, device=dev()), torch.zeros(eng.n_params, device=dev()), 1.0, 0.0)
    finally:
        eng.close()


def _grads(block, x, fused, w=None, eval_times=None, fused_forward=False):
    """(output, dL/dx, dL/dparams, backward stats).  fused_forward = False: both runs share the plane-kernel forward pass, so
    the comparison isolates the backward solve."""
    from tfdiffeq_amd import adjoint as ADJ
    ADJ.FUSED, ADJ.FUSED_FORWARD = fused, fused_forward
    try:
        for p in block.parameters():
            p.grad = None
        xi = x.clone().requires_grad_(True)
        out = block(xi) if eval_times is None else block(xi, eval_times=eval_times)
        loss = out.pow(2).sum() if w is None else (out * w).sum()
        loss.backward()
        stats = dict(ADJ.odeint_adjoint.last_backward_stats)
        return out.detach(), xi.grad.clone(), [p.grad.clone() for p in block.odefunc.parameters()], stats
    finally:
        ADJ.FUSED, ADJ.FUSED_FORWARD = True, True


@pytest.mark.parametrize('batch,dim,hidden,tol', [(64, 8, 32, 1e-3), (1000, 16, 64, 1e-4), (4096, 64, 128, 1e-3)])
def test_odeblock_gradients_fused_against_plane_kernel_adjoint(batch, dim, hidden, tol):
    """`odeint_adjoint` under ODEBlock without per-op launches, gradients against the plane-engine adjoint."""
    from tfdiffeq_amd import models
    torch.manual_seed(9)
    block = models.ODEBlock(models.ODEFunc(dim, hidden, non_linearity='tanh'), tol=tol, adjoint=True).to(dev())
    x = torch.randn(batch, dim, generator=torch.Generator().manual_seed(10)).to(dev())
    out_f, gx_f, gp_f, st_f = _grads(block, x, True)
    out_p, gx_p, gp_p, st_p = _grads(block, x, False)
    assert st_f['engine'].startswith('fused adjoint kernel') and st_p['engine'] == 'plane kernels'
    assert all(s['n_launches'] == 1 and s['status'] == 0 for s in st_f['segments'])
    assert torch.equal(out_f, out_p)
    # Band: the solver tolerance, not fp32 roundoff.  The error estimate of the adj_params component is a difference of
    # O(|grad|) terms; the generic path rounds every stage's full-batch gradient to fp32 before combining them (as the reference
    # does), the fused kernel combines inside the accumulation - when that estimate sits at roundoff level the two controllers
    # pick different (both valid) step sizes, and the results then differ like two solves of the same tolerance do.
    band = max(2e-5, tol)
    assert _rel(gx_f, gx_p) < band
    for a, b in zip(gp_f, gp_p):
        assert _rel(a, b) < band


def test_several_output_times_and_time_gradients():
    """T = 4 output times: three backward intervals, adj_y jumps by grad_output[i-1] between them (adjoint.py:160), the time
    gradients collect dL/dt_i (adjoint.py:134-140, 162-166)."""
    from tfdiffeq_amd import odeint_adjoint
    from tfdiffeq_amd import adjoint as ADJ
    func = _func(5, 24, 11)
    y0 = torch.randn(70, 5, generator=torch.Generator().manual_seed(12)).to(dev())
    w = torch.randn(4, 70, 5, generator=torch.Generator().manual_seed(13)).to(dev())
    res = {}
    for fused in (True, False):
        ADJ.FUSED, ADJ.FUSED_FORWARD = fused, False
        try:
            for p in func.parameters():
                p.grad = None
            yi = y0.clone().requires_grad_(True)
            t = torch.tensor([0.0, 0.3, 0.8, 1.5], requires_grad=True)
            sol = odeint_adjoint(func, yi, t, rtol=1e-5, atol=1e-7, method='dopri5')
            (sol * w).sum().backward()
            res[fused] = (sol.detach(), yi.grad.clone(), t.grad.clone(), [p.grad.clone() for p in func.parameters()],
                          dict(odeint_adjoint.last_backward_stats))
        finally:
            ADJ.FUSED, ADJ.FUSED_FORWARD = True, True
    assert res[True][4]['engine'].startswith('fused') and len(res[True][4]['segments']) == 3
    assert torch.equal(res[True][0], res[False][0])
    assert _rel(res[True][1], res[False][1]) < 2e-5
    assert _rel(res[True][2], res[False][2]) < 2e-5
    for a, b in zip(res[True][3], res[False][3]):
        assert _rel(a, b) < 2e-5


def test_fused_adjoint_gradients_against_autograd_through_the_restatement():
    """The independent checker: autograd through oracle/ode_torch_cpu.odeint_dopri5 (op-for-op restatement of the reference's
    Dopri5 path, pinned by the golden fixtures) in float64 with tight tolerances = the exact gradient of the exact flow up to
    1e-8; the fused fp32 adjoint at rtol = atol = 1e-6 must agree with it to the adjoint's own accuracy."""
    from tfdiffeq_amd import odeint_adjoint
    from oracle import ode_torch_cpu as TC
    func = _func(6, 24, 14)
    cpu64 = copy.deepcopy(func).cpu().double()
    y0 = torch.randn(40, 6, generator=torch.Generator().manual_seed(15))
    w = torch.randn(40, 6, generator=torch.Generator().manual_seed(16))
    y64 = y0.double().requires_grad_(True)
    sol64, _ = TC.odeint_dopri5(lambda t_, y_: cpu64(t_, y_), y64, [0.0, 1.0], rtol=1e-9, atol=1e-11)
    (sol64[1] * w.double()).sum().backward()
    yi = y0.to(dev()).requires_grad_(True)
    sol = odeint_adjoint(func, yi, torch.tensor([0.0, 1.0]), rtol=1e-6, atol=1e-6, method='dopri5')
    (sol[1] * w.to(dev())).sum().backward()
    assert odeint_adjoint.last_backward_stats['engine'].startswith('fused')
    assert_scalar(_rel(yi.grad.cpu().double(), y64.grad), 'fused_adjoint_vs_fp64_autograd/tol1e-6/dL_dy0')
    for i, (pg, pc) in enumerate(zip(func.parameters(), cpu64.parameters())):
        assert_scalar(_rel(pg.grad.cpu().double(), pc.grad), 'fused_adjoint_vs_fp64_autograd/tol1e-6/dL_dparam%d' % i)


def test_cases_the_fused_kernel_does_not_cover_stay_on_the_plane_engine():
    from tfdiffeq_amd import models, odeint_adjoint
    torch.manual_seed(17)
    x = torch.randn(32, 8).to(dev())
    # frozen parameter: adj_params no longer is the full parameter vector
    blk = models.ODEBlock(models.ODEFunc(8, 16, non_linearity='tanh'), adjoint=True).to(dev())
    blk.odefunc.fc2.bias.requires_grad_(False)
    blk(x.clone().requires_grad_(True)).sum().backward()
    assert odeint_adjoint.last_backward_stats['engine'] == 'plane kernels'
    # a non-linearity the kernels do not know, float64 state
    for kw, dt in ((dict(non_linearity='ELU'), torch.float32), (dict(non_linearity='tanh'), torch.float64)):
        blk = models.ODEBlock(models.ODEFunc(8, 16, **kw), adjoint=True).to(dev()).to(dt)
        blk(x.to(dt).clone().requires_grad_(True)).sum().backward()
        assert odeint_adjoint.last_backward_stats['engine'] == 'plane kernels', kw
        assert all(p.grad is not None for p in blk.odefunc.parameters())


def test_config5_size_backward_is_one_launch_per_interval():
    """BASELINE config 5's shape (batch 32768, 64-128-128-64, fp32) through ODEBlock(adjoint=True): the backward pass is one
    kernel launch, its gradients agree with the plane-kernel adjoint."""
    from tfdiffeq_amd import models
    torch.manual_seed(18)
    block = models.ODEBlock(models.ODEFunc(64, 128, non_linearity='tanh'), tol=1e-3, adjoint=True).to(dev())
    x = torch.randn(32768, 64, generator=torch.Generator().manual_seed(19)).to(dev())
    from tfdiffeq_amd import odeint
    out_f, gx_f, gp_f, st_f = _grads(block, x, True, fused_forward=True)        # the training step as a user gets it
    fwd = dict(odeint.last_stats)
    out_p, gx_p, gp_p, st_p = _grads(block, x, False)
    assert fwd.get('n_launches') == 1, fwd                                   # forward: the whole-call MLP kernel
    assert len(st_f['segments']) == 1 and st_f['segments'][0]['n_launches'] == 1 and st_f['segments'][0]['status'] == 0
    assert _rel(out_f, out_p) < 1e-5                                         # (fused forward: its own tanh, mi_ode_mlp.h)
    assert _rel(gx_f, gx_p) < 1e-4
    for a, b in zip(gp_f, gp_p):
        assert _rel(a, b) < 1e-4


def test_hand_off_time_out_falls_back_to_the_generic_path(monkeypatch):
    """A grid hand-off that times out (the GPU shared with another persistent kernel; forced here by a spin limit of one poll)
    must not fail the training step: nothing has been committed, the backward pass is redone on the plane-kernel engine."""
    from tfdiffeq_amd import adjoint as ADJ
    from tfdiffeq_amd import models, odeint_adjoint
    torch.manual_seed(20)
    block = models.ODEBlock(models.ODEFunc(8, 16, non_linearity='tanh'), tol=1e-3, adjoint=True).to(dev())
    x = torch.randn(4096, 8, generator=torch.Generator().manual_seed(21)).to(dev())          # 128 workgroups
    ref = _grads(block, x, False)
    monkeypatch.setenv('MI_ODE_PERSIST_SPIN_FIRST', '1')
    monkeypatch.setenv('MI_ODE_PERSIST_SPIN_LIMIT', '1')
    ADJ.clear_adjoint_engines()
    try:
        with pytest.warns(UserWarning, match='fused adjoint kernel unavailable'):
            got = _grads(block, x, True)
    finally:
        ADJ.clear_adjoint_engines()
    assert got[3]['engine'] == 'plane kernels'
    assert torch.equal(got[1], ref[1]) and all(torch.equal(a, b) for a, b in zip(got[2], ref[2]))


# ---------------------------------------------------------------------------------------------
# the reference's other non-linearities: relu (ODEFunc's default, dense_odenet.py:14) and softplus
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize('act', ['relu', 'softplus'])
def test_other_activations_dynamics_and_segment(act):
    func = _func(10, 48, 30, act)
    g = torch.Generator(device='cpu').manual_seed(31)
    y = torch.randn(500, 10, generator=g).to(dev())
    a = (torch.randn(500, 10, generator=g) / 500).to(dev())
    eng = _engine(500, 10, 48, 1e-4)
    try:
        f, vy, vp = eng.dynamics(func.device_rhs(), y, a)
        yr = y.clone().requires_grad_(True)
        fr = func(torch.tensor(0.), yr)
        grads = torch.autograd.grad(fr, (yr,) + tuple(func.parameters()), -a)
        assert _rel(f, fr.detach()) < 3e-6 and _rel(vy, grads[0]) < 3e-6 and _rel(vp, _canon(func, grads[1:])) < 1e-5
        theta = torch.zeros(eng.n_params, device=dev())
        adj_t = torch.tensor(0.1, device=dev())
        a1, t_1, p1 = eng.segment(func.device_rhs(), y, a, adj_t, theta, 1.0, 0.0)
        st = eng.stats.as_dict()
    finally:
        eng.close()
    ref, rs = _plane_segment(func, y, a, adj_t, theta, 1.0, 0.0, 1e-4)
    assert (st['n_attempts'], st['n_accepted']) == (rs['n_attempts'], rs['n_accepted'])
    assert _rel(a1, ref[1][1]) < 2e-5 and _rel(p1, ref[3][1]) < 2e-5


@pytest.mark.parametrize('act', ['relu', 'softplus'])
def test_default_odenet_trains_on_the_fused_kernels(act):
    """ODENet with the reference's default non-linearity: forward on the whole-call MLP kernel (against the torch-CPU
    restatement), backward on the fused adjoint kernel (against float64 autograd through the restatement)."""
    from tfdiffeq_amd import models, odeint, odeint_adjoint
    from oracle import ode_torch_cpu as TC
    torch.manual_seed(32)
    block = models.ODEBlock(models.ODEFunc(12, 40, non_linearity=act), tol=1e-5, adjoint=True).to(dev())
    x = torch.randn(300, 12, generator=torch.Generator().manual_seed(33))
    with torch.no_grad():
        got = block(x.to(dev()))
    assert odeint.last_stats.get('n_launches') == 1
    cpu = copy.deepcopy(block.odefunc).cpu()
    ref, _ = TC.odeint_dopri5(lambda t_, y_: cpu(t_, y_), x, [0., 1.], rtol=1e-5, atol=1e-5)
    assert_scalar(_rel(got.cpu(), ref[1].detach()), 'default_odenet_forward/%s' % act)
    out = block(x.to(dev()).requires_grad_(True))
    out.pow(2).sum().backward()
    assert odeint_adjoint.last_backward_stats['engine'].startswith('fused')
    cpu64 = copy.deepcopy(cpu).double()
    ref64, _ = TC.odeint_dopri5(lambda t_, y_: cpu64(t_, y_), x.double(), [0., 1.], rtol=1e-9, atol=1e-11)
    ref64[1].pow(2).sum().backward()
    # (relu: the kinks cap what an adaptive solve of tolerance 1e-5 delivers, on either path - its recorded bands are wider)
    for i, (pg, pc) in enumerate(zip(block.odefunc.parameters(), cpu64.parameters())):
        # a-priori ceilings (tests/bands.py): relu - a sample crossing a kink between the float32 and the float64 solve changes a whole
        # column of the weight gradient (1e-2); softplus - smooth, the default tol 1e-3 of ODEBlock bounds the two solves' distance (3e-3)
        assert_scalar(_rel(pg.grad.cpu().double(), pc.grad), 'default_odenet_grads/%s/param%d' % (act, i),
                      ceiling=1e-2 if act == 'relu' else 3e-3)


# ---------------------------------------------------------------------------------------------
# the time-dependent network (dense_odenet.py:79-84: fc1 sees concat([t, x])) - round 4.  adj_t has a real derivative
# (-a^T df/dt = dot(w_t, -a^T df/db1)), theta starts with w_t, the stage time shifts the first layer's bias.
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize('batch,dim,hidden,t', [(8, 4, 16, 0.0), (100, 10, 100, 0.7), (300, 64, 128, -1.3), (5000, 48, 96, 2.5)])
def test_time_dependent_dynamics_match_autograd(batch, dim, hidden, t):
    func = _func(dim, hidden, 41, time_dependent=True)
    g = torch.Generator(device='cpu').manual_seed(42)
    y = torch.randn(batch, dim, generator=g).to(dev())
    a = torch.randn(batch, dim, generator=g).to(dev())
    eng = _engine(batch, dim, hidden, time_dependent=True)
    try:
        assert eng.n_params == sum(p.numel() for p in func.parameters())
        f, vy, vp = eng.dynamics(func.device_rhs(), y, a, t)
    finally:
        eng.close()
    yr = y.clone().requires_grad_(True)
    tr = torch.tensor(t, device=dev(), requires_grad=True)
    fr = func(tr, yr)
    grads = torch.autograd.grad(fr, (tr, yr) + tuple(func.parameters()), -a)
    assert _rel(f, fr.detach()) < 3e-6
    assert _rel(vy, grads[1]) < 3e-6
    assert _rel(vp, _canon(func, grads[2:])) < 1e-5          # the canonical W1 block is [1 + dim, hidden]: its first row is the gradient of w_t
    # the identity the segment kernel integrates adj_t with: -a^T df/dt = dot(w_t, b1 slice of -a^T df/dtheta)
    w_t = func.fc1.weight.detach()[:, 0]
    o = (1 + dim) * hidden
    assert abs(float(torch.dot(w_t, vp[o:o + hidden])) - float(grads[0])) <= 2e-5 * max(abs(float(grads[0])), float(vp[o:o + hidden].abs().max()))


@pytest.mark.parametrize('batch,dim,hidden,tol,t0,t1', [
    (8, 4, 16, 1e-3, 1.0, 0.0), (100, 10, 16, 1e-4, 1.0, 0.0), (300, 64, 128, 1e-3, 1.0, 0.0), (64, 8, 32, 1e-6, 1.0, 0.25),
    (64, 8, 32, 1e-4, 0.5, 2.0), (2000, 33, 70, 1e-5, 0.7, -0.4)])
def test_time_dependent_segment_matches_the_plane_kernel_engine(batch, dim, hidden, tol, t0, t1):
    """One backward interval of the time-dependent network: same attempts and accepted steps as the generic tuple path; a, theta
    and the now evolving adj_t within fp32 roundoff per step of it."""
    func = _func(dim, hidden, 43, time_dependent=True)
    g = torch.Generator(device='cpu').manual_seed(44)
    y = torch.randn(batch, dim, generator=g).to(dev())
    a = (torch.randn(batch, dim, generator=g) / batch).to(dev())
    eng = _engine(batch, dim, hidden, tol, time_dependent=True)
    theta = (0.01 * torch.randn(eng.n_params, generator=g)).to(dev())
    adj_t = torch.tensor(0.3, device=dev())
    try:
        a1, t_1, p1 = eng.segment(func.device_rhs(), y, a, adj_t, theta, t0, t1)
        st = eng.stats.as_dict()
    finally:
        eng.close()
    ref, rs = _plane_segment(func, y, a, adj_t, theta, t0, t1, tol)
    assert st['status'] == 0 and st['n_launches'] == 1
    assert (st['n_attempts'], st['n_accepted']) == (rs['n_attempts'], rs['n_accepted'])
    nsteps = max(st['n_accepted'], 1)
    assert _rel(a1, ref[1][1]) < 4e-6 * nsteps
    assert _rel(p1, ref[3][1]) < 4e-6 * nsteps
    # adj_t: the generic path sums the batch's -a df/dt per stage in fp32 (one rocBLAS reduction of batch x hidden products), the
    # kernel takes dot(w_t, .) of fp32 column sums: both carry ~sqrt(batch) ulp of the LARGEST product, not of the (cancelling) sum
    scale = max(abs(float(ref[2][1])), abs(float(adj_t)), float(ref[3][1].abs().max()))
    assert abs(float(t_1) - float(ref[2][1])) <= 2e-5 * nsteps * scale, (float(t_1), float(ref[2][1]))
    assert float(t_1) != float(adj_t)                      # it did evolve


@pytest.mark.parametrize('batch,dim,hidden,tol,act', [(64, 8, 32, 1e-3, 'tanh'), (1000, 16, 64, 1e-4, 'softplus'), (4096, 64, 128, 1e-3, 'relu')])
def test_time_dependent_odeblock_trains_on_the_fused_adjoint(batch, dim, hidden, tol, act):
    from tfdiffeq_amd import models
    torch.manual_seed(45)
    block = models.ODEBlock(models.ODEFunc(dim, hidden, time_dependent=True, non_linearity=act), tol=tol, adjoint=True).to(dev())
    x = torch.randn(batch, dim, generator=torch.Generator().manual_seed(46)).to(dev())
    out_f, gx_f, gp_f, st_f = _grads(block, x, True)
    out_p, gx_p, gp_p, st_p = _grads(block, x, False)
    assert st_f['engine'].startswith('fused adjoint kernel') and st_p['engine'] == 'plane kernels'
    assert all(s['n_launches'] == 1 and s['status'] == 0 for s in st_f['segments'])
    assert torch.equal(out_f, out_p)
    # Band: the solver tolerance (see test_odeblock_gradients_fused_against_plane_kernel_adjoint) - two solves whose controllers
    # may pick different valid step sequences differ by a small multiple of it (observed: 1.3 x tol for softplus at 1e-4);
    # relu: kinks, see test_default_odenet_trains_on_the_fused_kernels
    band = 3 * max(2e-5, tol) * (10 if act == 'relu' else 1)
    assert _rel(gx_f, gx_p) < band
    for a_, b_ in zip(gp_f, gp_p):
        assert _rel(a_, b_) < band


@pytest.mark.parametrize('rtol,atol,band', [(1e-5, 1e-7, 1e-4), (1e-7, 1e-9, 1e-5)])
def test_time_dependent_time_gradients_over_several_intervals(rtol, atol, band):
    """dL/dt_i of the time-dependent network (adjoint.py:134-140, 162-166): adj_t now integrates -a^T df/dt between the output times.
    Band: ten solver tolerances at 1e-5 (adj_t's error estimate is a dot product of cancelling terms - the two engines' controllers
    pick different, equally valid step sequences; observed 4e-5), and the agreement tightens with the tolerance (observed 2e-6 at 1e-7)."""
    from tfdiffeq_amd import odeint_adjoint
    from tfdiffeq_amd import adjoint as ADJ
    func = _func(5, 24, 47, time_dependent=True)
    y0 = torch.randn(70, 5, generator=torch.Generator().manual_seed(48)).to(dev())
    w = torch.randn(4, 70, 5, generator=torch.Generator().manual_seed(49)).to(dev())
    res = {}
    for fused in (True, False):
        ADJ.FUSED, ADJ.FUSED_FORWARD = fused, False
        try:
            for p in func.parameters():
                p.grad = None
            yi = y0.clone().requires_grad_(True)
            t = torch.tensor([0.0, 0.3, 0.8, 1.5], requires_grad=True)
            sol = odeint_adjoint(func, yi, t, rtol=rtol, atol=atol, method='dopri5')
            (sol * w).sum().backward()
            res[fused] = (sol.detach(), yi.grad.clone(), t.grad.clone(), [p.grad.clone() for p in func.parameters()],
                          dict(odeint_adjoint.last_backward_stats))
        finally:
            ADJ.FUSED, ADJ.FUSED_FORWARD = True, True
    assert res[True][4]['engine'].startswith('fused') and len(res[True][4]['segments']) == 3
    assert torch.equal(res[True][0], res[False][0])
    assert _rel(res[True][1], res[False][1]) < band
    assert _rel(res[True][2], res[False][2]) < band
    assert float(res[True][2].abs().min()) > 0               # every dL/dt_i is live
    for a_, b_ in zip(res[True][3], res[False][3]):
        assert _rel(a_, b_) < band


def test_time_dependent_gradients_against_autograd_through_the_restatement():
    """The independent checker for the time-dependent network: float64 autograd through oracle/ode_torch_cpu.odeint_dopri5 (rtol
    1e-9) = the exact gradient of the exact flow; the fused fp32 adjoint at rtol = atol = 1e-6 agrees to the adjoint's own accuracy.
    A-priori band 2e-4: what the time-independent twin of this test observes (2e-5, tests/golden/fp32_bands.json) x 10."""
    from tfdiffeq_amd import odeint_adjoint
    from oracle import ode_torch_cpu as TC
    func = _func(6, 24, 50, time_dependent=True)
    cpu64 = copy.deepcopy(func).cpu().double()
    y0 = torch.randn(40, 6, generator=torch.Generator().manual_seed(51))
    w = torch.randn(40, 6, generator=torch.Generator().manual_seed(52))
    y64 = y0.double().requires_grad_(True)
    sol64, _ = TC.odeint_dopri5(lambda t_, y_: cpu64(t_, y_), y64, [0.25, 1.5], rtol=1e-9, atol=1e-11)
    (sol64[1] * w.double()).sum().backward()
    yi = y0.to(dev()).requires_grad_(True)
    sol = odeint_adjoint(func, yi, torch.tensor([0.25, 1.5]), rtol=1e-6, atol=1e-6, method='dopri5')
    (sol[1] * w.to(dev())).sum().backward()
    assert odeint_adjoint.last_backward_stats['engine'].startswith('fused')
    assert _rel(yi.grad.cpu().double(), y64.grad) < 2e-4
    for pg, pc in zip(func.parameters(), cpu64.parameters()):
        assert _rel(pg.grad.cpu().double(), pc.grad) < 2e-4
    assert float(func.fc1.weight.grad[:, 0].abs().max()) > 0        # the column of fc1 that multiplies t is trained too
