"""GPU tests of the linear right-hand side beyond dim 128 (round-5 review, item 6): the 256-wide MFMA tile kernels with W streamed from a
copy in consumption order (csrc/mi_ode_step_fused.h, LinCtx<T, 256>::STREAM; dims 129 .. 256, zero padded) - whole call in one launch,
launch per attempt and fixed grid - against the numpy oracle on the same system: the reference's exact attempt / accept counts in
float64 and agreement at 1e-11; float32 inside the bands.  f(t, y) = y @ W (+ b): /root/reference/tests/problems.py:43-68 at a wider state."""
import numpy as np
import pytest
import torch

import oracle.ode_numpy as O

pytestmark = pytest.mark.gpu


def dev():
    return torch.device('cuda:0')


def _system(D, batch, dtype, seed=2):
    g2 = torch.Generator().manual_seed(seed)
    S = torch.randn(D, D, generator=g2, dtype=torch.float64)
    A = -0.5 * torch.eye(D, dtype=torch.float64) + 0.5 * (S - S.t()) / np.sqrt(D)
    y0 = torch.randn(batch, D, generator=torch.Generator().manual_seed(seed + 1), dtype=torch.float64)
    b = 0.1 * torch.randn(D, generator=torch.Generator().manual_seed(9), dtype=torch.float64)
    return A.t().contiguous().to(dtype), y0.to(dtype), b.to(dtype)


CASES = [('dopri5', [0., 1.], False), ('dopri5', list(np.linspace(0., 2., 7)), True), ('tsit5', [0., 0.4, 1.], False), ('bosh3', [0., 0.5], True),
         ('rk4', list(np.linspace(0., 1., 6)), False), ('euler', list(np.linspace(0., 1., 9)), True), ('dopri5', [1., 0.], False)]


@pytest.mark.parametrize('D,batch', [(129, 1000), (200, 33), (256, 4100)])
@pytest.mark.parametrize('dtype', [torch.float64, torch.float32])
@pytest.mark.parametrize('method,tt,bias', CASES)
def test_wide_linear_system_against_the_oracle(D, batch, dtype, method, tt, bias):
    from tfdiffeq_amd import odeint, rhs
    W, y0, b = _system(D, batch, dtype)
    t = np.array(tt)
    kw = dict(rtol=1e-6, atol=1e-9) if dtype == torch.float64 else dict(rtol=1e-4, atol=1e-6)
    Wn, bn = W.numpy(), b.numpy()
    fo = (lambda t_, y: y @ Wn + bn) if bias else (lambda t_, y: y @ Wn)
    ref, st_ref = O.odeint(fo, y0.numpy(), t.astype(Wn.dtype), method=method, return_stats=True,
                           options={'tsit5_fixed': True} if method == 'tsit5' else None, **kw)     # (tsit5: the published tableau)
    f = rhs.Linear(W, b if bias else None)
    for fusion in ('auto', 'step'):
        if fusion == 'step' and method in ('rk4', 'euler'):
            continue
        sol = odeint(f, y0.to(dev()), torch.tensor(t), method=method, options={'fusion': fusion}, **kw)
        st = dict(odeint.last_stats)
        diff = float(np.abs(sol.cpu().numpy() - ref).max())
        if fusion == 'auto':
            assert st['n_launches'] == 1, st
        if method not in ('rk4', 'euler'):
            if dtype == torch.float64:
                assert (st['n_attempts'], st['n_accepted']) == (st_ref.n_attempts, st_ref.n_accepted), (fusion, st, vars(st_ref))
            else:
                assert abs(st['n_attempts'] - st_ref.n_attempts) <= 1, (fusion, st, vars(st_ref))
        assert diff < (1e-11 if dtype == torch.float64 else 2e-4), (fusion, diff)


def test_wide_linear_matrix_updated_in_place_is_seen_by_the_next_call():
    """The streamed kernels read a COPY of W: it is refreshed on the stream in front of every launch, so an optimizer step on W
    between two calls of the same (cached) engine is seen."""
    from tfdiffeq_amd import odeint, rhs
    W, y0, _ = _system(256, 64, torch.float64)
    Wd = W.to(dev())
    f = rhs.Linear(Wd)
    t = torch.tensor([0., 1.])
    a = odeint(f, y0.to(dev()), t, rtol=1e-6, atol=1e-9)
    Wd.mul_(0.5)
    b = odeint(f, y0.to(dev()), t, rtol=1e-6, atol=1e-9)
    Wn = (0.5 * W).numpy()
    ref = O.odeint(lambda t_, y: y @ Wn, y0.numpy(), np.array([0., 1.]), rtol=1e-6, atol=1e-9, method='dopri5')
    assert float((a - b).abs().max()) > 1e-3
    assert np.abs(b.cpu().numpy() - ref).max() < 1e-11


def test_wide_linear_callable_is_lowered_onto_the_tile_kernels():
    """`lambda t, y: y @ A` at dim 256 (the call shape of the reference's tests): lowered to rhs.Linear -> one launch of the 256-wide kernel."""
    from tfdiffeq_amd import odeint
    W, y0, _ = _system(256, 500, torch.float64)
    Wd = W.to(dev())
    sol = odeint(lambda t, y: y @ Wd, y0.to(dev()), torch.tensor([0., 1.]), rtol=1e-6, atol=1e-9, method='dopri5',
                 options={'lower': 'auto'})              # (tests/conftest.py keeps the lowering off by default outside its own modules)
    st = dict(odeint.last_stats)
    Wn = W.numpy()
    ref, st_ref = O.odeint(lambda t_, y: y @ Wn, y0.numpy(), np.array([0., 1.]), rtol=1e-6, atol=1e-9, method='dopri5', return_stats=True)
    assert st['n_launches'] == 1 and st['n_attempts'] == st_ref.n_attempts, st
    assert np.abs(sol.cpu().numpy() - ref).max() < 1e-11


def test_config4_shape_at_dim_256_full_size_against_the_oracle():
    """65536 x 256 float64, dopri5 rtol 1e-6 atol 1e-9, t = [0, 1] (config 4 at twice the width) against the oracle on the SAME full-size
    input: one launch, identical step sequence, agreement at 1e-9 (measured ~1e-14); and against the closed form y0 expm(W)."""
    from tfdiffeq_amd import odeint, rhs
    W, y0, _ = _system(256, 65536, torch.float64)
    sol = odeint(rhs.Linear(W), y0.to(dev()), torch.tensor([0., 1.]), rtol=1e-6, atol=1e-9, method='dopri5')
    st = dict(odeint.last_stats)
    assert st['n_launches'] == 1 and st['status'] == 0, st
    Wn = W.numpy()
    ref, st_ref = O.odeint(lambda t_, y: y @ Wn, y0.numpy(), np.array([0., 1.]), rtol=1e-6, atol=1e-9, method='dopri5', return_stats=True)
    assert (st['n_attempts'], st['n_accepted']) == (st_ref.n_attempts, st_ref.n_accepted), (st, vars(st_ref))
    assert np.abs(sol.cpu().numpy() - ref).max() < 1e-9
    exact = y0.numpy() @ torch.linalg.matrix_exp(W).numpy()               # y(1) = y0 exp(W)
    assert np.abs(sol[-1].cpu().numpy() - exact).max() < 1e-4            # (the solver's own tolerance, loosely)


@pytest.mark.parametrize('D', [130, 144, 145, 160, 161, 193, 224, 225, 230, 241, 255])
@pytest.mark.parametrize('dtype', [torch.float64, torch.float32])
def test_wide_linear_every_trip_count_and_idle_wavefront_pattern(D, dtype):
    """The chain of the 256-wide kernels stops at the state's row length (ceil(dim / 32) trips in float64, ceil(dim / 64) in float32) and a
    wavefront whose sixteen columns lie beyond dim skips it: every trip count, with and without idle wavefronts, against the oracle
    (dopri5 + bias, 37 rows: three 16-row tiles, the last one ragged)."""
    from tfdiffeq_amd import odeint, rhs
    W, y0, b = _system(D, 37, dtype, seed=D)
    kw = dict(rtol=1e-6, atol=1e-9) if dtype == torch.float64 else dict(rtol=1e-4, atol=1e-6)
    Wn, bn = W.numpy(), b.numpy()
    t = np.array([0., 0.7, 1.])
    ref, st_ref = O.odeint(lambda t_, y: y @ Wn + bn, y0.numpy(), t.astype(Wn.dtype), method='dopri5', return_stats=True, **kw)
    sol = odeint(rhs.Linear(W, b), y0.to(dev()), torch.tensor(t), method='dopri5', **kw)
    st = dict(odeint.last_stats)
    assert st['n_launches'] == 1
    assert abs(st['n_attempts'] - st_ref.n_attempts) <= (0 if dtype == torch.float64 else 1), (st, vars(st_ref))
    assert float(np.abs(sol.cpu().numpy() - ref).max()) < (1e-11 if dtype == torch.float64 else 2e-4)


def test_wide_linear_edges_batch_one_many_outputs_and_error_exits():
    """dim 200 on the 256-wide kernels: one trajectory, 1500 output times (beyond the kernel's LDS cache of output times), T = 1 (no integration),
    an output 1e-9 behind t0, `first_step`, and the reference's assertions - max_num_steps (dopri5.py:96-100 shape), non-finite W -> 'underflow in dt nan'."""
    from tfdiffeq_amd import odeint, rhs
    D = 200
    W, _, _ = _system(D, 1, torch.float64, seed=5)
    Wn = W.numpy()
    f = rhs.Linear(W)
    g = torch.Generator().manual_seed(6)
    for batch, tt in ((1, np.array([0., 1.])), (15, np.linspace(0., 2., 1500)), (16, np.array([0.3])), (17, np.array([0., 1e-9, 1.]))):
        y0 = torch.randn(batch, D, generator=g, dtype=torch.float64)
        ref, st = O.odeint(lambda t_, y: y @ Wn, y0.numpy(), tt, rtol=1e-6, atol=1e-9, method='dopri5', return_stats=True)
        sol = odeint(f, y0.to(dev()), torch.tensor(tt), rtol=1e-6, atol=1e-9, method='dopri5')
        s = dict(odeint.last_stats)
        assert np.abs(sol.cpu().numpy() - ref).max() < 1e-11, (batch, len(tt))
        if len(tt) > 1:
            assert s['n_launches'] == 1 and s['n_attempts'] == st.n_attempts, (batch, len(tt), s, vars(st))
    y0 = torch.randn(40, D, generator=g, dtype=torch.float64)
    with pytest.raises(AssertionError, match='max_num_steps exceeded'):
        odeint(f, y0.to(dev()), torch.tensor([0., 5.]), rtol=1e-9, atol=1e-11, method='dopri5', options={'max_num_steps': 2})
    ref, st = O.odeint(lambda t_, y: y @ Wn, y0.numpy(), np.array([0., 5.]), rtol=1e-9, atol=1e-11, method='dopri5', return_stats=True, options={'first_step': 0.01})
    sol = odeint(f, y0.to(dev()), torch.tensor([0., 5.]), rtol=1e-9, atol=1e-11, method='dopri5', options={'first_step': 0.01})
    assert dict(odeint.last_stats)['n_attempts'] == st.n_attempts and np.abs(sol.cpu().numpy() - ref).max() < 1e-11
    Wbad = W.clone()
    Wbad[0, 0] = float('nan')
    with pytest.raises(AssertionError, match='underflow in dt'):
        odeint(rhs.Linear(Wbad), y0.to(dev()), torch.tensor([0., 1.]), method='dopri5')
