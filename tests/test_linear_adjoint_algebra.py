"""CPU tests of oracle/linear_adjoint_numpy.py: the stage derivatives of the parameter component of odeint_adjoint's augmented system
(/root/reference/tfdiffeq/adjoint.py:69-105) for f = y W + b - the direct form against torch.autograd (the reference's route: a tape over
f), and the factored form the planned one-launch kernel uses (one outer product over the batch per step) against the
direct one."""
import numpy as np
import pytest
import torch

from oracle import linear_adjoint_numpy as LA
from oracle import ode_numpy as O


def _problem(D, B, bias, seed):
    rng = np.random.default_rng(seed)
    S_ = rng.standard_normal((D, D))
    W = -0.4 * np.eye(D) + 0.6 * (S_ - S_.T) / np.sqrt(D) + 0.1 * rng.standard_normal((D, D)) / np.sqrt(D)
    b = 0.3 * rng.standard_normal(D) if bias else None
    return W, b, rng.standard_normal((B, D)), rng.standard_normal((B, D)) / B


@pytest.mark.parametrize('D,B,bias,s,h', [(3, 7, True, 1.0, 0.2), (6, 50, False, -1.0, 0.35), (16, 300, True, -1.0, 0.1), (5, 1, True, 1.0, 0.8)])
def test_direct_stage_derivatives_are_the_vjps_autograd_gives(D, B, bias, s, h):
    W, b, y0, a0 = _problem(D, B, bias, D + B)
    direct = LA.theta_stage_derivatives_direct(W, b, y0, a0, h, s, O.DOPRI5)
    fy = (lambda y: s * (y @ W + b)) if bias else (lambda y: s * (y @ W))
    Ys, _ = LA.stage_inputs(fy, y0, h, O.DOPRI5)
    As, _ = LA.stage_inputs(lambda a: -s * (a @ W.T), a0, h, O.DOPRI5)
    for (kW, kb), Y, A in zip(direct, Ys, As):
        Wt = torch.tensor(W, requires_grad=True)
        bt = torch.tensor(b, requires_grad=True) if bias else None
        f = torch.tensor(Y) @ Wt + (bt if bias else 0.0)
        g = torch.autograd.grad(f, (Wt,) + ((bt,) if bias else ()), torch.tensor(-A))       # adjoint.py:83-95: vjp with -adj_y
        assert np.abs(s * g[0].numpy() - kW).max() <= 1e-13 * max(1.0, np.abs(kW).max())     # (time reversal: the sign, misc.py:318-321)
        if bias:
            assert np.abs(s * g[1].numpy() - kb).max() <= 1e-13 * max(1.0, np.abs(kb).max())


@pytest.mark.parametrize('tableau', ['DOPRI5', 'BOSH3'])
@pytest.mark.parametrize('D,B,bias,s,h', [(3, 7, True, 1.0, 0.2), (6, 50, False, -1.0, 0.35), (16, 300, True, -1.0, 0.1), (32, 1000, True, 1.0, 0.05),
                                          (5, 1, True, -1.0, 0.8)])
def test_factored_stage_derivatives_equal_the_direct_ones(tableau, D, B, bias, s, h):
    tb = getattr(O, tableau)
    W, b, y0, a0 = _problem(D, B, bias, 3 * D + B)
    direct = LA.theta_stage_derivatives_direct(W, b, y0, a0, h, s, tb)
    factored = LA.theta_stage_derivatives_factored(W, b, y0, a0, h, s, tb)
    assert len(direct) == len(factored) == len(tb.beta) + 1
    for (kW, kb), (qW, qb) in zip(direct, factored):
        assert np.abs(kW - qW).max() <= 1e-12 * max(1.0, np.abs(kW).max())
        assert np.abs(kb - qb).max() <= 1e-12 * max(1.0, np.abs(kb).max())
    # and with them any combination over the stages - the step, the error estimate (rk_common.py:54-60), the dense-output fold
    for c in (tb.c_sol, tb.c_error):
        dW = sum((h * cj) * kW for cj, (kW, _) in zip(c, direct))
        fW = sum((h * cj) * qW for cj, (qW, _) in zip(c, factored))
        assert np.abs(dW - fW).max() <= 1e-12 * max(1.0, np.abs(dW).max())


@pytest.mark.parametrize('x', [0.0, 0.137, 0.5, 0.8125, 1.0])
def test_dense_output_as_one_combination_of_the_stage_derivatives(x):
    """interp.py:6-67 + dopri5.py:39-45 (oracle: interp_fit_mid / interp_evaluate) against the folded weights the fused adjoint kernel
    uses for its parameter component - float64, a nonlinear right-hand side so that the k_j are independent."""
    rng = np.random.default_rng(3)
    y0 = (rng.standard_normal((7, 4)),)
    f = lambda t, y: (np.tanh(y[0] @ M) + 0.3 * np.cos(t) * y[0],)  # noqa: E731
    M = rng.standard_normal((4, 4))
    t0, dt = 0.3, 0.21
    f0 = f(t0, y0)
    y1, f1, _, k = O.runge_kutta_step(f, y0, f0, t0, dt, O.DOPRI5)
    co = O.interp_fit_mid(y0, y1, k, dt, O.DOPRI5_C_MID)
    ref = O.interp_evaluate(co, t0, t0 + dt, t0 + x * dt)[0]
    w = LA.dense_output_fold_weights(O.DOPRI5, O.DOPRI5_C_MID, x)
    got = y0[0] + sum((dt * wj) * kj for wj, kj in zip(w, k[0]))
    assert np.abs(got - ref).max() <= 2e-14 * max(1.0, np.abs(ref).max())
    if x == 1.0:
        assert np.abs(got - y1[0]).max() <= 1e-15 * max(1.0, np.abs(y1[0]).max())


@pytest.mark.parametrize('act', ['tanh', 'relu', 'softplus'])
def test_time_dependent_network_time_vjp_is_a_dot_product_with_the_bias_vjp(act):
    """The identity the fused adjoint kernel integrates adj_t with (csrc/mi_ode_adjoint.h, round 4): for the network of
    dense_odenet.py:79-84 - fc1 over concat([t, x]) - the first pre-activation is W1x^T x + t w_t + b1, hence
        -a^T df/dt = dot(w_t, -a^T df/db1)        and        -a^T df/dw_t = t (-a^T df/db1)
    checked against torch.autograd on the CPU (the reference's route: a tape over f, adjoint.py:83-95)."""
    from tfdiffeq_amd.models import ODEFunc
    torch.manual_seed(0)
    func = ODEFunc(5, 12, time_dependent=True, non_linearity=act).double()
    x = torch.randn(9, 5, dtype=torch.float64, requires_grad=True)
    a = torch.randn(9, 5, dtype=torch.float64)
    t = torch.tensor(0.73, dtype=torch.float64, requires_grad=True)
    f = func(t, x)
    g_t, g_w1, g_b1 = torch.autograd.grad(f, (t, func.fc1.weight, func.fc1.bias), -a)
    w_t = func.fc1.weight.detach()[:, 0]                      # the column of fc1 that multiplies t ([out, in] layout; row 0 of the Keras kernel)
    assert abs(float(torch.dot(w_t, g_b1)) - float(g_t)) <= 1e-13 * max(1.0, abs(float(g_t)))
    assert float((g_w1[:, 0] - float(t.detach()) * g_b1).abs().max()) <= 1e-13 * max(1.0, float(g_b1.abs().max()))


@pytest.mark.parametrize('tableau', ['DOPRI5', 'BOSH3'])
@pytest.mark.parametrize('D,B,bias,s,h', [(3, 7, True, 1.0, 0.2), (6, 50, False, -1.0, 0.35), (16, 300, True, -1.0, 0.1), (64, 500, True, -1.0, 0.25), (5, 1, True, -1.0, 0.8)])
def test_power_form_of_the_parameter_combinations(tableau, D, B, bias, s, h):
    """What csrc/mi_ode_linadj.h evaluates (round 5): any combination over the stages from the start state's (S + 1)^2 small products M_pq - held
    to the direct form (the stage inputs of the whole batch) at 1e-13 of the terms' scale; the ERROR ESTIMATE, a difference of O(1) terms that
    is itself O(h^5), keeps an absolute error of 1e-15 of that scale."""
    tb = getattr(O, tableau)
    W, b, y0, a0 = _problem(D, B, bias, 5 * D + B)
    direct = LA.theta_stage_derivatives_direct(W, b, y0, a0, h, s, tb)
    scale = h * max(np.abs(kW).max() for kW, _ in direct)
    scale_b = h * max(max(np.abs(kb).max() for _, kb in direct), 1e-300)
    for c in (tb.c_sol, tb.c_error):
        dW = sum((h * cj) * kW for cj, (kW, _) in zip(c, direct))
        db = sum((h * cj) * kb for cj, (_, kb) in zip(c, direct))
        pW, pb = LA.theta_combination_powers(W, b, y0.T @ a0, a0.sum(0), h, s, tb, c)
        assert np.abs(dW - pW).max() <= 1e-13 * scale
        assert np.abs(db - pb).max() <= 1e-13 * scale_b


@pytest.mark.parametrize('D,B,bias,s,h', [(6, 50, True, -1.0, 0.3), (32, 400, True, -1.0, 0.2), (16, 300, False, 1.0, 0.1)])
def test_the_product_over_the_batch_is_carried_from_step_to_step(D, B, bias, s, h):
    """G0 = y^T a and g0 = sum_rows a of a step's END state from its START state's small products (oracle: end_state_products_powers) - the
    kernel forms the product over the batch once per backward interval; after eight steps of propagation it still equals the product of the
    actual states to 1e-13."""
    W, b, y, a = _problem(D, B, bias, 7 * D + B)
    G, g = y.T @ a, a.sum(0)
    fy = (lambda y_: s * (y_ @ W + b)) if bias else (lambda y_: s * (y_ @ W))
    for _ in range(8):
        Ys, _k = LA.stage_inputs(fy, y, h, O.DOPRI5)
        As, _k = LA.stage_inputs(lambda a_: -s * (a_ @ W.T), a, h, O.DOPRI5)
        G, g = LA.end_state_products_powers(W, b, G, g, h, s, O.DOPRI5)
        y, a = Ys[-1], As[-1]                                    # FSAL shaped: y1 = the last stage input (rk_common.py:54-58)
    assert np.abs(G - y.T @ a).max() <= 1e-13 * max(1.0, np.abs(G).max())
    assert np.abs(g - a.sum(0)).max() <= 1e-13 * max(1.0, np.abs(g).max())
