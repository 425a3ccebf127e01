"""ORACLE - test infrastructure, not product code.

The augmented dynamics of odeint_adjoint (/root/reference/tfdiffeq/adjoint.py:69-105) for the LINEAR right-hand side f = y W + b, and the
identity the planned one-launch kernel rests on: the parameter component's Runge-Kutta stage derivatives
need no per-stage GEMM over the batch.

Direct form (what the reference computes through its tape, stage by stage, rk_common.py:22-61):
    k_W,sigma = -s Y_sigma^T A_sigma        k_b,sigma = -s sum_rows A_sigma
with Y_sigma, A_sigma the stage inputs of the y and a = adj_y components ([batch, dim] each), s = +1 / -1 for increasing / decreasing time
(misc.py:311-321).

Factored form: y' = s (y W + b) and a' = -s a W^T are autonomous and (affine-)linear, so every stage input is the initial value times a
matrix that depends on (h, W) only, plus - for y - an affine row that depends on (h, W, b) only:
    Y_sigma = y0 Pi_sigma + 1 r_sigma          A_sigma = a0 Phi_sigma
    Y_sigma^T A_sigma = Pi_sigma^T (y0^T a0) Phi_sigma + r_sigma^T ((1^T a0) Phi_sigma)
i.e. with G0 = y0^T a0 ([dim, dim]: ONE product over the batch per step) and g0 = sum_rows a0:
    U_sigma = G0 Phi_sigma   and   v_sigma = g0 Phi_sigma     are the a-type stage inputs started from the rows of G0 and from g0,
    Pi_sigma^T U_sigma = (U_sigma^T Pi_sigma)^T               are the HOMOGENEOUS y-type stage-sigma inputs started from the rows of U_sigma^T,
    r_sigma                                                   is the y-type stage-sigma input started from the zero row.
Everything after G0 / g0 is Runge-Kutta stages on dim (+1) rows: the work of a few 16-row tiles of the existing tile pass.

`stage_inputs` restates rk_common.py:44-52 (the stage loop) and returns the stage INPUTS; `theta_stage_derivatives_direct` /
`_factored` are the two forms; tests/test_linear_adjoint_algebra.py holds them together to 1e-13 and checks the direct form against
torch.autograd (the reference's own route) - CPU only.
"""
import numpy as np


def stage_inputs(f, y0, h, tableau):
    """rk_common.py:44-52 for an autonomous f: the inputs y_sigma of every evaluation k_sigma = f(y_sigma), sigma = 0 .. S (sigma = 0: y0
    itself, the FSAL evaluation; sigma = S: the solution of an FSAL-shaped tableau), and the k's."""
    k = [f(y0)]
    ys = [y0]
    for beta_i in tableau.beta:
        yi = y0 + sum((h * b) * k_ for b, k_ in zip(beta_i, k))      # rk_common.py:51 (order of the sum: as misc._scaled_dot_product)
        ys.append(yi)
        k.append(f(yi))
    return ys, k


def theta_stage_derivatives_direct(W, b, y0, a0, h, s, tableau):
    """(k_W,sigma, k_b,sigma) for sigma = 0 .. S from the stage inputs of the full batch."""
    fy = (lambda y: s * (y @ W + b)) if b is not None else (lambda y: s * (y @ W))
    fa = lambda a: -s * (a @ W.T)  # noqa: E731
    Ys, _ = stage_inputs(fy, y0, h, tableau)
    As, _ = stage_inputs(fa, a0, h, tableau)
    return [(-s * (Y.T @ A), -s * A.sum(0)) for Y, A in zip(Ys, As)]


def theta_stage_derivatives_factored(W, b, y0, a0, h, s, tableau):
    """The same from G0 = y0^T a0 and g0 = sum_rows a0 alone (module docstring)."""
    D = W.shape[0]
    fa = lambda a: -s * (a @ W.T)  # noqa: E731
    fy_hom = lambda y: s * (y @ W)  # noqa: E731
    G0, g0 = y0.T @ a0, a0.sum(0)
    UV, _ = stage_inputs(fa, np.vstack([G0, g0[None, :]]), h, tableau)          # rows of G0 and the row g0 through the a-system
    if b is not None:
        R, _ = stage_inputs(lambda y: s * (y @ W + b), np.zeros((1, D)), h, tableau)   # the affine response r_sigma
    out = []
    for sigma, uv in enumerate(UV):
        U, v = uv[:D], uv[D]
        Ysig, _ = stage_inputs(fy_hom, U.T.copy(), h, tableau)                   # only stage sigma of this run is used: U^T Pi_sigma
        kW = Ysig[sigma].T
        if b is not None:
            kW = kW + np.outer(R[sigma][0], v)
        out.append((-s * kW, -s * v))
    return out


def dense_output_fold_weights(tableau, c_mid, x):
    """The dense output of dopri5 (interp.py:6-67 over dopri5.py:39-45's y_mid) as ONE combination of the stage derivatives:
        y(x) = y0 + dt sum_j w_j(x) k_j,     w_j = b_j (-8x^4 + 14x^3 - 5x^2) + c_mid_j (16x^4 - 32x^3 + 16x^2)
                                                   + [j = 0] (-2x^4 + 5x^3 - 4x^2 + x) + [j = S] (2x^4 - 3x^3 + x^2)
    The fit is linear in (y0, y1, y_mid, f0, f1), each of which is y0 plus a combination of the k_j (y1: b = c_sol, y_mid: c_mid, f0 = k_0,
    f1 = k_S), and the y0 terms of the x^4, x^3, x^2 coefficients cancel (-8 - 8 + 16 = 18 + 14 - 32 = -11 - 5 + 16 = 0).  This is what the
    fused adjoint kernel forms for its parameter component (csrc/mi_ode_adjoint.h: one weight-gradient pass instead of three) and what the
    planned linear kernel needs; `x` = (t - t0) / dt."""
    S = len(tableau.beta)
    x2, x3, x4 = x * x, x * x * x, x * x * x * x
    p1 = -8 * x4 + 14 * x3 - 5 * x2
    pm = 16 * x4 - 32 * x3 + 16 * x2
    p0 = -2 * x4 + 5 * x3 - 4 * x2 + x
    pS = 2 * x4 - 3 * x3 + x2
    return [tableau.c_sol[j] * p1 + c_mid[j] * pm + (p0 if j == 0 else 0.0) + (pS if j == S else 0.0) for j in range(S + 1)]


# ---- the power form (round 5): what csrc/mi_ode_linadj.h evaluates ---------------------------------------------------------------------
# Both systems are linear with CONSTANT matrices, so the stage inputs are polynomials in h W whose coefficients depend on the tableau
# only:   Y_sigma = y0 R_sigma(s h W) + 1 rho_sigma,   A_sigma = a0 R_sigma(-s h W^T),   R_0 = 1,  R_sigma(z) = 1 + z sum_j beta_sigma,j R_j(z)
# (rho_sigma = b [R_sigma(s h W) - 1] W^-1, i.e. the same coefficients shifted by one power).  With pi[sigma][p] the coefficient of z^p in
# R_sigma, P_p = (W^T)^p, c_p = P_{p-1} b^T (c_0 = 0), G0 = y0^T a0 and g0 = sum_rows a0:
#     Y_sigma^T A_sigma = sum_pq pi[sigma][p] pi[sigma][q] (s h)^p (-s h)^q  M_pq,        M_pq = (P_p G0 + c_p g0) P_q
# and ANY combination over the stages (solution, error estimate, dense output at t_end) is
#     sum_sigma (h c_sigma) k_W,sigma = -s h sum_pq K^c_pq (s h)^p (-s h)^q M_pq,         K^c_pq = sum_sigma c_sigma pi[sigma][p] pi[sigma][q]
#     sum_sigma (h c_sigma) k_b,sigma = -s h sum_q  K^c_0q (-s h)^q (g0 P_q).
# The M_pq depend on the step's START STATE only (not on h): one set of (S+1)^2 small products per ACCEPTED step, then an attempt costs an
# elementwise combination.  tests/test_linear_adjoint_algebra.py holds this form to the direct one.

def stage_polynomials(tableau):
    """pi[sigma][p], sigma = 0 .. S, p = 0 .. S: the coefficient of z^p in the stage-input polynomial R_sigma(z) (rk_common.py:44-52 applied to
    y' = y z)."""
    S = len(tableau.beta)
    pi = np.zeros((S + 1, S + 1))
    pi[0, 0] = 1.0
    for sg in range(1, S + 1):
        pi[sg, 0] = 1.0
        for j, bj in enumerate(tableau.beta[sg - 1]):
            pi[sg, 1:] += bj * pi[j, :-1]
    return pi


def combination_table(tableau, c):
    """K^c[p][q] = sum_sigma c_sigma pi[sigma][p] pi[sigma][q]."""
    pi = stage_polynomials(tableau)
    c = np.asarray(c, dtype=np.float64)
    return np.einsum('s,sp,sq->pq', c, pi, pi)


def start_state_products_S(W, b, G0, g0, S):
    """(M[p][q], r[q]) of the comment above: M_pq = (P_p G0 + c_p g0) P_q  [D, D],  r_q = g0 P_q  [D], p, q = 0 .. S."""
    D = W.shape[0]
    P = [np.eye(D)]
    for _ in range(S):
        P.append(P[-1] @ W.T)
    L = []
    for p in range(S + 1):
        Lp = P[p] @ G0
        if b is not None and p >= 1:
            Lp = Lp + np.outer(P[p - 1] @ b, g0)
        L.append(Lp)
    M = [[L[p] @ P[q] for q in range(S + 1)] for p in range(S + 1)]
    r = [g0 @ P[q] for q in range(S + 1)]
    return M, r


def theta_combination_powers(W, b, G0, g0, h, s, tableau, c):
    """(sum_sigma (h c_sigma) k_W,sigma, sum_sigma (h c_sigma) k_b,sigma) from the start-state products."""
    S = len(tableau.beta)
    K = combination_table(tableau, c)
    M, r = start_state_products_S(W, b, G0, g0, S)
    dW = np.zeros_like(G0)
    db = np.zeros_like(g0)
    for p in range(S + 1):
        for q in range(S + 1):
            dW = dW + (K[p, q] * (s * h) ** p * (-s * h) ** q) * M[p][q]
    for q in range(S + 1):
        db = db + (K[0, q] * (-s * h) ** q) * r[q]
    return -s * h * dW, -s * h * db


def end_state_products_powers(W, b, G0, g0, h, s, tableau):
    """(y1^T a1, sum_rows a1) of the step's END state from the start-state products alone: for an FSAL-shaped tableau y1 = Y_S and a1 = A_S are
    the last stage inputs, so  y1^T a1 = sum_pq pi[S][p] pi[S][q] (s h)^p (-s h)^q M_pq  and  sum_rows a1 = sum_q pi[S][q] (-s h)^q (g0 P_q) -
    the combination of the M_pq with the table pi_S pi_S^T and NO factor -s h.  csrc/mi_ode_linadj.h carries G0 | g0 from step to step this
    way: one product over the batch per backward INTERVAL (at its start state), none per step."""
    S = len(tableau.beta)
    pi = stage_polynomials(tableau)
    M, r = start_state_products_S(W, b, G0, g0, S)
    G1 = np.zeros_like(G0)
    g1 = np.zeros_like(g0)
    for p in range(S + 1):
        for q in range(S + 1):
            G1 = G1 + (pi[S, p] * pi[S, q] * (s * h) ** p * (-s * h) ** q) * M[p][q]
    for q in range(S + 1):
        g1 = g1 + (pi[S, q] * (-s * h) ** q) * r[q]
    return G1, g1
