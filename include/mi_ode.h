/*
 * mi_ode.h - C ABI of the MI355X-native explicit Runge-Kutta ODE engine.
 *
 * This is the drop-in boundary underneath the reference's three Python
 * contracts; the reference itself has no FFI, so each
 * entry point cites the reference interface it replaces (paths relative to
 * /root/reference/tfdiffeq/).  Plain C: integer status returns, raw device
 * pointers + sizes, no exceptions, no torch types.  All work is enqueued on the
 * caller's hipStream_t (passed as void*).  The caller owns every state / output
 * buffer; a handle owns only its workspace (stage planes, block partials, the
 * device-resident controller record).  One handle per host thread; no globals.
 *
 * Two families of entry points:
 *
 *  (A) fused engine  - the RHS f(t, y) is one of the device catalogue kinds and
 *      is evaluated INSIDE each stage kernel (y_sigma is never materialised),
 *      the step-size controller runs on the device, dense output is emitted by
 *      a kernel.  Replaces, for a single [batch, dim] state tensor:
 *        Dopri5Solver / Tsit5Solver / Bosh3Solver .before_integrate + .advance
 *          (dopri5.py:70-121, tsit5.py:91-151, bosh3.py:53-99)
 *        _runge_kutta_step                      (rk_common.py:22-61)
 *        _compute_error_ratio / _optimal_step_size / _select_initial_step
 *                                               (misc.py:183-287, tsit5.py:53-62)
 *        _interp_fit / _interp_evaluate         (interp.py:6-67, tsit5.py:33-50)
 *        FixedGridODESolver.integrate + Euler/RK4.step_func
 *                                               (solvers.py:82-104, fixed_grid.py:4-46)
 *
 *  (B) stateless plane kernels - for an arbitrary Python callable f (evaluated
 *      by the caller between launches) and tuple states.  They are the fused
 *      equivalents of the reference's eager-op sequences:
 *        mi_ode_lincomb        <- misc._scaled_dot_product (misc.py:118-121)
 *        mi_ode_error_norms    <- misc._compute_error_ratio reductions (misc.py:256-263)
 *        mi_ode_scaled_sumsq   <- misc._norm(x / scale)   (misc.py:170-175, 225-237)
 *        mi_ode_interp_eval    <- interp._interp_fit + _interp_evaluate / tsit5._interp_eval_tsit5
 *
 *  (C) opaque right-hand side, controller on the device (mi_ode_opq_*) - f is still
 *      evaluated by the caller between launches, but an attempt contains no host
 *      decision, so the caller can record it once as a hipGraph and replay it:
 *        mi_ode_opq_finish     <- rk_common.py:60 (error estimate) + misc._compute_error_ratio +
 *                                 the accept test + misc._optimal_step_size (dopri5.py:103-121)
 *        mi_ode_opq_commit     <- the accepted branch of _adaptive_dopri5_step (dopri5.py:113-121)
 *                                 + Dopri5Solver.advance's _interp_evaluate (dopri5.py:87)
 */
#ifndef MI_ODE_H
#define MI_ODE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 9: mi_ode_desc.multistep (fixed-grid Adams family).  10: multistep = 3 + ms_gamma_star (variable-order Adams), the four
 * mi_ode_adams_* plane entry points.  11: family (C) mi_ode_opq_* (opaque right-hand side, device-resident controller: what a
 * captured hipGraph of an attempt needs), mi_ode_stats.clock_mhz.
 * 13: (A''') mi_ode_linadj_* - the backward segment of odeint_adjoint for the linear right-hand side in one launch. */
#define MI_ODE_ABI_VERSION 13
#define MI_ODE_MAX_STAGES 13         /* rows of the tableau (dopri8 = 13, dopri5 / tsit5 = 6, bosh3 = 3, rk4 = 3, adaptive_heun = 1) */
#define MI_ODE_MAX_K (MI_ODE_MAX_STAGES + 1)
#define MI_ODE_MAX_LINCOMB 14        /* stateless lincomb: up to 14 planes (dopri8: f0 + 13 stages) */
#define MI_ODE_MAX_SEGMENTS 8        /* components of a tuple state packed into one buffer (mi_ode_desc.n_segments) */

/* ---- status bits (also the bits of mi_ode_stats.status) --------------------------------- */
#define MI_ODE_OK 0
#define MI_ODE_ST_DT_UNDERFLOW 0x1u    /* assert t0 + dt > t0            (dopri5.py:98)      */
#define MI_ODE_ST_NONFINITE 0x2u       /* assert _is_finite(abs(y0))     (dopri5.py:99-100)  */
#define MI_ODE_ST_MAX_STEPS 0x4u       /* assert n_steps < max_num_steps (dopri5.py:85-86)   */
#define MI_ODE_ST_SYNC_TIMEOUT 0x10u   /* engine: the in-kernel grid hand-off of the whole-integration kernel timed out */
#define MI_ODE_ST_BAD_T 0x8u           /* _assert_increasing / interpolation range (misc.py:158, interp.py:59) */
/* negative returns: API errors */
#define MI_ODE_E_INVALID (-1)          /* bad argument / unsupported combination */
#define MI_ODE_E_HIP (-2)              /* a HIP runtime call failed (see mi_ode_last_error) */
#define MI_ODE_E_NODEVICE (-3)
#define MI_ODE_E_EXCHANGE (-4)         /* the multi-rank exchange callback failed */

enum mi_ode_dtype { MI_ODE_F32 = 0, MI_ODE_F64 = 1 };

/* Device RHS catalogue.  All are trajectory(row)-local on a [batch, dim] row-major state.      */
enum mi_ode_rhs_kind {
  MI_ODE_RHS_LINEAR = 1,         /* f = y @ W (+ b)            W [dim,dim] row-major  (config 4: W = A^T)
                                    3 <= dim <= 128: MFMA tile kernels, W slices resident in registers (every schedule);
                                    129 <= dim <= 256 (round 6): 256-wide MFMA tile kernels, W streamed from a copy in
                                    consumption order - three- / six-row FSAL tableaus and the fixed grid, whole call /
                                    per attempt; anything else up to dim 256: vector-ALU stage kernels               */
  MI_ODE_RHS_CUBIC_LINEAR = 2,   /* f = (y**3) @ W             examples/ode_demo.py:33-35 (spiral)          */
  MI_ODE_RHS_LOTKA_VOLTERRA = 3, /* dim 2: [a u - b u v, -c v + d u v]   scalars = {a,b,c,d}                */
  MI_ODE_RHS_LORENZ = 4,         /* dim 3: examples/lorenz_attractor.py:20-37, scalars = {sigma,beta,rho}   */
  MI_ODE_RHS_MLP_TANH = 5,       /* dim->hidden->hidden->dim MLP of models/dense_odenet.py:41-92; hidden activation =
                                    scalars[0]: 0 tanh, 1 relu (the reference's default, :14), 2 softplus;
                                    scalars[1] != 0: time dependent (:79-84, fc1 sees concat([t, x])): w[0] is
                                    [dim + 1, hidden] and its row 0 multiplies the stage time.  float32, dim <= 64,
                                    hidden <= 128: MFMA tile kernels (every Runge-Kutta schedule).  Otherwise (float32 /
                                    float64, dim, hidden <= 256; round 5): a cooperative kernel, a thread per state element -
                                    adaptive 3- / 6-row tableaus as ONE launch per call (fusion 0 / 4, one rank, co-resident
                                    grid, T > 1: else MI_ODE_E_INVALID) and multistep != 0 (the Adams family in one launch) */
  MI_ODE_RHS_PLUGIN = 6          /* user device code for a trajectory-local system: mi_ode_rhs.plugin (csrc/mi_ode_plugin.h) -
                                    a trajectory per thread (dim <= 32) or, with mi_ode_rowlocal_plugin.cooperative = 1, a
                                    state element per thread (dim <= 256; adaptive methods as one launch per call, euler /
                                    rk4, the Adams family); scalars[0..7] and w[]/b[] are passed through to it          */
};

enum mi_ode_controller {
  MI_ODE_CTRL_MISC = 0,          /* misc._optimal_step_size: sqrt + float32-rounded exponent (misc.py:267-287) */
  MI_ODE_CTRL_TSIT5 = 1          /* tsit5._optimal_step_size: no sqrt, float64 exponent     (tsit5.py:53-62)  */
};

enum mi_ode_interp {
  MI_ODE_INTERP_QUARTIC_MID = 0, /* interp._interp_fit/_interp_evaluate with y_mid from c_mid (dopri5, bosh3) */
  MI_ODE_INTERP_TSIT5 = 1,       /* tsit5 seven-weight dense output from y0 (corrected)                       */
  MI_ODE_INTERP_TSIT5_REF = 2    /* tsit5 dense output from k[0] = f0, exactly as tsit5.py:45-50 (defect F6b) */
};

/* _ButcherTableau(alpha, beta, c_sol, c_error) (rk_common.py:5) + the dense-output mid-point weights. */
typedef struct mi_ode_tableau {
  int32_t n_stages;                                   /* S = len(alpha) */
  int32_t fsal;                                       /* 1 iff c_sol[-1]==0 and c_sol[:-1]==beta[-1] (rk_common.py:54) */
  double alpha[MI_ODE_MAX_STAGES];
  double beta[MI_ODE_MAX_STAGES][MI_ODE_MAX_STAGES];  /* row sigma uses entries [0..sigma] */
  double c_sol[MI_ODE_MAX_K];
  double c_error[MI_ODE_MAX_K];
  double c_mid[MI_ODE_MAX_K];                         /* DPS_C_MID / BS_C_MID; unused for tsit5 */
} mi_ode_tableau;

typedef struct mi_ode_rhs {
  int32_t kind;               /* enum mi_ode_rhs_kind */
  int32_t hidden;             /* MLP hidden width (else 0) */
  double sign;                /* +1, or -1 for a reversed time axis: f <- -f(-t, y) (misc.py:318-321) */
  double scalars[8];          /* kind specific scalars */
  const void* w[3];           /* device pointers, state dtype: LINEAR/CUBIC {W}; MLP {W1,W2,W3} ([in,out] row-major) */
  const void* b[3];           /* device pointers (nullable): biases */
  const void* plugin;         /* MI_ODE_RHS_PLUGIN: what the plugin's mi_ode_plugin_get(dtype) returned */
} mi_ode_rhs;

/* Exchange hook for batch-sharded runs: all-gather `count` doubles per rank.
 * sendbuf/recvbuf are DEVICE pointers; the hook must enqueue the collective so that it is ordered
 * after prior work on `stream` and before later work on it (RCCL via torch.distributed in the
 * Python shim).  recvbuf holds world_size * count doubles in rank order.  Return 0 on success. */
typedef int (*mi_ode_allgather_fn)(void* user, const void* sendbuf, void* recvbuf, int32_t count, void* stream);

typedef struct mi_ode_desc {
  int32_t dtype;              /* enum mi_ode_dtype: state dtype (time is always double in adaptive solvers) */
  int32_t adaptive;           /* 1: adaptive tableau solver; 0: fixed grid (tableau = the RK scheme, c_error unused) */
  int64_t batch;              /* rows held by THIS rank */
  int64_t dim;                /* state width */
  mi_ode_tableau tableau;
  mi_ode_rhs rhs;
  int32_t controller;         /* enum mi_ode_controller */
  int32_t interp;             /* enum mi_ode_interp */
  int32_t order;              /* controller order: 5 (dopri5, tsit5), 3 (bosh3)      (dopri5.py:68, bosh3.py:93) */
  int32_t init_order;         /* order passed to _select_initial_step: 4 / 2         (dopri5.py:74, bosh3.py:56) */
  double rtol, atol;          /* rtol[0], atol[0] of the reference's per-component lists */
  double safety, ifactor, dfactor;   /* pass the float32-rounded values the reference ends up with (misc.py:137-144) */
  double first_step;          /* NaN: select automatically (misc.py:183-247) */
  int64_t max_num_steps;      /* per output time, rejected attempts included (dopri5.py:83-88) */
  /* batch sharding (world_size 1: leave zero / NULL) */
  int32_t world_size, rank;
  mi_ode_allgather_fn allgather;
  void* allgather_user;
  double* exchange_send_dev;  /* optional caller-owned device buffers (8 doubles / world_size*8 doubles) that the */
  double* exchange_recv_dev;  /* hook gathers from / into; NULL: the handle allocates its own */
  /* tuning knobs (0 = default) */
  int32_t linear_variant;     /* 0 auto, 1 force VALU fallback, 2 force MFMA tile kernel */
  int32_t chunk_attempts;     /* attempts enqueued between host polls (0 = adaptive) */
  int32_t reserved0;          /* (was a hipGraph knob: the attempt chain is GPU-latency-bound, not launch-bound - DESIGN.md) */
  int32_t profile;            /* 1: bracket the stage kernels of every attempt with HIP events (mi_ode_get_profile) */
  int32_t fusion;             /* 0 auto, 1 one kernel per RK stage (34 planes/attempt), 2 whole attempt in one kernel
                                 (4 planes/attempt; row-local RHS keep k_2..k_S on chip; single rank: the controller
                                 runs in the kernel's last workgroup), 3 as 2 but with the controller as its own launch,
                                 4 whole integration in ONE launch (tiny row-local systems, single rank; auto picks it
                                 when every workgroup can be co-resident; mi_ode_integrate only) */
  int32_t reserved;
  /* Optional cross-rank hand-off memory for the whole-call kernels (world_size > 1): a HOST memory segment shared by
   * all ranks of the node (e.g. a /dev/shm mapping), at least mi_ode_xrank_bytes(world_size) bytes, zero-filled before the
   * first use.  The handle registers it with HIP; its kernels then exchange their per-attempt records through it
   * (system-scope stores / loads over PCIe or xGMI, no host involvement, no collective launch), so a sharded run keeps
   * the one-launch-per-call schedule.  The allgather hook stays the fallback (mi_ode_xrank_enable). */
  void* xrank_host;
  int64_t xrank_bytes;
  /* Tuple states (odeint.py:28-81: y0 may be a tuple of tensors; the RHS is applied to every component).  The components -
   * each [rows_k, dim] - travel in ONE buffer with a segment table: component k starts at a multiple of MI_ODE_SEGMENT_ALIGN
   * rows (the caller pads; padding rows are never read or written) and has seg_rows[k] rows; `batch` = the padded total.  The
   * scalar decisions run per component exactly as in the reference: one error ratio each (misc.py:250-264), accepted if all
   * are <= 1 (dopri5.py:108), python max() of them for the next step (misc.py:270) and over the per-component norms of the
   * initial step (misc.py:227-245); with the tsit5 controller ONE ratio pooled over all components (tsit5.py:126-138).
   * n_segments <= 1: a single tensor.  Row-local right-hand sides (catalogue and plugins), adaptive tableaus, one rank,
   * whole-call schedule. */
  int32_t n_segments;
  int32_t seg_tolerances;     /* 1: seg_rtol / seg_atol hold one pair per component (dopri5.py:60-61 accepts lists); 0: rtol / atol for all */
  int64_t seg_rows[MI_ODE_MAX_SEGMENTS];
  double seg_rtol[MI_ODE_MAX_SEGMENTS], seg_atol[MI_ODE_MAX_SEGMENTS];   /* (the initial step uses rtol / atol = the first pair, dopri5.py:74) */
  /* Fixed-grid multistep solvers (adaptive = 0; the tableau is ignored): the reference's AdamsBashforth / AdamsBashforthMoulton
   * (fixed_adams.py:152-212) as ONE launch per mi_ode_fixed_grid_integrate[_on] call, for the row-local catalogue systems.
   * ms_ab / ms_am / ms_am0 are HOST arrays, copied at create, holding the coefficients as the reference forms them in Python
   * floats: ms_ab[o * 12 + j] = (1 / DIVISOR[o]) * BASHFORTH[o][j], ms_am[o * 12 + j] = (1 / DIVISOR[o + 1]) * MOULTON[o + 1][j + 1],
   * ms_am0[o] = MOULTON[o + 1][0] / DIVISOR[o + 1], o = 0 .. 12 (rows of unused orders: zeros).  rtol / atol: the corrector's
   * convergence test (misc.py:129-134; odeint passes ITS rtol / atol).  stats.n_rejected = steps whose corrector did not converge
   * (the reference prints a warning for each and drops its oldest history entry, fixed_adams.py:197-200). */
  int32_t multistep;          /* 0: none, 1: Adams-Bashforth ('explicit_adams'), 2: Adams-Bashforth-Moulton ('fixed_adams'),
                               * 3: the variable-step, variable-order Adams solver ('adams', adams.py:66-211; ABI 10): adaptive = 1, the whole
                               * mi_ode_integrate call is ONE launch for the row-local catalogue systems (csrc/mi_ode_adams_vc.h); uses
                               * rtol / atol / safety / ifactor / dfactor, ms_max_order (<= 12) and ms_gamma_star; the tableau is ignored */
  int32_t ms_max_order;       /* <= 12 (fixed_adams.py:89) */
  int32_t ms_max_iters;       /* corrector iterations (fixed_adams.py:90: 4) */
  int32_t ms_min_order;       /* below it the start-up RK4 3/8 step runs (fixed_adams.py:88: 4) */
  const double* ms_ab;
  const double* ms_am;
  const double* ms_am0;
  const double* ms_gamma_star; /* multistep = 3: HOST array of 13 doubles (adams.py:15-18), copied at create */
} mi_ode_desc;
#define MI_ODE_SEGMENT_ALIGN 256

typedef struct mi_ode_stats {
  int64_t n_attempts, n_accepted, n_rejected, nfe;
  double t, dt;               /* rk_state.t1 and the next step size */
  double last_ratio;          /* last mean_sq_error_ratio */
  uint32_t status;            /* MI_ODE_ST_* bits */
  int32_t n_polls;            /* host synchronisations taken */
  int64_t n_launches;         /* kernels enqueued */
  double clock_mhz;           /* shader clock the one-launch kernels of the call ran at (their own cycle counter against the
                                 100 MHz constant clock); 0 where not measured */
  double handoff_us;          /* ABI 13: time workgroup 0 of the whole-call linear tile kernel spent in the call's grid hand-offs (the
                                 cross-rank exchange of a sharded run included); 0 where not measured */
} mi_ode_stats;

typedef struct mi_ode_solver* mi_ode_handle;

/* ---- library ------------------------------------------------------------------------------ */
int mi_ode_abi_version(void);
const char* mi_ode_status_string(uint32_t status_bits);   /* reference assertion text for the first set bit */
const char* mi_ode_last_error(void);                      /* thread-local text of the last negative return */
int64_t mi_ode_reduce_workspace_bytes(void);              /* scratch the stateless reductions need */
int64_t mi_ode_sizeof(int32_t which);                     /* 0: mi_ode_desc, 1: mi_ode_stats, 2: mi_ode_tableau, 3: mi_ode_rhs,
                                                             5: mi_ode_ctrl_params, 6: mi_ode_adjoint_desc, 7: mi_ode_opq_desc, 8: mi_ode_linadj_desc
                                                             (lets a foreign-language binding verify its struct layout) */

/* ---- (A) fused engine ---------------------------------------------------------------------- */
int mi_ode_create(const mi_ode_desc* desc, mi_ode_handle* out);
int mi_ode_destroy(mi_ode_handle h);

/* Solver.before_integrate(t): load y0, evaluate f0, pick the first step (dopri5.py:70-79). */
int mi_ode_begin(mi_ode_handle h, const void* y0_dev, double t0, void* stream);
/* Solver.advance(next_t) for n_out strictly increasing times (dopri5.py:81-89); out_dev: [n_out, batch*dim].
 * Blocks until the outputs are produced or a status bit is raised; returns status bits (>=0) or an error (<0). */
int mi_ode_advance(mi_ode_handle h, const double* t_out_host, int32_t n_out, void* out_dev, void* stream);
/* AdaptiveStepsizeODESolver.integrate(t) (solvers.py:27-35): out_dev [T, batch*dim], out[0] = y0. */
int mi_ode_integrate(mi_ode_handle h, const void* y0_dev, const double* t_host, int32_t T, void* out_dev,
                     mi_ode_stats* stats, void* stream);
/* FixedGridODESolver.integrate(t) with grid = t (solvers.py:82-104); no host synchronisation inside. */
int mi_ode_fixed_grid_integrate(mi_ode_handle h, const void* y0_dev, const double* t_host, int32_t T,
                                void* out_dev, mi_ode_stats* stats, void* stream);
/* The same on a time grid of its own (FixedGridODESolver(step_size=... | grid_constructor=..., eps=...), solvers.py:41-56, 86-100):
 * steps are taken on grid_host[0..G-1] (grid[0] == t[0], grid[G-1] == t[T-1]); a requested time that is not a grid point is
 * linearly interpolated inside the step that reaches it (_linear_interp, solvers.py:106-115); `eps` is added to the time the
 * step function evaluates f at (fixed_grid.py:7, 42).  One launch; row-local and MFMA-linear RHS families. */
int mi_ode_fixed_grid_integrate_on(mi_ode_handle h, const void* y0_dev, const double* grid_host, int32_t G,
                                   const double* t_host, int32_t T, double eps, void* out_dev, mi_ode_stats* stats, void* stream);
/* One attempt with a given dt - the parity surface mirroring _runge_kutta_step (rk_common.py:22-61):
 * y1/f1/err_norms outputs are device pointers (nullable); err_norms = {max|y0|, max|y1|, sum err^2, nonfinite}.
 * k_out (nullable): [S+1, batch*dim] stage derivatives. */
int mi_ode_rk_step_fused(mi_ode_handle h, const void* y0_dev, const void* f0_dev, double t0, double dt,
                         void* y1_dev, void* f1_dev, double* err_norms_host, void* k_out_dev, void* stream);
/* f(t, y) through the fused kernels (y0 -> f0), e.g. to seed mi_ode_rk_step_fused. */
int mi_ode_eval_rhs(mi_ode_handle h, const void* y_dev, double t, void* f_dev, void* stream);
int mi_ode_get_stats(mi_ode_handle h, mi_ode_stats* stats, void* stream);
/* Kernel timing collected with hipEvents on the launch stream when desc.profile = 1 (real attempts only):
 * out[0] = sum of last-stage (stage+error kernel) durations [ms], out[1] = number of those launches,
 * out[2] = sum over attempts of the span first-stage-start .. last-stage-end [ms], out[3] = attempts counted. */
int mi_ode_get_profile(mi_ode_handle h, double* out4);
/* current rk_state: y1, f1 (device, nullable). */
int mi_ode_get_state(mi_ode_handle h, void* y_dev, void* f_dev, void* stream);

/* Cross-rank hand-off (mi_ode_desc.xrank_host).  mi_ode_xrank_selftest: every rank of the group calls it at the same
 * point; a few hand-off rounds run through the shared segment with a bounded wait; returns 0 when this rank saw every
 * peer's records intact.  The caller combines the verdicts (e.g. all-reduce MIN) and calls mi_ode_xrank_enable(h, 1) on
 * every rank only if all passed; otherwise the allgather hook keeps being used. */
int64_t mi_ode_xrank_bytes(int32_t world_size);
int mi_ode_xrank_selftest(mi_ode_handle h, void* stream);
int mi_ode_xrank_enable(mi_ode_handle h, int32_t on);

/* The same hand-off through PEER DEVICE MEMORY - the transport for the GPUs of one xGMI node (the only
 * cross-rank traffic of the path is one 6-double record per rank per step attempt; no reference counterpart, the
 * reference is single-device).  Every rank owns a mailbox in its own HBM:
 *   mi_ode_xpeer_prepare  allocates it (uncached device memory, mi_ode_xrank_bytes(world_size) bytes, zeroed) and returns
 *                         its hipIpcMemHandle_t (MI_ODE_IPC_HANDLE_BYTES bytes) for the caller to all-gather over any
 *                         channel it has (torch.distributed in the Python shim, MPI, a file ...);
 *   mi_ode_xpeer_connect  takes the handles of ALL ranks in rank order, maps the peers' mailboxes into this process
 *                         (hipIpcOpenMemHandle; entry `rank` is this rank's own allocation) and hands the pointer table
 *                         to the kernels.  From then on mi_ode_xrank_selftest / mi_ode_xrank_enable use this transport
 *                         (it takes precedence over xrank_host): each rank's gateway lane q stores the rank record
 *                         into rank q's mailbox over the link to q and polls only its own, local mailbox.
 * Both return 0 or a negative error; a failure leaves the handle on its previous transport. */
#define MI_ODE_IPC_HANDLE_BYTES 64
int mi_ode_xpeer_prepare(mi_ode_handle h, void* ipc_handle_out);
int mi_ode_xpeer_connect(mi_ode_handle h, const void* all_ipc_handles, int32_t world_size);

/* RCCL from inside the library, for the launch-per-attempt schedule (ranks on several nodes, or no usable mailbox):
 * the per-attempt record exchange becomes ncclAllGather(send, recv, 8, ncclDouble, comm, stream) enqueued by
 * libmi_ode itself between the attempt kernel and the controller kernel - no callback into the host language.
 * librccl is taken from the process if it is already loaded (PyTorch-ROCm ships one), else dlopen'ed.
 *   mi_ode_rccl_unique_id   ncclGetUniqueId into id_out (MI_ODE_RCCL_ID_BYTES bytes); rank 0 calls it and broadcasts the bytes;
 *   mi_ode_rccl_connect     ncclCommInitRank(world_size, id, rank) (collective over the ranks); the communicator belongs to the
 *                           handle and is destroyed with it.  Takes precedence over mi_ode_desc.allgather.  id == NULL: drop the
 *                           communicator again (when some rank could not join, every rank has to fall back together). */
#define MI_ODE_RCCL_ID_BYTES 128
int mi_ode_rccl_unique_id(void* id_out);
int mi_ode_rccl_connect(mi_ode_handle h, const void* id, int32_t world_size, int32_t rank);

/* ---- (A') fused backward segment of odeint_adjoint --------------------------------------------------------------- */
/* tfdiffeq/adjoint.py:57-178: the reference's backward pass calls odeint on the augmented tuple state
 *     (y, adj_y, adj_t, adj_params)   with dynamics   (f, -adj_y^T df/dy, -adj_y^T df/dt, -adj_y^T df/dparams)
 * over [t_i, t_{i-1}] (adjoint.py:148-153), once per output interval.  This entry point is that call for the ODEFunc MLP
 * (MI_ODE_RHS_MLP_TANH, fp32, state [batch, dim]) in ONE launch: dopri5 over all four components, per-component error
 * ratios and their max (misc.py:250-287), initial step over all components (misc.py:183-247), dense output at t_end
 * (interp.py:6-67).  adj_params is one flat vector in CANONICAL order W1 [dim,hidden], b1, W2 [hidden,hidden], b2,
 * W3 [hidden,dim], b3 (weights [in, out] row-major as in mi_ode_rhs.w); mi_ode_adjoint_num_params() entries. */
typedef struct mi_ode_adjoint_desc {
  int64_t batch;
  int32_t dim, hidden;
  mi_ode_tableau tableau;     /* dopri5: 6 rows, FSAL shaped, c_mid filled, c_sol[1] = c_error[1] = c_mid[1] = 0 */
  double rtol, atol;          /* scalars: adjoint.py passes them through to every component (dopri5.py:60-61) */
  double safety, ifactor, dfactor;
  int32_t order, init_order;
  int64_t max_num_steps;
  int32_t time_dependent;     /* 1: the network of dense_odenet.py:79-84 with time_dependent=True - fc1 sees concat([t, x]): mi_ode_rhs.w[0]
                               * is [dim + 1, hidden] (row 0 = w_t multiplies t, mi_ode_rhs.scalars[1] != 0), adj_t has the derivative
                               * -adj_y^T df/dt, and adj_params starts with the `hidden` entries of w_t (canonical order
                               * w_t, W1 [dim,hidden], b1, W2, b2, W3, b3) */
  int32_t reserved;
} mi_ode_adjoint_desc;
typedef struct mi_ode_adjoint* mi_ode_adjoint_handle;
int mi_ode_adjoint_create(const mi_ode_adjoint_desc* desc, mi_ode_adjoint_handle* out);
int mi_ode_adjoint_destroy(mi_ode_adjoint_handle h);
int64_t mi_ode_adjoint_num_params(mi_ode_adjoint_handle h);
/* One backward interval t_start -> t_end (either direction; decreasing time is handled as misc._check_inputs does,
 * misc.py:311-321).  All state pointers are DEVICE memory, fp32; adj_t is a device scalar.  rhs: the weights (sign is ignored).
 * y_out is nullable (the reference discards it, adjoint.py:155-160).  Blocks until done; returns status bits or an error. */
int mi_ode_adjoint_segment(mi_ode_adjoint_handle h, const mi_ode_rhs* rhs, const void* y_dev, const void* adj_y_dev,
                           const void* adj_t_dev, const void* adj_params_dev, double t_start, double t_end, void* y_out_dev,
                           void* adj_y_out_dev, void* adj_t_out_dev, void* adj_params_out_dev, mi_ode_stats* stats, void* stream);
/* One evaluation of the augmented dynamics (adjoint.py:69-105), the function-level parity surface of the kernel:
 * f_out = f(y), vjp_y_out = -adj_y^T df/dy, vjp_params_out = -adj_y^T df/dparams (canonical order). */
int mi_ode_adjoint_dynamics(mi_ode_adjoint_handle h, const mi_ode_rhs* rhs, const void* y_dev, const void* adj_y_dev,
                            void* f_out_dev, void* vjp_y_out_dev, void* vjp_params_out_dev, void* stream);
/* ... at time t (the time-dependent network; mi_ode_adjoint_dynamics evaluates at t = 0).  -adj_y^T df/dt is not a separate output:
 * it is dot(w_t, b1 slice of vjp_params_out) (the identity the segment kernel integrates adj_t with). */
int mi_ode_adjoint_dynamics_at(mi_ode_adjoint_handle h, const mi_ode_rhs* rhs, double t, const void* y_dev, const void* adj_y_dev,
                               void* f_out_dev, void* vjp_y_out_dev, void* vjp_params_out_dev, void* stream);

/* ---- (A'') the linear right-hand side under odeint_adjoint -------------------------------------------------------- */
/* tfdiffeq/adjoint.py:69-105 for f(t, y) = y W + b: the augmented dynamics are (y W + b, -adj_y W^T, 0, -(y^T adj_y), -sum_rows adj_y).
 * The first two are the linear right-hand side itself (mi_ode_eval_rhs of a handle created with W, and of one created with -W^T);
 * the parameter part is a GEMM with M = N = dim and K = batch - the shape vendor BLAS serves worst (0.3 TFLOP/s fp64 in rocBLAS at
 * 65536 x 128, 7.0 ms against 63 us here) - and this entry point:
 *     out_w[dim, dim] = scale * (y^T a)   (weights [in, out], the layout of mi_ode_rhs.w)      out_b[dim] = scale * column sums of a
 * over two [batch, dim] device planes, dim <= 128, fp32 or fp64 (MFMA); out_b nullable.  Deterministic (slab partials folded in
 * slab order).  workspace_dev: mi_ode_outer_workspace_bytes(dtype, batch, dim) bytes of device scratch.  Enqueues two kernels on
 * `stream` and returns: no synchronisation, capturable in a hipGraph. */
int64_t mi_ode_outer_workspace_bytes(int32_t dtype, int64_t batch, int32_t dim);
int mi_ode_outer_reduce(int32_t dtype, int64_t batch, int32_t dim, const void* y_dev, const void* a_dev, double scale,
                        void* out_w_dev, void* out_b_dev, void* workspace_dev, void* stream);

/* ---- (A''') the backward segment of odeint_adjoint for the linear right-hand side, ONE launch (ABI 13) ------------- */
/* tfdiffeq/adjoint.py:148-160: odeint(augmented_dynamics, (y, adj_y, adj_t, adj_params), [t_i, t_{i-1}]) for f(t, y) = y W + b,
 * dopri5 over the four components with the reference's per-component error ratios, python max() of them, the initial step over all
 * components (misc.py:183-287) and dense output at t_end (interp.py:6-67).  y and adj_y run on config 4's tile kernels (W and W^T
 * resident in registers); adj_params needs ONE product over the batch per INTERVAL - y0^T a0, carried from step to step through the
 * stage polynomials of the two linear systems (csrc/mi_ode_linadj.h, docs/KERNELS.md 4e).  State dtype
 * float32 or float64, 1 <= dim <= 128.  adj_params is one flat vector: W's gradient [dim, dim] in W's own [in, out] layout, then
 * (with a bias) the dim entries of b's. */
typedef struct mi_ode_linadj_desc {
  int64_t batch;
  int32_t dim;
  int32_t dtype;              /* enum mi_ode_dtype */
  mi_ode_tableau tableau;     /* dopri5: 6 rows, FSAL shaped, c_mid filled */
  double rtol, atol;          /* scalars: adjoint.py passes them through to every component (dopri5.py:60-61) */
  double safety, ifactor, dfactor;
  int32_t order, init_order;
  int64_t max_num_steps;
} mi_ode_linadj_desc;
typedef struct mi_ode_linadj* mi_ode_linadj_handle;
int mi_ode_linadj_create(const mi_ode_linadj_desc* desc, mi_ode_linadj_handle* out);
int mi_ode_linadj_destroy(mi_ode_linadj_handle h);
/* One backward interval t_start -> t_end (either direction; decreasing time as misc._check_inputs handles it, misc.py:311-321).
 * All pointers are DEVICE memory in the state dtype; w_dev [dim, dim] row-major ([in, out]: f = y W), b_dev nullable; adj_t is a
 * device scalar.  grad_out_dev (nullable): grad_output at t_start, [batch, dim] - the segment then starts from
 * adj_t - f(t_start, y) . grad_out (the time gradient of the measurement point, adjoint.py:134-140; f(t_start, y) is the kernel's own first
 * evaluation) and stores that dot product in dldt_out_dev (nullable device scalar).  host_scalars (nullable): {the dot product,
 * adj_t(t_end)} as doubles, for a caller that assembles the time gradients on the host.  y(t_end) is not produced (the reference
 * discards it, adjoint.py:155-160).  Blocks until done; returns status bits (>= 0) or an error (< 0).  stats->n_launches == 1. */
int mi_ode_linadj_segment(mi_ode_linadj_handle h, const void* w_dev, const void* b_dev, const void* y_dev, const void* adj_y_dev,
                          const void* adj_t_dev, const void* adj_params_dev, const void* grad_out_dev, double t_start, double t_end,
                          void* adj_y_out_dev, void* adj_t_out_dev, void* adj_params_out_dev, void* dldt_out_dev, double* host_scalars,
                          mi_ode_stats* stats, void* stream);
/* where the time of the last segment went, microseconds of workgroup 0: {tile passes, adj_params combinations, hand-offs of the
 * attempts, slab passes, folds + small products with their hand-offs, prologue, epilogue, hand-offs (count)} */
int mi_ode_linadj_profile(mi_ode_linadj_handle h, double* out8);

/* ---- function-level parity surface of the step controller ----------------------------------- */
/* The scalar tail of one step attempt exactly as the kernels run it (csrc/mi_ode_ctrl_dev.h, ONE device thread per case):
 *   phase 2 (attempt): misc._compute_error_ratio's scalar part + accept test + misc._optimal_step_size / tsit5._optimal_step_size
 *                      (misc.py:256-287, tsit5.py:53-62, dopri5.py:103-121)
 *   phase 0 / 1      : the two halves of misc._select_initial_step (misc.py:227-245)
 * on caller-supplied numbers, so that the DEVICE arithmetic can be driven with the reference's own vectors.
 * in[c]  (8 doubles per case): {max|y0|, max|y1|, sum_a, sum_b, nonfinite flag, N (elements behind the sums), -, -}
 * st[c]  (4 doubles per case): {t1, dt, h0, d1}   (rk_state.t1 / rk_state.dt; h0, d1: what phase 0 produced, for phase 1)
 * out[c] (8 doubles per case): {ratio, accepted, dt_next, t1_next, t0_next, status bits, h0, d0 (phase 0) | d1 in out[c][6..7]}
 *          phase 0 writes {-, -, -, -, -, -, h0, d0} and d1 into out[c][2]; phase 1 writes the first step size into out[c][2].
 * All pointers are HOST memory; the call is synchronous. */
typedef struct mi_ode_ctrl_params {
  double rtol, atol, safety, ifactor, dfactor;   /* safety / ifactor / dfactor: the float32-rounded values (misc.py:137-144) */
  int32_t order, init_order;                     /* dopri5.py:68,74 */
  int32_t controller;                            /* enum mi_ode_controller */
  int32_t dtype;                                 /* enum mi_ode_dtype of the STATE: float32 states take the float32 detours */
} mi_ode_ctrl_params;
int mi_ode_controller_update(const mi_ode_ctrl_params* p, int32_t phase, int32_t n_cases, const double* in_host,
                             const double* st_host, double* out_host, void* stream);
/* y_sigma = y0 + add_n((dt * beta_j) * k_j)  - one stage combination of rk_common._runge_kutta_step (rk_common.py:51);
 * the same kernel as mi_ode_lincomb, under the name SURVEY.md 8(b) gives it */
int mi_ode_rk_stage_combine(int32_t dtype, int64_t n, const void* y0_dev, const void* const* k_dev, const double* beta_row,
                            int32_t n_k, double dt, void* out_dev, void* stream);
/* {max|y0|, max|y1|, sum err^2, nonfinite} of one attempt (misc.py:256-263): mi_ode_error_norms under its 8(b) name */
int mi_ode_rk_error_reduce(int32_t dtype, int64_t n, const void* err_dev, const void* y0_dev, const void* y1_dev,
                           double* result_dev, void* workspace_dev, void* stream);

/* ---- (C) opaque right-hand side, device-resident controller ------------------------------------------------ */
/* The reference's own call shape - odeint(func, y0, t) with a Python callable (tests/odeint_tests.py:30-77,
 * examples/ode_demo.py:39,169) - costs one host decision per attempt when the controller runs on the host
 * (dopri5.py:103-121 reads the error ratio back).  Here the scalar state of the solver (rk_state.t0 / t1 / dt, the output
 * cursor, the counters, the reference's assertions as status bits) lives in a device record; the caller evaluates f between
 * our launches, the stage combinations are mi_ode_lincomb_dev with scale_dev = mi_ode_opq_dt_dev(h), and f receives its time
 * argument as a 0-d view of the stage-time array - so one attempt
 *     for sigma = 1..S:  y_sigma = lincomb_dev(y0, k_1..k_sigma, beta_sigma);  k_{sigma+1} = f(ts[sigma-1], y_sigma)
 *     [not FSAL shaped: y1 = lincomb_dev(y0, k, c_sol)]                          (rk_common.py:49-56)
 *     mi_ode_opq_finish(y0, y1, k);  mi_ode_opq_commit(y0, f0, y1, k)
 * contains no host value and can be captured as a hipGraph and replayed.  A tuple state has n_comp components (all pointer
 * arrays are [n_comp], k is [n_comp][S + 1] flattened, k[c][0] = f0 of component c); ratios are per component, accepted if all
 * are <= 1, python max() of them drives the step size (misc.py:250-287); the tsit5 controller pools them (tsit5.py:126-138).
 * Every kernel is a no-op once the record says `done` (all outputs produced, or a status bit), and mi_ode_opq_commit is a
 * no-op after a rejected attempt: a host that replays attempts blindly in chunks and polls once per chunk is correct. */
typedef struct mi_ode_opq_desc {
  int32_t dtype;              /* enum mi_ode_dtype of every component */
  int32_t n_comp;             /* 1 .. MI_ODE_MAX_SEGMENTS */
  int64_t n[MI_ODE_MAX_SEGMENTS];                 /* elements per component */
  mi_ode_tableau tableau;     /* 1, 3, 6 or 13 rows (dense output is instantiated for S + 1 = 2, 4, 7, 14 stage derivatives) */
  int32_t controller, interp, order, init_order;  /* as mi_ode_desc */
  double rtol[MI_ODE_MAX_SEGMENTS], atol[MI_ODE_MAX_SEGMENTS];   /* one pair per component (dopri5.py:60-61); tsit5: pair 0 */
  double safety, ifactor, dfactor;                /* the float32-rounded values (misc.py:137-144) */
  int64_t max_num_steps;
} mi_ode_opq_desc;
typedef struct mi_ode_opq* mi_ode_opq_handle;
int mi_ode_opq_create(const mi_ode_opq_desc* desc, mi_ode_opq_handle* out);
int mi_ode_opq_destroy(mi_ode_opq_handle h);
/* device address of rk_state.dt (float64): the scale_dev of the attempt's mi_ode_lincomb_dev calls */
const double* mi_ode_opq_dt_dev(mi_ode_opq_handle h);
/* Arms an integration: rk_state = (t0, t0, first_dt) (dopri5.py:79), the output cursor over t_out_host[0 .. n_out-1]
 * (strictly increasing, all > t0; solvers.py:31-35), out_dev[c] = the [n_out, n[c]] solution rows of component c, and the
 * first attempt's stage times into stage_times_dev ([S] elements of the state dtype; rewritten after every attempt).
 * A recorded attempt (finish + commit captured in a hipGraph) stays valid across opq_begin calls: the kernels read the solution
 * rows' addresses and the output times from tables inside the handle when they run - as long as n_out never exceeds what the
 * handle has held before (1024 at creation; a larger n_out moves the output-time table, and the attempt must be recorded again). */
int mi_ode_opq_begin(mi_ode_opq_handle h, double t0, double first_dt, const double* t_out_host, int32_t n_out,
                     void* const* out_dev, void* stage_times_dev, void* stream);
int mi_ode_opq_finish(mi_ode_opq_handle h, const void* const* y0_dev, const void* const* y1_dev, const void* const* k_dev,
                      void* stream);
int mi_ode_opq_commit(mi_ode_opq_handle h, void* const* y0_dev, void* const* f0_dev, const void* const* y1_dev,
                      const void* const* k_dev, void* stream);
/* Reads the scalar state back (one stream synchronisation): *done, the counters, status bits (also the return value). */
int mi_ode_opq_poll(mi_ode_opq_handle h, mi_ode_stats* stats, int32_t* done, void* stream);

/* ---- (B) stateless plane kernels (arbitrary Python f, tuple states) ------------------------- */
/* out[i] = (base ? base[i] : 0) + sum_j (scale * coef[j]) * xs[j][i]       (misc.py:118-121; zeros not skipped) */
int mi_ode_lincomb(int32_t dtype, int64_t n, const void* base_dev, const void* const* xs_dev, const double* coef,
                   int32_t nx, double scale, void* out_dev, void* stream);
/* same with the scale (dt) read from device memory at execution time: what a captured hipGraph of an RK attempt needs
 * (the graph is replayed with a new dt without re-recording) */
int mi_ode_lincomb_dev(int32_t dtype, int64_t n, const void* base_dev, const void* const* xs_dev, const double* coef,
                       int32_t nx, const double* scale_dev, void* out_dev, void* stream);
/* result_dev[4] (double) = {max|y0|, max|y1|, sum err^2, nonfinite(y0)}            (misc.py:256-263) */
int mi_ode_error_norms(int32_t dtype, int64_t n, const void* err_dev, const void* y0_dev, const void* y1_dev,
                       double* result_dev, void* workspace_dev, void* stream);
/* result_dev[1] (double) = sum_i ((x[i] - (xsub ? xsub[i] : 0)) / (atol + |y0[i]| * rtol))^2   (misc.py:225-237) */
int mi_ode_scaled_sumsq(int32_t dtype, int64_t n, const void* x_dev, const void* xsub_dev, const void* y0_dev,
                        double rtol, double atol, double* result_dev, void* workspace_dev, void* stream);
/* The variable-order Adams solver's host loop (tfdiffeq/adams.py:134-210, any callable f) on four plane kernels instead of one linear
 * combination at a time.  phi: the implicit-phi planes, newest first (`order` of them); g: the values of the float32 g vector,
 * g[0..order]; beta[0..order-1] (beta[0] unused); all planes hold n elements of `dtype`.
 *   predict     p = y + dt * sum_{j < max(1, order-1)} g_j explicit_phi_j,  explicit_phi_0 = phi_0, explicit_phi_j = beta_j phi_j
 *   correct     ip_0 = f_p, ip_j = ip_{j-1} - explicit_phi_{j-1};  y_next = p + dt g_{order-1} ip_{order-1};  also returns the planes
 *               ip_order, ip_{order-1}, ip_{order-2} (nullable) and result[4] = {max|y|, max|y_next|, sum local_error^2, nonfinite(y)}
 *   error_sums  result[2] = sums of ((dt * coef) x / tol)^2 for one or two planes (xb nullable): the error ratios times n
 *   update_phi  new_0 = f_new, new_j = new_{j-1} - explicit_phi_{j-1}, j <= order  (order + 1 output planes) */
int mi_ode_adams_predict(int32_t dtype, int64_t n, const void* y_dev, const void* const* phi_dev, int32_t order, const double* g,
                         const double* beta, double dt, void* p_out_dev, void* stream);
int mi_ode_adams_correct(int32_t dtype, int64_t n, const void* y_dev, const void* p_dev, const void* f_p_dev, const void* const* phi_dev,
                         int32_t order, const double* g, const double* beta, double dt, void* y_next_out_dev, void* ip_k_out_dev,
                         void* ip_k1_out_dev, void* ip_k2_out_dev, double* result_dev, void* workspace_dev, void* stream);
int mi_ode_adams_error_sums(int32_t dtype, int64_t n, const void* xa_dev, double coef_a, const void* xb_dev, double coef_b, double dt,
                            double tol, double* result_dev, void* workspace_dev, void* stream);
int mi_ode_adams_update_phi(int32_t dtype, int64_t n, const void* f_new_dev, const void* const* phi_dev, int32_t order,
                            const double* beta, void* const* new_phi_out_dev, void* stream);
/* result_dev[1] (double) = 1.0 if ANY element violates |a - b| < atol + rtol * max(|a|, |b|), else 0.0
 * (misc._has_converged, misc.py:129-134: the corrector iteration test of fixed_adams.py:196) */
int mi_ode_not_converged(int32_t dtype, int64_t n, const void* a_dev, const void* b_dev, double rtol, double atol,
                         double* result_dev, void* workspace_dev, void* stream);
/* dense output at one time t in [t0, t1] from the accepted step's (y0, y1, k[0..nk-1]):
 * interp = QUARTIC_MID: interp.py:6-67 with y_mid = y0 + dt*sum c_mid[j] k[j] (dt = the step size the step was
 * taken with, dopri5.py:41); TSIT5 / TSIT5_REF: tsit5.py:33-50 */
int mi_ode_interp_eval(int32_t dtype, int32_t interp, int64_t n, const void* y0_dev, const void* y1_dev,
                       const void* const* ks_dev, int32_t nk, const double* c_mid, double dt, double t0, double t1,
                       double t, void* out_dev, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MI_ODE_H */
