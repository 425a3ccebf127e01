// The backward solve of odeint_adjoint for the LINEAR right-hand side f(t, y) = y W + b in ONE launch per output interval
// (include/mi_ode.h section A'''; tfdiffeq/adjoint.py:57-178 is the reference: odeint over the tuple (y, adj_y, adj_t, adj_params)
// with dynamics (f, -adj_y^T df/dy, -adj_y^T df/dt, -adj_y^T df/dparams), adjoint.py:69-105).
//
// For this right-hand side the four components are
//     y' = s (y W + b)          a' = -s a W^T          adj_t' = 0          theta' = -s [y^T a ; sum_rows a]        (s = -1: decreasing time)
// Structure of the kernel - a persistent grid of G workgroups, one per CU, hand-offs as in mi_ode_persist.h:
//   * y and a are two linear systems: every workgroup runs config 4's tile pass (lin_attempt_pass, mi_ode_step_fused.h) over its tiles
//     of the y planes with its slice of W resident, reloads the slice of W^T (sign -s, no bias) and runs it over the same tiles of the a
//     planes - so the rows of (y1, a1) a workgroup needs for the slab product below are rows it wrote itself;
//   * theta needs NO per-stage product over the batch.  Both systems have constant matrices, so every stage input is the step's start
//     value times a polynomial in h W whose coefficients depend on the tableau only (oracle/linear_adjoint_numpy.py, "the power form"):
//         sum_sigma (h c_sigma) k_theta,sigma = -s h sum_pq K^c_pq (s h)^p (-s h)^q M_pq,    M_pq = ((W^T)^p G0 + c_p g0) (W^T)^q
//     with G0 = y0^T a0, g0 = sum_rows a0 of the step's START state, c_p = (W^T)^(p-1) b^T and K^c = sum_sigma c_sigma pi_sigma pi_sigma^T
//     (tableau constants, one table per combination: solution, error estimate, y_mid, the last stage).  Per ACCEPTED step: one slab
//     pass over the two planes (G0, g0: the only batch-sized product), its fold, (S+1) + (S+1)^2 products of dim x dim matrices spread
//     over the grid's wavefronts; per ATTEMPT: an elementwise combination of the M_pq, each workgroup its share of the entries;
//   * adj_t has a zero derivative (f does not depend on t): its error estimate is 0, its norms enter the controller all the same;
//   * the controller runs redundantly in every workgroup over the four components (misc.py:250-287: per-component ratios, python
//     max(); misc.py:183-247 for the initial step, where h0 = +inf is the NORMAL case of this tuple - see mi_ode_adjoint.h);
//   * adj_y(t_end) is the tile pass's speculative dense output; theta(t_end) is ONE more combination with the folded dense-output
//     weights (oracle: dense_output_fold_weights); adj_t(t_end) the fit of a constant evaluated as the reference does.
// The small products and theta are held in float64 whatever the state dtype (they are O(dim^2): their cost is hand-off latency, not
// arithmetic); the slab product runs in the state dtype like the reference's own y^T a.
// Cross-workgroup data (slab partials, G0, the small matrices): plain stores -> agent-scope release by one lane -> the hand-off record
// -> agent-scope acquire by one lane per workgroup -> plain loads (MI355X_MICROARCH.md, "inter-workgroup visibility", valid form 1).
#pragma once
#include "mi_ode_persist.h"
#ifndef MI_LA_CHK
#define MI_LA_CHK 16      // operand pairs of a small product in flight at once (32 - every operand of a dim-128 product, 48 loads - gave WRONG products in this kernel: stale accumulator rows; 8 and 16 are right)
#endif

namespace mi {

constexpr int kLaS = 6;                  // dopri5 (adjoint.py's default; other tableaus take the generic path)
constexpr int kLaP = kLaS + 1;           // powers 0 .. S of W^T
constexpr int kLaPP = kLaP * kLaP;
constexpr int kLaMaxG = 256;             // workgroups (one per CU)

struct LinAdjResult {                    // pinned host: the kernel's last act is a zero-copy store of this record
  double t1, dt, ratio, h0;
  long long n_attempt, n_accept;
  unsigned status;
  int handoffs;
  long long prof[16];                    // 10 ns ticks of workgroup 0: tile passes / theta combinations / hand-offs of attempts / slab passes /
                                         // folds + small products (with their hand-offs) / prologue / epilogue / -
  long long clk_cycles, clk_ticks;
  double dldt, adjt_end;                 // f(t_start, y) . grad_out (0 without grad_out) and adj_t(t_end), for the host's time gradients
};

struct LinAdjArgs {
  PersistArgs p;                         // p.s: tableau, rhs (W, b, sign = s), controller parameters, partials; p.s.out = adj_y(t_end)
  const void* y_in;                      // state at t_start: y, adj_y [batch, dim]; adj_t scalar; adj_params [dim * dim (+ dim)]
  const void* a_in;
  const void* th_in;
  const void* adjt_in;
  const void* grad_in;                   // nullable: grad_output at t_start [batch, dim] - adj_t starts at adjt_in - f(t_start, y) . grad_in (adjoint.py:134-140)
  void* dldt_out;                        // nullable: that dot product
  void* th_out;
  void* adjt_out;
  char* planes;                          // 8 planes of batch * dim elements: y_a, y_b, fy_a, fy_b, a_a, a_b, fa_a, fa_b
  long long stride;
  double* pw;                            // [kLaP][D][D]       (W^T)^q, zero padded to the tile width (P_0 = identity)
  double* cvec;                          // [kLaP][D]          c_p = (W^T)^(p-1) b^T, c_0 = 0
  void* gpart;                           // [G][D * D + D]     slab partials of (G0 | g0), state dtype
  double* g0;                            // [2][D * D + D]     G0 | g0 of the current step's start state / of the attempt's end state
  double* lmat;                          // [kLaP][D * D]      L_p = (W^T)^p G0 + c_p g0  (p >= 1; L_0 = G0 is read in place)
  double* mmat;                          // [kLaPP][(D + 1) D] M_pq; row D: g0 (W^T)^q for p = 0, zero otherwise (the bias entries)
  double* theta;                         // [2][D * D + D]     adj_params at the step's start / after the attempt, padded layout
  void* wpad;                            // [2][D][D]          W and W^T zero padded to the tile width, state dtype (what the tile passes keep resident)
  const double* ktab;                    // device [4][kLaPP]: K^sol, K^err, K^mid, pi_S pi_S^T
  double t_end;
  LinAdjResult* res;
  int has_bias;
  long long* skew;                       // device [32][G][2] or null (MI_ODE_LINADJ_PROF): every workgroup's arrival at / departure from the first 32 hand-offs
  int dbg;                               // tuning / bisecting aid (MI_ODE_LINADJ_DBG): bit 0: the chain of small products right after the accept, not between the passes
};
static_assert(sizeof(LinAdjArgs) <= 4096, "kernel arguments");

template <int MAXG>
struct LaShared {
  Ctl c;
  PersistPub pub;
  AttemptState st;
  double red[80];
  double coef[kLinCoefMax];
  double vals[8][MAXG];                  // every workgroup's record, staged for the fixed-order folds
  double tout[kPersistTSmall];
  double seg_rec[4][kRec];               // combined records of (y, adj_y, adj_t, adj_params)
  SegState seg;
  double kq[3][kLaPP];                   // the attempt's combination weights of the M_pq: solution, error estimate, G0 | g0 of the end state (or: dense output)
  double adjt;                           // adj_t (constant over the segment)
  double th0_max;                        // max |adj_params| at the step's start (thread 0)
  double dldt, adjt_end;
  double mine[8];                        // thread 0: this workgroup's record while its passes run
  long long prof[16], tk_prev;            // workgroup 0: where the time of a segment goes (LinAdjResult.prof)
  int ok;
  int skip_initb;
};

// ---- 8-value hand-off records (kPRec = 16 words: all of a record) -------------------------------------------------------------------
__device__ __forceinline__ bool load_record8_sc1(const double* p, unsigned seq, double (&val)[8]) {
  u64x2_t v[8];
  asm volatile(
      "global_load_dwordx4 %0, %8, off sc1\n\t"
      "global_load_dwordx4 %1, %8, off offset:16 sc1\n\t"
      "global_load_dwordx4 %2, %8, off offset:32 sc1\n\t"
      "global_load_dwordx4 %3, %8, off offset:48 sc1\n\t"
      "global_load_dwordx4 %4, %8, off offset:64 sc1\n\t"
      "global_load_dwordx4 %5, %8, off offset:80 sc1\n\t"
      "global_load_dwordx4 %6, %8, off offset:96 sc1\n\t"
      "global_load_dwordx4 %7, %8, off offset:112 sc1\n\t"
      "s_waitcnt vmcnt(0)"
      : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7])
      : "v"(p)
      : "memory");
  bool ok = true;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    ok = ok && ((unsigned)(v[i].x >> 32) == seq) && ((unsigned)(v[i].y >> 32) == seq);
    val[i] = __longlong_as_double((long long)((v[i].x & 0xffffffffull) | (v[i].y << 32)));
  }
  return ok;
}

// Cross-workgroup data of the small products (G0 | g0, L_p, M_pq: kilobytes per workgroup) is stored WRITE-THROUGH (sc1): it is in memory
// when the storing wavefront's vmcnt drains, and the publishing side needs no write-back of its XCD's L2 - which, behind a tile pass,
// holds megabytes of dirty state planes (a `buffer_wbl2` there costs the hand-off several microseconds for a few kilobytes of payload).
__device__ __forceinline__ void st_wt(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// Grid hand-off in two halves: thread 0 publishes `mine` (la_publish), every workgroup gathers all records into sh.vals (la_gather).
// RELEASE: this workgroup wrote data other workgroups read after the hand-off; ACQUIRE: it reads such data.
__device__ __forceinline__ void la_publish(const PersistArgs& P, unsigned gen, const double (&mine)[8], bool release) {
  const int G = (int)gridDim.x;
  double* buf = P.s.partials + (long long)(gen & 1u) * G * kPRec;
  const unsigned seq = P.seq_base + gen + 1u;
  // EVERY wavefront drains its own stores before the barrier: a workgroup-scope fence (what __syncthreads() carries) does not wait for
  // global stores on this target (one CU, one L1: nothing to wait for at THAT scope), and a line that reaches the L2 after thread 0's
  // write-back below stays dirty there - invisible to the other XCDs until it is evicted (observed: stale G0 rows, two steps old)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    if (release) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // (the compiler may drop the wait behind the write-back: MI355X_MICROARCH.md)
    double* rec = buf + (long long)blockIdx.x * kPRec;
#pragma unroll
    for (int i = 0; i < 8; ++i) store_ll_sc1(rec + 2 * i, mine[i], seq);
  }
}
// Returns false (to every thread) on a time-out.
template <class SH>
__device__ __forceinline__ bool la_gather(const PersistArgs& P, SH& sh, unsigned gen, bool acquire) {
  const int G = (int)gridDim.x;
  const double* buf = P.s.partials + (long long)(gen & 1u) * G * kPRec;
  const unsigned seq = P.seq_base + gen + 1u;
  const int limit = gen == 0 ? P.spin_first : P.spin_limit;
  for (int b = threadIdx.x; b < G; b += blockDim.x) {
    const double* p = buf + (long long)b * kPRec;
    double v[8];
    int spins = 0;
    for (int i = 0; i < P.sleep_first; ++i) __builtin_amdgcn_s_sleep(1);
    for (;;) {
      if (load_record8_sc1(p, seq, v)) break;
      for (int i = 0; i < P.sleep_poll; ++i) __builtin_amdgcn_s_sleep(1);
      if (++spins > limit) { sh.ok = 0; break; }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) sh.vals[i][b] = v[i];
  }
  if (acquire) {                                                // the invalidate must come after the LAST record was seen (any wavefront's)
    __syncthreads();
    if (threadIdx.x == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");   // this CU's L1 forgets what it held
  }
  __syncthreads();
  return sh.ok != 0;
}
template <class SH>
__device__ __forceinline__ bool la_exchange(const PersistArgs& P, SH& sh, unsigned gen, const double (&mine)[8], bool release, bool acquire) {
  la_publish(P, gen, mine, release);
  return la_gather(P, sh, gen, acquire);
}

// "My share of the `cgen`-th flag round is written" (write-through stores, drained): one stamped 16-byte word per workgroup and parity of
// cgen, behind the grid records in the same allocation; waited for with an acquire.  Every workgroup takes part in every round, in order:
// who has seen all flags of round r knows that nobody still polls round r - 1, whose words round r + 1 overwrites.  A workgroup
// sets its word, goes through a whole tile pass and only then waits for everybody's - by then they have long been set: the wait is one
// poll round, not a rendez-vous (and a workgroup that IS late is simply waited for).
__device__ __forceinline__ double* la_flag_words(const PersistArgs& P, int which, unsigned cgen) {
  return P.s.partials + 2 * kLaMaxG * kPRec + (long long)((which * 2 + (int)(cgen & 1u)) * kLaMaxG) * 2;
}
__device__ __forceinline__ void la_flag_set(const PersistArgs& P, int which, unsigned cgen) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // every wavefront's write-through stores have landed
  __syncthreads();
  if (threadIdx.x == 0) store_ll_sc1(la_flag_words(P, which, cgen) + 2 * blockIdx.x, 0.0, P.seq_base + cgen + 1u);
}
template <class SH>
__device__ __forceinline__ bool la_flag_wait(const PersistArgs& P, SH& sh, int which, unsigned cgen) {
  const unsigned seq = P.seq_base + cgen + 1u;
  const double* w = la_flag_words(P, which, cgen);
  for (int b = threadIdx.x; b < (int)gridDim.x; b += blockDim.x) {
    double v;
    int spins = 0;
    while (!load_ll_sc1(w + 2 * b, seq, v)) {
      __builtin_amdgcn_s_sleep(2);
      if (++spins > P.spin_limit) { sh.ok = 0; break; }
    }
  }
  __syncthreads();                                              // (the invalidate after the last flag was seen)
  if (threadIdx.x == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  __syncthreads();
  return sh.ok != 0;
}

// wavefront 0: fixed-order fold of one record slot over workgroups [b0, b1); the result is valid in lane 0
__device__ __forceinline__ double la_fold_sum(const double* v, int b0, int b1) {
  double s = 0.0;
  for (int b = b0 + (int)threadIdx.x; b < b1; b += 64) s += v[b];
  return wave_sum(s);
}
__device__ __forceinline__ double la_fold_max(const double* v, int b0, int b1) {
  double s = 0.0;
  for (int b = b0 + (int)threadIdx.x; b < b1; b += 64) s = fmax(s, v[b]);
  return wave_max(s);
}

// ---- small products: one wavefront = one 16 x 16 tile of C = A B (+ u v^T), float64, K = D, row-major -------------------------------
// Lane (li, lg) feeds A[16 tm + li][lg KS + m] and B[lg KS + m][16 tn + li] at step m (any k-permutation is legal as long as both
// operands agree; this one makes a lane's A elements contiguous).
template <int D>
__device__ __forceinline__ void la_tile_job(const double* __restrict__ A, const double* __restrict__ B, double* __restrict__ C, int ldc,
                                            int tm, int tn, const double* __restrict__ u, const double* __restrict__ v) {
  using TR = MfmaTraits<double>;
  constexpr int KS = D / 4, CHK = (MI_LA_CHK) < KS ? (MI_LA_CHK) : KS;
  const int lane = threadIdx.x & 63, li = lane & 15, lg = lane >> 4;
  TR::acc_t acc = {0, 0, 0, 0};
  const double* ap = A + (long long)(16 * tm + li) * D + lg * KS;
  const double* bp = B + (long long)(lg * KS) * D + 16 * tn + li;
#pragma unroll
  for (int m0 = 0; m0 < KS; m0 += CHK) {
    double av[CHK], bv[CHK];
#pragma unroll
    for (int m = 0; m < CHK; ++m) { av[m] = ap[m0 + m]; bv[m] = bp[(long long)(m0 + m) * D]; }
    __builtin_amdgcn_sched_barrier(0);                         // all loads first, ONE wait, then the chain (the scheduler would interleave a wait per step)
#pragma unroll
    for (int m = 0; m < CHK; ++m) acc = TR::mfma(av[m], bv[m], acc);
  }
  const int col = 16 * tn + li;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = 16 * tm + TR::acc_row(lane, i);
    double c = acc[i];
    if (u != nullptr) c = c + u[row] * v[col];
    st_wt(C + (long long)row * ldc + col, c);
  }
}

// The same tile by KSPLIT wavefronts of ONE workgroup, wavefront `part` taking the k-steps [part KS / KSPLIT, (part + 1) KS / KSPLIT): a
// quarter of the operand loads per wavefront, all in flight at once - ONE memory round trip where the whole-K job needs two (its operands
// are freshly written by other XCDs: every load goes to memory).  The partial tiles meet in LDS ([slot][part][64 lanes][4]), part 0 adds
// them in a fixed order and stores.  Called by every wavefront of the workgroup together (`active`: this slot has a job this round).
template <int D, int KSPLIT>
__device__ __forceinline__ void la_tile_job_split(const double* __restrict__ A, const double* __restrict__ B, double* __restrict__ C, int ldc,
                                                  int tm, int tn, const double* __restrict__ u, const double* __restrict__ v, bool active,
                                                  int slot, int part, double* lds) {
  using TR = MfmaTraits<double>;
  constexpr int KS = D / 4, KQ = KS / KSPLIT;
  static_assert(KS % KSPLIT == 0 && KQ >= 1, "k-steps per wavefront");
  const int lane = threadIdx.x & 63, li = lane & 15, lg = lane >> 4;
  TR::acc_t acc = {0, 0, 0, 0};
  if (active) {
    const double* ap = A + (long long)(16 * tm + li) * D + lg * KS + part * KQ;
    const double* bp = B + (long long)(lg * KS + part * KQ) * D + 16 * tn + li;
    double av[KQ], bv[KQ];
#pragma unroll
    for (int m = 0; m < KQ; ++m) { av[m] = ap[m]; bv[m] = bp[(long long)m * D]; }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int m = 0; m < KQ; ++m) acc = TR::mfma(av[m], bv[m], acc);
    if (KSPLIT > 1 && part > 0) {
      double* o = lds + ((long long)(slot * KSPLIT + part) * 64 + lane) * 4;
#pragma unroll
      for (int i = 0; i < 4; ++i) o[i] = acc[i];
    }
  }
  if (KSPLIT > 1) __syncthreads();
  if (active && part == 0) {
    const int col = 16 * tn + li;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = 16 * tm + TR::acc_row(lane, i);
      double c = acc[i];
#pragma unroll
      for (int k = 1; k < KSPLIT; ++k) c += lds[((long long)(slot * KSPLIT + k) * 64 + lane) * 4 + i];
      if (u != nullptr) c = c + u[row] * v[col];
      st_wt(C + (long long)row * ldc + col, c);
    }
  }
  if (KSPLIT > 1) __syncthreads();                             // (the LDS slots are free for the next round)
}

// seven lanes share one entry of the folds / the combinations (49 = 7 x 7 terms; nine groups per wavefront, lane 63 idles): every lane
// of a group ends up with the group's sum, formed in a fixed order
__device__ __forceinline__ double group7_sum(double v, int lane) {
  const int base = lane / 7 * 7;
  double s_ = 0.0;
#pragma unroll
  for (int k = 0; k < 7; ++k) s_ += __shfl(v, base + k < 64 ? base + k : 63, 64);
  return s_;
}
// eight lanes share one output of the small vector jobs: every lane ends up with the group's sum
__device__ __forceinline__ double group8_sum(double v) {
  v += __shfl_xor(v, 1, 64);
  v += __shfl_xor(v, 2, 64);
  v += __shfl_xor(v, 4, 64);
  return v;
}

// ---- the slab pass: this workgroup's partial of (y^T a | sum_rows a) over ITS OWN 16-row tiles (tile = blk, blk + nblk, ...: the rows
// its tile passes wrote - no other workgroup's stores are read).  k_outer_partial's loop (mi_ode_outer.hip). ----
template <typename T, int D>
__device__ __forceinline__ void la_slab_pass(const T* __restrict__ y, const T* __restrict__ a, long long batch, int dim, int blk, int nblk,
                                             T* __restrict__ out, T* lds) {
  using TR = MfmaTraits<T>;
  using acc_t = typename TR::acc_t;
  constexpr int MB = D / 16, NT = D * 4, R = 16, EPT = R * D / NT, VEC = TR::VEC, CPT = EPT / VEC;
  using CH = Chunk<T, VEC>;
  static_assert(EPT % VEC == 0, "tile / workgroup geometry");
  constexpr int LDP = D + 16;                                  // row stride: the four row groups of an MFMA operand read land in different banks
  T* sy = lds;
  T* sa = lds + R * LDP;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, li = lane & 15, lg = lane >> 4;
  const bool vl = dim % VEC == 0 && ((((unsigned long long)y) | ((unsigned long long)a)) & 15ull) == 0;
  const long long ntiles = (batch + R - 1) / R;
  acc_t acc[MB];
#pragma unroll
  for (int m = 0; m < MB; ++m) acc[m] = acc_t{0, 0, 0, 0};
  T colsum = (T)0;
  const int ncol = 16 * w + li;
  T py[EPT], pa[EPT], qy[EPT], qa[EPT];                        // the next tile and the one after it (two tiles in flight)
  auto fetch = [&](long long tile_i, T (&py)[EPT], T (&pa)[EPT]) {
    const long long t0 = tile_i * R;
    if (vl) {
#pragma unroll
      for (int c = 0; c < CPT; ++c) {
        const int idx = (c * NT + tid) * VEC, row = idx / D, col = idx % D;
        const bool ok = t0 + row < batch && col < dim;
        CH vy, va;
#pragma unroll
        for (int v = 0; v < VEC; ++v) { vy.v[v] = (T)0; va.v[v] = (T)0; }
        if (ok) { vy = *(const CH*)(y + (t0 + row) * dim + col); va = *(const CH*)(a + (t0 + row) * dim + col); }
#pragma unroll
        for (int v = 0; v < VEC; ++v) { py[c * VEC + v] = vy.v[v]; pa[c * VEC + v] = va.v[v]; }
      }
    } else {
#pragma unroll
      for (int e = 0; e < EPT; ++e) {
        const int idx = (e / VEC * NT + tid) * VEC + e % VEC, row = idx / D, col = idx % D;   // (the same element -> LDS slot map as above)
        const bool ok = t0 + row < batch && col < dim;
        py[e] = ok ? y[(t0 + row) * dim + col] : (T)0;
        pa[e] = ok ? a[(t0 + row) * dim + col] : (T)0;
      }
    }
  };
  if ((long long)blk < ntiles) fetch(blk, py, pa);
  if ((long long)blk + nblk < ntiles) fetch((long long)blk + nblk, qy, qa);
  for (long long tile_i = blk; tile_i < ntiles; tile_i += nblk) {
    __syncthreads();                                           // (the previous tile's operands have been read)
#pragma unroll
    for (int c = 0; c < CPT; ++c) {
      CH vy, va;
#pragma unroll
      for (int v = 0; v < VEC; ++v) { vy.v[v] = py[c * VEC + v]; va.v[v] = pa[c * VEC + v]; }
      const int idx = (c * NT + tid) * VEC;
      *(CH*)(sy + idx / D * LDP + idx % D) = vy;
      *(CH*)(sa + idx / D * LDP + idx % D) = va;
    }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < EPT; ++e) { py[e] = qy[e]; pa[e] = qa[e]; }
    if (tile_i + 2 * (long long)nblk < ntiles) fetch(tile_i + 2 * (long long)nblk, qy, qa);   // in flight under two tiles' MFMAs
    T ay[R / 4][MB], bv[R / 4];                                // every operand of the tile out of LDS first, then the chain (the scheduler would
#pragma unroll                                                 // otherwise put an LDS wait in front of every MFMA)
    for (int u = 0; u < R / 4; ++u) {
      const int row = 4 * u + lg;
      bv[u] = sa[row * LDP + ncol];
#pragma unroll
      for (int m = 0; m < MB; ++m) ay[u][m] = sy[row * LDP + 16 * m + li];
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < R / 4; ++u) {
      colsum += bv[u];
#pragma unroll
      for (int m = 0; m < MB; ++m) acc[m] = TR::mfma(ay[u][m], bv[u], acc[m]);
    }
  }
#pragma unroll
  for (int m = 0; m < MB; ++m)
#pragma unroll
    for (int i = 0; i < 4; ++i) out[(16 * m + TR::acc_row(lane, i)) * D + ncol] = acc[m][i];
  colsum += __shfl_xor(colsum, 16, 64);
  colsum += __shfl_xor(colsum, 32, 64);
  if (lg == 0) out[D * D + ncol] = colsum;
  __syncthreads();                                             // the LDS tiles are free again (the tile passes use the same bytes)
}

template <typename T, int D>
__global__ __launch_bounds__(D * 4) void k_linadj(LinAdjArgs A) {
  constexpr int S = kLaS;
  constexpr int NW = D / 16;                                   // wavefronts per workgroup
  constexpr int E = D * D + D;                                 // theta entries in the padded layout: [D][D], then the bias row
  constexpr int TJ = (D / 16) * (D / 16);                      // 16 x 16 tiles of a D x D product
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  __shared__ LaShared<kLaMaxG> sh;
  Ctl& s_c = sh.c;
  const int G = (int)gridDim.x, blk = (int)blockIdx.x;
  const int dim = A.p.s.dim;
  const double sgn = A.p.s.rhs.sign;                           // s
  const T* Wm = (const T*)A.p.s.rhs.w[0];
  const T* bias = A.has_bias ? (const T*)A.p.s.rhs.b[0] : nullptr;
  // every workgroup runs BOTH systems over its own tiles, one after the other, and reloads its resident matrix slice in between
  // (128 KB from L2 per switch at dim 128: microseconds against the 200 of a pass) - so that the rows of y1 and a1 a workgroup needs for
  // its slab partial are rows it wrote itself, and every workgroup does the same amount of work per pass (no half waits for the other)
  LinCtx<T, D> cx;
  bool padded_ready = false;                                   // (A.wpad is written in the prologue: readable after the first hand-off)
  auto load_system = [&](int sys) {                            // 0: y' = s (y W + b);  1: a' = -s a W^T
    if (padded_ready) cx.init_padded((const T*)A.wpad + (long long)sys * D * D, sys == 0 ? bias : (const T*)nullptr, sys == 0 ? sgn : -sgn, (T*)smem_raw, dim);
    else cx.init_matrix(Wm, sys == 0 ? bias : (const T*)nullptr, sys == 0 ? sgn : -sgn, sys != 0, (T*)smem_raw, dim);
  };
  CtrlParams cp = A.p.s.cp;
  cp.t_out = persist_stage_tout(A.p, sh.tout, kPersistTSmall);
  const double* t_out = cp.t_out;
  const long long batch = A.p.s.batch;
  const long long n_state = batch * dim, n_theta = (long long)dim * dim + (A.has_bias ? dim : 0);
  const int tid = (int)threadIdx.x, nthr = (int)blockDim.x;
  const int wave = tid >> 6;
  const int gw = blk * NW + wave, ngw = G * NW;                // this wavefront among the grid's
  const int gt = blk * nthr + tid, ngt = G * nthr;
  const int gg = gt >> 3, ngg = ngt >> 3, part8 = tid & 7;     // groups of eight lanes (one output of a vector job each)
  const int lane = tid & 63, part7 = lane % 7;                 // groups of seven lanes (one entry of a fold / combination each)
  const int grp7 = lane < 63 ? wave * 9 + lane / 7 : -1, ngrp7 = NW * 9;
  const int EPW = (E + G - 1) / G;                             // theta entries per workgroup (a contiguous run: coalesced)
  const int e_lo = blk * EPW, e_hi = e_lo + EPW < E ? e_lo + EPW : E;
  auto entry_valid = [&](int e) { return e < D * D ? (e / D < dim && e % D < dim) : (A.has_bias && e - D * D < dim); };
  unsigned gen = 0;
  bool ok = true;
  double mine[8];
  auto zero_mine = [&]() {
#pragma unroll
    for (int i = 0; i < 8; ++i) mine[i] = 0.0;
  };
  auto tick = [&](int slot) {                                  // workgroup 0, thread 0: where the time of a segment goes
    if (blk == 0 && tid == 0) { const long long now = (long long)wall_clock64(); sh.prof[slot] += now - sh.tk_prev; sh.tk_prev = now; }
  };
  if (tid == 0) {
    persist_init_ctl(s_c, A.p); sh.ok = 1; sh.skip_initb = 0;
    for (int i = 0; i < 16; ++i) sh.prof[i] = 0;
    sh.tk_prev = (long long)wall_clock64();
    sh.adjt = (double)*(const T*)A.adjt_in;
  }
  __syncthreads();

  T* const pl = (T*)A.planes;
  const long long pstride = A.stride / (long long)sizeof(T);
  T* const y_pa = pl; T* const y_pb = pl + pstride; T* const y_fa = pl + 2 * pstride; T* const y_fb = pl + 3 * pstride;
  T* const a_pa = pl + 4 * pstride; T* const a_pb = pl + 5 * pstride; T* const a_fa = pl + 6 * pstride; T* const a_fb = pl + 7 * pstride;
  double* const th0 = A.theta;                                 // adj_params at the step's start
  double* const th1 = A.theta + E;                             // ... after the attempt
  T* const my_part = (T*)A.gpart + (long long)blk * E;

  // ---- the small products after a new start state (y0, a0): slab partials -> G0 | g0 -> L_p, g0 P_q -> M_pq ----------------------
  auto fold_g0 = [&]() {                                       // entries e_lo .. e_hi of G0 | g0: seven lanes per entry, lane j sums the
    const T* part = (const T*)A.gpart;                         // slabs j, j + 7, ... in eight interleaved chains (a fixed order)
    for (int e0 = e_lo; e0 < e_hi; e0 += ngrp7) {              // (uniform trip count: the shuffles need every lane)
      const int e = e0 + grp7;
      const bool live = grp7 >= 0 && e < e_hi;
      double q[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      if (live) {
        int g = part7;
        for (; g + 49 < G; g += 56) {
#pragma unroll
          for (int u = 0; u < 8; ++u) q[u] += (double)part[(long long)(g + 7 * u) * E + e];
        }
        for (; g < G; g += 7) q[0] += (double)part[(long long)g * E + e];
      }
      const double s_ = group7_sum(((q[0] + q[1]) + (q[2] + q[3])) + ((q[4] + q[5]) + (q[6] + q[7])), lane);
      if (live && part7 == 0) st_wt(A.g0 + e, s_);
    }
  };
  auto level_l = [&](const double* g0c) {                      // L_p = P_p G0 + c_p g0 (p = 1..S) and the bias rows g0 P_q (q = 0..S)
    // (this level starts the moment an attempt is accepted and nothing else runs beside it: its latency is the step's - every product is
    // split over KSPLIT wavefronts of a workgroup, one memory round trip each)
    constexpr int KSPLIT = NW >= 4 ? 4 : NW, JPW = NW / KSPLIT;
    const int slot = wave / KSPLIT, kpart = wave % KSPLIT;
    for (int j0 = 0; j0 < S * TJ; j0 += G * JPW) {             // (uniform trip count: the job synchronises the workgroup)
      const int j = j0 + blk * JPW + slot;
      const bool active = j < S * TJ;
      const int p = active ? 1 + j / TJ : 1, t = active ? j % TJ : 0;
      la_tile_job_split<D, KSPLIT>(A.pw + (long long)p * D * D, g0c, A.lmat + (long long)p * D * D, D, t / (D / 16), t % (D / 16),
                                   A.has_bias ? A.cvec + p * D : nullptr, g0c + D * D, active, slot, kpart, (double*)smem_raw);
    }
    // row D of M_0q: g0 P_q - small vector jobs, dealt from the LAST lane groups of the grid (the first workgroups carry the split tile
    // jobs above and two M jobs per wavefront: with these on top they were the last to reach every hand-off by 10 - 12 us), two partial
    // sums per lane (a chain of 16 dependent loads otherwise)
    for (int j = ngg - 1 - gg; j < kLaP * D; j += ngg) {
      const int q = j / D, c = j % D;
      const double* P = A.pw + (long long)q * D * D;
      double s0 = 0.0, s1 = 0.0;
      int k = part8;
      for (; k + 8 < D; k += 16) {
        s0 = fma(g0c[D * D + k], P[(long long)k * D + c], s0);
        s1 = fma(g0c[D * D + k + 8], P[(long long)(k + 8) * D + c], s1);
      }
      for (; k < D; k += 8) s0 = fma(g0c[D * D + k], P[(long long)k * D + c], s0);
      const double s_ = group8_sum(s0 + s1);
      if (part8 == 0) st_wt(A.mmat + (long long)q * E + D * D + c, s_);
    }
  };
  auto level_m = [&](const double* g0c) {                      // M_pq = L_p P_q (L_0 = G0)
    for (int j = gw; j < kLaPP * TJ; j += ngw) {
      const int pq = j / TJ, t = j % TJ, p = pq / kLaP, q = pq % kLaP;
      la_tile_job<D>(p == 0 ? g0c : A.lmat + (long long)p * D * D, A.pw + (long long)q * D * D, A.mmat + (long long)pq * E, D,
                     t / (D / 16), t % (D / 16), nullptr, nullptr);
    }
  };
  auto barrier_handoff = [&]() {                               // (the small products' outputs are write-through stores: no release)
    zero_mine();
    ok = ok && la_exchange(A.p, sh, gen++, mine, false, true);
  };

  // ---- prologue ------------------------------------------------------------------------------------------------------------------
  // adj_params -> the padded float64 layout; P_0 = I, P_1 = W^T, c_0 = 0, c_1 = b; W and W^T zero padded in the state dtype
  for (int e = e_lo + tid; e < e_hi; e += nthr) {              // (the entries this workgroup owns for the whole segment)
    double v = 0.0;
    if (e < D * D) { if (e / D < dim && e % D < dim) v = (double)((const T*)A.th_in)[(e / D) * dim + e % D]; }
    else if (A.has_bias && e - D * D < dim) v = (double)((const T*)A.th_in)[dim * dim + (e - D * D)];
    th0[e] = v;
  }
  for (int e = gt; e < D * D; e += ngt) {
    const int i = e / D, j = e % D;
    st_wt(A.pw + e, i == j ? 1.0 : 0.0);
    const T wt = (i < dim && j < dim) ? Wm[(long long)j * dim + i] : (T)0;
    st_wt(A.pw + D * D + e, (double)wt);
    const T wn = (i < dim && j < dim) ? Wm[(long long)i * dim + j] : (T)0;
    if constexpr (sizeof(T) == 8) { st_wt((double*)A.wpad + e, (double)wn); st_wt((double*)A.wpad + D * D + e, (double)wt); }
    else {
      __hip_atomic_store((float*)A.wpad + e, (float)wn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store((float*)A.wpad + D * D + e, (float)wt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  for (int e = gt; e < 2 * D; e += ngt) st_wt(A.cvec + e, (e >= D && bias != nullptr && e - D < dim) ? (double)bias[e - D] : 0.0);
  // The powers of W^T by doubling (P_q = P_n P_(q-n), c_q = P_n c_(q-n), n = 1, 2, 4) ride BETWEEN the three batch-sized passes of the
  // prologue, like the chain of an accepted step between the tile passes: a workgroup writes its share of a level, flags it, runs a pass
  // and only then asks for everybody's flags.
  unsigned cgen = 0;                                           // flag rounds so far (uniform over the grid)
  auto power_level = [&](int n) {
    const int q_hi = 2 * n < S ? 2 * n : S;
    for (int j = gw; j < (q_hi - n) * TJ; j += ngw) {
      const int q = n + 1 + j / TJ, t = j % TJ;
      la_tile_job<D>(A.pw + (long long)n * D * D, A.pw + (long long)(q - n) * D * D, A.pw + (long long)q * D * D, D, t / (D / 16), t % (D / 16),
                     nullptr, nullptr);
    }
    for (int j = gg; j < (q_hi - n) * D; j += ngg) {
      const int q = n + 1 + j / D, i = j % D;
      const double* P = A.pw + (long long)n * D * D + (long long)i * D;
      const double* c = A.cvec + (q - n) * D;
      double s_ = 0.0;
#pragma unroll
      for (int k = part8; k < D; k += 8) s_ = fma(P[k], c[k], s_);
      s_ = group8_sum(s_);
      if (part8 == 0) st_wt(A.cvec + q * D + i, s_);
    }
  };
  la_flag_set(A.p, 0, cgen++);                                 // P_0, P_1, c_1, the padded matrices
  tick(5);
  // f0 of both systems (misc.py:225-233's sums ride in the record), the slab partial of the start state
  {
#pragma clang loop unroll(disable)
    for (int sys = 0; sys < 2; ++sys) {                        // (one instance of the pass in the code: the systems differ in pointers only)
      Acc acc;
      load_system(sys);
      double dsum = 0.0;
      if (sys == 0 && A.grad_in != nullptr)
        lin_f0_pass<T, D, true, true>(A.p.s, (const T*)A.y_in, y_fa, (T*)nullptr, (T*)nullptr, cx, acc, blk, G, (const T*)A.grad_in, &dsum);
      else
        lin_f0_pass<T, D, true>(A.p.s, sys == 0 ? (const T*)A.y_in : (const T*)A.a_in, sys == 0 ? y_fa : a_fa, (T*)nullptr, (T*)nullptr, cx, acc);
      double r[5];
      block_reduce_thread0(acc, sh.red, r);
      if (tid == 0) { sh.mine[3 * sys] = r[2]; sh.mine[3 * sys + 1] = r[3]; sh.mine[3 * sys + 2] = r[4]; }
      if (sys == 0) {                                          // the dot product's block sum (a sum: the sumb slot of a second reduction)
        Acc ad; ad.sumb = dsum;
        block_reduce_thread0(ad, sh.red, r);
        if (tid == 0) sh.mine[6] = r[3];
      }
      tick(14);
      ok = ok && la_flag_wait(A.p, sh, 0, cgen - 1u);          // (set a whole pass ago)
      padded_ready = true;
      if (ok) power_level(sys == 0 ? 1 : 2);                   // P_2 ; P_3, P_4
      la_flag_set(A.p, 0, cgen++);
      tick(15);
    }
    la_slab_pass<T, D>((const T*)A.y_in, (const T*)A.a_in, batch, dim, blk, G, my_part, (T*)smem_raw);
    tick(3);
    ok = ok && la_flag_wait(A.p, sh, 0, cgen - 1u);
    if (ok) power_level(4);                                    // P_5, P_6 (visible after the hand-off below)
    tick(15);
    zero_mine();
    if (tid == 0) {
#pragma unroll
      for (int i = 0; i < 7; ++i) mine[i] = sh.mine[i];
    }
    ok = la_exchange(A.p, sh, gen++, mine, true, true) && ok;  // (release: the slab partials are plain stores)
    if (ok && tid < 64) {                                      // the y and adj_y records of the initial step
      const double dsum = la_fold_sum(sh.vals[6], 0, G);
      if (tid == 0) {                                          // adjoint.py:134-140: adj_t -= f(t_i, y_i) . grad_output_i (f0 carries the time-reversal sign)
        const T dl = (T)(sgn * dsum);
        sh.dldt = A.grad_in != nullptr ? (double)dl : 0.0;
        if (A.grad_in != nullptr) sh.adjt = (double)((T)sh.adjt - dl);
        if (blk == 0 && A.dldt_out != nullptr) *(T*)A.dldt_out = A.grad_in != nullptr ? dl : (T)0;
      }
      for (int k = 0; k < 2; ++k) {
        const double sa = la_fold_sum(sh.vals[3 * k], 0, G), sb = la_fold_sum(sh.vals[3 * k + 1], 0, G), fl = la_fold_max(sh.vals[3 * k + 2], 0, G);
        if (tid == 0) {
          double* o = sh.seg_rec[k];
          o[R_MAXA] = 0; o[R_MAXB] = 0; o[R_SUMA] = sa; o[R_SUMB] = sb; o[R_FLAG] = fl; o[R_N] = (double)n_state; o[6] = 0; o[7] = 0;
        }
      }
    }
  }
  tick(4);
  if (ok) fold_g0();
  tick(11);
  barrier_handoff();
  tick(4);
  // adj_params' share of misc._select_initial_step: f0 = -s M_00 = -s [G0 ; g0] - the fold's output itself.  The products L_p, M_pq of the
  // start state are not needed before the first attempt's combinations: they ride between ITS tile passes like every later step's.
  {
    Acc at;
    if (ok) {
      for (int e = e_lo + tid; e < e_hi; e += nthr) {
        if (!entry_valid(e)) continue;
        const T v0 = (T)th0[e], f0 = (T)(-sgn * A.g0[e]);
        const T sc = (T)cp.atol + fabs(v0) * (T)cp.rtol;       // misc.py:225
        const double q0 = (double)(v0 / sc), q1 = (double)(f0 / sc);
        at.suma += q0 * q0; at.sumb += q1 * q1;
        at.maxa = fmax(at.maxa, fabs((double)v0));
        if (!finite_(v0)) at.flag = 1;
      }
    }
    double r[5];
    block_reduce_thread0(at, sh.red, r);
    zero_mine();
    mine[0] = r[2]; mine[1] = r[3]; mine[2] = r[4]; mine[3] = r[0];
    ok = ok && la_exchange(A.p, sh, gen++, mine, false, false);
    if (ok && tid < 64) {
      const double sa = la_fold_sum(sh.vals[0], 0, G), sb = la_fold_sum(sh.vals[1], 0, G), fl = la_fold_max(sh.vals[2], 0, G);
      const double m0 = la_fold_max(sh.vals[3], 0, G);
      if (tid == 0) {
        sh.th0_max = m0;
        double* o = sh.seg_rec[3];
        o[R_MAXA] = 0; o[R_MAXB] = 0; o[R_SUMA] = sa; o[R_SUMB] = sb; o[R_FLAG] = fl; o[R_N] = (double)n_theta; o[6] = 0; o[7] = 0;
        double* t_ = sh.seg_rec[2];                            // adj_t: one element, zero derivative
        const T v0 = (T)sh.adjt;
        const T sc = (T)cp.atol + fabs(v0) * (T)cp.rtol;
        const double q0 = (double)(v0 / sc);
        t_[R_MAXA] = 0; t_[R_MAXB] = 0; t_[R_SUMA] = q0 * q0; t_[R_SUMB] = 0; t_[R_FLAG] = finite_(v0) ? 0.0 : 1.0; t_[R_N] = 1.0; t_[6] = 0; t_[7] = 0;
        controller_apply_seg(&s_c, &sh.seg, sh.seg_rec, 4, PH_F0, cp);
        // h0 = +inf is the normal case of this tuple (adj_t has d1 = 0, misc.py:233): the second evaluation cannot change
        // min(100 h0, h1) then - every d2 is 0 or NaN, never positive - and is skipped unless max(d1) <= 1e-15 (mi_ode_adjoint.h)
        sh.skip_initb = (isinf(s_c.h0) && s_c.h0 > 0.0 && py_max(sh.seg.d1, 4) > 1e-15) ? 1 : 0;
        if (sh.skip_initb) {
          for (int k = 0; k < 4; ++k) sh.seg_rec[k][R_SUMA] = 0.0;     // d2 = 0 / inf
          controller_apply_seg(&s_c, &sh.seg, sh.seg_rec, 4, PH_INITB, cp);
        }
      }
    }
    __syncthreads();
  }
  int chain = 0;                                               // 1: the L_p of the new start state are being written (my flag is set), the M_pq
                                                               // follow after the next y pass; 2: ... the M_pq are being written; 0: M_pq valid
  if (ok) level_l(A.g0);
  la_flag_set(A.p, 0, cgen++);
  chain = 1;
  if (ok && !uniform_i(sh.skip_initb)) {                       // misc.py:235-245 for a finite h0 (rare: adj_t = 0): the M_pq right away
    ok = ok && la_flag_wait(A.p, sh, 0, cgen - 1u);
    if (ok) level_m(A.g0);
    la_flag_set(A.p, 0, cgen++);
    ok = ok && la_flag_wait(A.p, sh, 0, cgen - 1u);
    chain = 0;
    Acc at;
    const double h0d = uniform_d(s_c.h0);
#pragma clang loop unroll(disable)
    for (int sys = 0; sys < 2; ++sys) {
      Acc acc;
      load_system(sys);
      lin_initb_pass<T, D, true>(A.p.s, sys == 0 ? (const T*)A.y_in : (const T*)A.a_in, sys == 0 ? y_fa : a_fa, (T)h0d, cx, acc);
      double r[5];
      block_reduce_thread0(acc, sh.red, r);
      if (tid == 0) sh.mine[sys] = r[2];
    }
    const double hT = (double)(T)h0d, sh_ = sgn * hT;
    for (int e = e_lo + tid; e < e_hi; e += nthr) {
      if (!entry_valid(e)) continue;
      const double m00 = A.mmat[e], m01 = A.mmat[(long long)1 * E + e], m10 = A.mmat[(long long)kLaP * E + e], m11 = A.mmat[(long long)(kLaP + 1) * E + e];
      const T v0 = (T)th0[e], f0 = (T)(-sgn * m00);
      const T f1 = (T)(-sgn * (m00 + sh_ * m10 - sh_ * m01 - sh_ * sh_ * m11));   // -s (I + s h W)^T [G0; g0] (I - s h W^T), bias row: p = 0 only
      const T sc = (T)cp.atol + fabs(v0) * (T)cp.rtol;
      const double q = (double)((f1 - f0) / sc);               // misc.py:237
      at.suma += q * q;
    }
    double r3[5];
    block_reduce_thread0(at, sh.red, r3);
    zero_mine();
    if (tid == 0) { mine[0] = sh.mine[0]; mine[1] = sh.mine[1]; mine[2] = r3[2]; }
    ok = la_exchange(A.p, sh, gen++, mine, false, false);
    if (ok && tid < 64) {
      const double sy = la_fold_sum(sh.vals[0], 0, G), sa = la_fold_sum(sh.vals[1], 0, G), st_ = la_fold_sum(sh.vals[2], 0, G);
      if (tid == 0) {
        sh.seg_rec[0][R_SUMA] = sy; sh.seg_rec[1][R_SUMA] = sa; sh.seg_rec[2][R_SUMA] = 0.0; sh.seg_rec[3][R_SUMA] = st_;
        controller_apply_seg(&s_c, &sh.seg, sh.seg_rec, 4, PH_INITB, cp);
      }
    }
  }
  auto publish = [&](const AttemptState& st) {                 // thread 0: what the next attempt needs
    sh.pub.dt = st.dt; sh.pub.t1 = st.t1; sh.pub.accepted = st.accepted; sh.pub.done = st.done;
    int j = st.next_out;                                       // speculative output range of the NEXT attempt (resolve_step)
    const double t_new = st.t1 + st.dt;
    while (j < st.n_out && !(t_out[j] > t_new)) ++j;
    sh.pub.emit_lo = st.next_out; sh.pub.emit_hi = j;
    sh.pub.emit_t0 = st.emit_t0; sh.pub.emit_t1 = st.emit_t1; sh.pub.emit_dt = st.emit_dt;
  };
  if (tid == 0) {
    if (!ok) { s_c.status |= MI_ODE_ST_SYNC_TIMEOUT; s_c.done = 1; }
    else set_outputs_apply(&s_c, A.p.n_out);
    AttemptState st;
    st.load(s_c);
    st.accepted = 0;
    publish(st);
    sh.st = st;
  }
  __syncthreads();
  tick(5);

  // ---- the adaptive loop (dopri5.py:82-121) --------------------------------------------------------------------------------------
  int cur = -1, cur_f = 0;                                     // state: -1 the caller's (y_in, a_in), 0 / 1 planes a / b; derivative: plane fa / fb
  int gcur = 0;                                                // which half of A.g0 holds G0 | g0 of the step's start state
  const double* ktab = A.ktab;
  bool emitted = false;
  while (!uniform_i(sh.pub.done)) {
    const double dt_u = uniform_d(sh.pub.dt), t1_u = uniform_d(sh.pub.t1);
    const int nxt = cur == 0 ? 1 : 0;
    const int j_lo = uniform_i(sh.pub.emit_lo), j_hi = uniform_i(sh.pub.emit_hi);
    const T hs_T = (T)dt_u;
    double* const g0c = A.g0 + (long long)gcur * E;           // G0 | g0 of the start state ...
    double* const g0n = A.g0 + (long long)(1 - gcur) * E;     // ... of this attempt's end state
    auto planes_of = [&](int sys, StepPlanes<T, S>& P) {       // state / derivative planes of system `sys` for this attempt
      T* const pa = sys == 0 ? y_pa : a_pa; T* const pb = sys == 0 ? y_pb : a_pb;
      T* const fa = sys == 0 ? y_fa : a_fa; T* const fb = sys == 0 ? y_fb : a_fb;
      const T* const in = sys == 0 ? (const T*)A.y_in : (const T*)A.a_in;
      P.y0 = cur < 0 ? in : (cur == 0 ? pa : pb); P.f0 = cur_f == 0 ? fa : fb;
      P.y1 = nxt == 0 ? pa : pb; P.f1 = cur_f == 0 ? fb : fa;
      P.hs = hs_T; P.t0 = (T)t1_u;
      P.t_start = t1_u; P.dt64 = dt_u; P.t_new = t1_u + dt_u;
      // adj_y(t_end) is the tile pass's speculative dense output; the reference discards y(t_end) (adjoint.py:155-160)
      P.j_lo = sys == 0 ? 0 : j_lo; P.j_hi = sys == 0 ? 0 : j_hi;
    };
    // the weights of this attempt's combinations of the M_pq, h = dt in the state dtype (rk_common.py:46): -s h K^c_pq (s h)^p (-s h)^q for
    // the solution and the error estimate; pi_S[p] pi_S[q] (s h)^p (-s h)^q - no factor -s h - for G0 | g0 of the END state: y1 and a1 are the
    // last stage inputs, so y1^T a1 follows from the start state's products like everything else (oracle: end_state_products_powers) and
    // the only product over the batch of a whole backward interval is the prologue's
    {
      const double hT = (double)hs_T;
      for (int i = tid; i < 3 * kLaPP; i += nthr) {
        const int c = i / kLaPP, pq = i % kLaPP, p = pq / kLaP, q = pq % kLaP;
        double w_ = c < 2 ? -sgn * hT * ktab[c * kLaPP + pq] : ktab[3 * kLaPP + pq];
        for (int u = 0; u < p; ++u) w_ *= sgn * hT;
        for (int u = 0; u < q; ++u) w_ *= -sgn * hT;
        sh.kq[c][pq] = w_;
      }
    }
#pragma clang loop unroll(disable)
    for (int sys = 0; sys < 2; ++sys) {
      StepPlanes<T, S> P;
      planes_of(sys, P);
      Acc acc;
      tick(7);
      load_system(sys);
      tick(8);
      lin_attempt_pass<T, D, S, false, true>(A.p.s, P, cx, acc, t_out, (T*)sh.coef);   // (its first barrier also publishes sh.kq)
      tick(0);
      double r[5];
      block_reduce_thread0(acc, sh.red, r);
      if (tid == 0) { sh.mine[3 * sys] = r[0]; sh.mine[3 * sys + 1] = r[1]; sh.mine[3 * sys + 2] = r[2]; }
      tick(9);
      // the chain of small products of an accepted step rides between the tile passes of the next attempt: every workgroup wrote its
      // share of the L_p before its y pass, of the M_pq before its a pass - when it asks for everybody's, they have been there for 200 us
      if (chain == 1 && sys == 0) {
        ok = ok && la_flag_wait(A.p, sh, 0, cgen - 1u);
        tick(10);
        if (ok) level_m(g0c);
        la_flag_set(A.p, 0, cgen++);
        chain = 2;
        tick(13);
      }
    }
    if (chain == 2) { ok = ok && la_flag_wait(A.p, sh, 0, cgen - 1u); chain = 0; tick(10); }
    Acc at;
    for (int e0 = e_lo; e0 < e_hi; e0 += ngrp7) {              // seven lanes per entry, lane j the terms pq = 7 j .. 7 j + 6
      const int e = e0 + grp7;
      const bool live = grp7 >= 0 && e < e_hi;
      double d_sol = 0.0, d_err = 0.0, d_nxt = 0.0;
      if (live) {
#pragma unroll
        for (int u = 0; u < 7; ++u) {
          const int pq = 7 * part7 + u;
          const double m = A.mmat[(long long)pq * E + e];
          d_sol = fma(sh.kq[0][pq], m, d_sol);
          d_err = fma(sh.kq[1][pq], m, d_err);
          d_nxt = fma(sh.kq[2][pq], m, d_nxt);
        }
      }
      d_sol = group7_sum(d_sol, lane); d_err = group7_sum(d_err, lane); d_nxt = group7_sum(d_nxt, lane);
      if (live && part7 == 0) {
        const double v1 = th0[e] + d_sol;
        th1[e] = v1;
        st_wt(g0n + e, d_nxt);
        if (entry_valid(e)) {
          at.maxb = fmax(at.maxb, fabs((double)(T)v1));
          const double er = (double)(T)d_err;
          at.suma += er * er;
        }
      }
    }
    tick(1);
    {
      double r3[5];
      block_reduce_thread0(at, sh.red, r3);
      zero_mine();
      if (tid == 0) {
#pragma unroll
        for (int i = 0; i < 6; ++i) mine[i] = sh.mine[i];
        mine[6] = r3[1]; mine[7] = r3[2];
      }
      if (A.skew != nullptr && tid == 0 && gen < 32u) A.skew[((long long)gen * G + blk) * 2] = (long long)wall_clock64();
      ok = la_exchange(A.p, sh, gen++, mine, false, true) && ok;  // (acquire: G0 | g0 of the end state - write-through stores - read by the L_p products)
      if (A.skew != nullptr && tid == 0 && gen <= 32u) A.skew[((long long)(gen - 1u) * G + blk) * 2 + 1] = (long long)wall_clock64();
    }
    if (tid < 64) {
      double th1_max = 0.0;
      if (ok) {
        for (int k = 0; k < 2; ++k) {
          const double ma = la_fold_max(sh.vals[3 * k], 0, G), mb = la_fold_max(sh.vals[3 * k + 1], 0, G), sa = la_fold_sum(sh.vals[3 * k + 2], 0, G);
          if (tid == 0) {
            double* o = sh.seg_rec[k];
            o[R_MAXA] = ma; o[R_MAXB] = mb; o[R_SUMA] = sa; o[R_SUMB] = 0; o[R_FLAG] = 0; o[R_N] = (double)n_state;
          }
        }
        const double mb = la_fold_max(sh.vals[6], 0, G), sa = la_fold_sum(sh.vals[7], 0, G);
        if (tid == 0) {
          th1_max = mb;
          double* o = sh.seg_rec[3];                           // (max |theta0| is the previous accepted attempt's max |theta1|)
          o[R_MAXA] = sh.th0_max; o[R_MAXB] = mb; o[R_SUMA] = sa; o[R_SUMB] = 0; o[R_FLAG] = 0; o[R_N] = (double)n_theta;
          double* t_ = sh.seg_rec[2];                          // adj_t: y0 = y1 = adj_t, error estimate 0 (misc.py:256-263 all the same)
          const double av = fabs((double)(T)sh.adjt);
          t_[R_MAXA] = av; t_[R_MAXB] = av; t_[R_SUMA] = 0; t_[R_SUMB] = 0; t_[R_FLAG] = 0; t_[R_N] = 1.0;
        }
      }
      if (tid == 0) {
        AttemptState st = sh.st;
        if (!ok) { st.status |= MI_ODE_ST_SYNC_TIMEOUT; st.done = 1; st.accepted = 0; }
        else attempt_core_seg(st, sh.seg_rec, 4, cp, nullptr, nullptr);
        if (st.accepted) sh.th0_max = th1_max;
        publish(st);
        sh.st = st;
      }
    }
    __syncthreads();
    tick(2);
    if (!uniform_i(sh.pub.accepted)) continue;
    if (uniform_i(sh.pub.done)) {                              // the accepted step covers t_end: adj_params(t_end), adj_t(t_end)
      if (ok && j_hi > j_lo) {
        // dense output as ONE combination (oracle: dense_output_fold_weights): K^w = p1 K^sol + pm K^mid + p0 [p = q = 0] + pS pi_S pi_S^T
        const double hT = (double)hs_T;
        const double x = (double)interp_x<T>(sh.pub.emit_t0, sh.pub.emit_t1, t_out[j_lo]);
        const double x2 = x * x, x3 = x2 * x, x4 = x3 * x;
        const double p1 = -8 * x4 + 14 * x3 - 5 * x2, pm = 16 * x4 - 32 * x3 + 16 * x2, p0 = -2 * x4 + 5 * x3 - 4 * x2 + x, pS = 2 * x4 - 3 * x3 + x2;
        for (int pq = tid; pq < kLaPP; pq += nthr) {
          const int p = pq / kLaP, q = pq % kLaP;
          double w_ = -sgn * hT * (p1 * ktab[pq] + pm * ktab[2 * kLaPP + pq] + pS * ktab[3 * kLaPP + pq] + (pq == 0 ? p0 : 0.0));
          for (int u = 0; u < p; ++u) w_ *= sgn * hT;
          for (int u = 0; u < q; ++u) w_ *= -sgn * hT;
          sh.kq[2][pq] = w_;
        }
        __syncthreads();
        for (int e0 = e_lo; e0 < e_hi; e0 += ngrp7) {
          const int e = e0 + grp7;
          const bool live = grp7 >= 0 && e < e_hi;
          double d_out = 0.0;
          if (live) {
#pragma unroll
            for (int u = 0; u < 7; ++u) d_out = fma(sh.kq[2][7 * part7 + u], A.mmat[(long long)(7 * part7 + u) * E + e], d_out);
          }
          d_out = group7_sum(d_out, lane);
          if (live && part7 == 0) {
            const double v = th0[e] + d_out;
            if (e < D * D) { if (e / D < dim && e % D < dim) ((T*)A.th_out)[(e / D) * dim + e % D] = (T)v; }
            else if (A.has_bias && e - D * D < dim) ((T*)A.th_out)[dim * dim + (e - D * D)] = (T)v;
          }
        }
        if (blk == 0 && tid == 0) {                            // adj_t: the fit of a constant, evaluated as the reference does (interp.py:6-67)
          const T at_ = (T)sh.adjt;
          T co[5];
          quartic_from_mid<T>(at_, at_, at_, (T)0, (T)0, (T)sh.pub.emit_dt, co);
          const T ae = quartic_eval<T>(co, (T)x);
          *(T*)A.adjt_out = ae;
          sh.adjt_end = (double)ae;
        }
        emitted = true;
      }
      tick(6);
      break;
    }
    // accepted, more to come: the planes, adj_params and G0 | g0 move on; the chain L_p -> M_pq of the new start state begins here and
    // ends between the tile passes of the next attempt
    cur = nxt; cur_f ^= 1; gcur ^= 1;
    for (int e = e_lo + tid; e < e_hi; e += nthr) th0[e] = th1[e];
    if (ok) level_l(g0n);                                      // (g0n of this attempt = the new start state's)
    la_flag_set(A.p, 0, cgen++);
    chain = 1;
    tick(12);
    if (A.dbg & 1) {
      ok = ok && la_flag_wait(A.p, sh, 0, cgen - 1u);
      if (ok) level_m(g0n);
      la_flag_set(A.p, 0, cgen++);
      ok = ok && la_flag_wait(A.p, sh, 0, cgen - 1u);
      chain = 0;
    }
    if (!ok) {
      if (tid == 0) { AttemptState st = sh.st; st.status |= MI_ODE_ST_SYNC_TIMEOUT; st.done = 1; publish(st); sh.st = st; }
      __syncthreads();
    }
  }

  if (blk == 0 && tid == 0) {
    AttemptState st = sh.st;
    if (!emitted && st.status == 0) st.status |= MI_ODE_ST_SYNC_TIMEOUT;   // (cannot happen: done without a status means the output was written)
    LinAdjResult res;
    res.t1 = st.t1; res.dt = st.dt; res.ratio = st.ratio; res.h0 = s_c.h0;
    res.n_attempt = st.n_attempt; res.n_accept = st.n_accept; res.status = st.status; res.handoffs = (int)(gen > cgen ? gen : cgen);
    for (int i = 0; i < 16; ++i) res.prof[i] = sh.prof[i];
    res.clk_cycles = s_c.clk_cycles + (long long)__builtin_readcyclecounter();
    res.clk_ticks = s_c.clk_ticks + (long long)wall_clock64();
    res.dldt = sh.dldt; res.adjt_end = sh.adjt_end;
    const long long* src = (const long long*)&res;
    long long* dst = (long long*)A.res;
    for (int i = 0; i < (int)(sizeof(LinAdjResult) / sizeof(long long)); ++i)
      __hip_atomic_store(dst + i, src[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

template <typename T, int D>
constexpr size_t linadj_lds_bytes() {                        // the tile passes' two stage tiles | the slab pass' two operand tiles (padded rows)
  return persist_linear_lds_bytes<T, D>() > (size_t)2 * 16 * (D + 16) * sizeof(T) ? persist_linear_lds_bytes<T, D>() : (size_t)2 * 16 * (D + 16) * sizeof(T);
}

}  // namespace mi
