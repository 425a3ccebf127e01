// Whole-integration kernel for the tiny row-local systems: ONE launch per odeint() call.
//
// Small problems (the reference's own demos and tests: 1 .. 65536 trajectories of dim 2-3) are bound by launch
// latency, not by bytes: a whole-attempt kernel lasts ~5 us and the gap to the next launch another ~5 us.  Here the
// adaptive loop of AdaptiveStepsizeODESolver.integrate (solvers.py:28-41, dopri5.py:82-121) runs INSIDE one kernel:
//   * a thread owns one trajectory; y, f0 and the stage derivatives k_1..k_{S+1} never leave its registers;
//   * before_integrate (f0, misc._select_initial_step) is the kernel's prologue (two grid reductions);
//   * per attempt: stages -> block reduction -> grid hand-off -> EVERY workgroup reduces the same records in the same
//     fixed order and applies the controller to its own copy of the scalar state (same bits everywhere, no broadcast
//     phase); accepted steps write the requested outputs from registers (dense output, dopri5.py:87/interp.py);
//   * HBM traffic = y0 in + solution rows out.
// Grid hand-off: every workgroup publishes its record as sequence-numbered 8-byte words (sc1 write-through stores),
// polls everybody's records with sc1 loads until all words carry the expected number, and folds (no atomics, no
// fences - see below).  Records are double-buffered by hand-off parity: a workgroup can be at most one hand-off ahead
// of the slowest one.
// All workgroups must be co-resident: the host only picks this kernel when gridDim.x <= CUs x occupancy, and the
// spin is bounded (MI_ODE_ST_SYNC_TIMEOUT) so that a scheduling surprise ends in an error, not a hang.
// The arithmetic is that of k_stage_rowlocal<M_F0 / M_INITB> and k_step_rowlocal, operation for operation, and the
// records are reduced in the same order (same grid, same row -> thread map): results are bit-identical to the
// launch-per-attempt path.
#pragma once
#include "mi_ode_step_fused.h"

namespace mi {

#ifdef MI_PERSIST_PROF
#define MI_TICK(var) const long long var = (long long)wall_clock64()
#define MI_TOCK(slot, a, b) do { if (threadIdx.x == 0 && blockIdx.x == 0) s_c.prof[slot] += (b) - (a); } while (0)   /* s_c: LDS */
// every workgroup: when it reached hand-off `g` and when it left it (skew between workgroups vs latency of the exchange)
#define MI_SKEW(g, which) do { if (threadIdx.x == 0 && (g) < 8u) ((long long*)(A.s.partials + 8192))[((g) * gridDim.x + blockIdx.x) * 2 + (which)] = (long long)wall_clock64(); } while (0)
#else
#define MI_TICK(var)
#define MI_TOCK(slot, a, b)
#define MI_SKEW(g, which)
#endif
// Always on (product build too), the linear tile kernel only: the time workgroup 0 spends in the grid hand-offs of a call - what a sharded
// run pays for its records (mi_ode_stats.handoff_us; bench.py prints it per rank).  Two reads of the 100 MHz clock per hand-off by one lane.
#ifdef MI_PERSIST_PROF
#define MI_HANDOFF_T0()
#define MI_HANDOFF_T1()
#else
#define MI_HANDOFF_T0() long long ho_t0_ = 0; if (threadIdx.x == 0 && blockIdx.x == 0) ho_t0_ = (long long)wall_clock64()
#define MI_HANDOFF_T1() do { if (threadIdx.x == 0 && blockIdx.x == 0) s_c.prof[3] += (long long)wall_clock64() - ho_t0_; } while (0)
#endif

constexpr int kPersistTSmall = 8;      // output times that travel as kernel arguments
constexpr int kPersistTout = 1024;     // output times cached in LDS
constexpr int kEmitShareRows = 8;      // up to this many trajectories: the lanes of wavefront 0 share the dense output of a step

struct PersistArgs {
  StepArgs s;                  // tableau, RHS, controller parameters; out = solution[1:], t_out = t[1:] (device);
                               // partials = 2 x gridDim.x records; ctl = initial scalar state in, final state out
  const void* y0;              // caller's initial state [batch, D]
  void* out0;                  // solution[0]
  Ctl* ctl_host;               // pinned host copy of the final scalar state (zero-copy store: the host only synchronises)
  double t0, first_dt;         // scalar state at entry (the kernel builds its Ctl itself: no upload)
  double t_small[kPersistTSmall];   // the output times when n_out <= kPersistTSmall (else s.t_out, device)
  // batch-sharded runs: records also cross ranks, through a registered host segment (null: single rank)
  double* xrank;               // [2 parities][world][kXRec] doubles, host memory seen by every rank's GPU
  double* const* xpeers;       // preferred: device array of `world` pointers, entry q = rank q's mailbox ([2][world][kXRec], PEER DEVICE
                               // memory mapped into this process with hipIpcOpenMemHandle; entry `rank` = this rank's own); null: use xrank
  double* gbuf;                // device, [2 parities][kPRec]: the global record, broadcast by workgroup 0
  int world, rank;
  unsigned seq_base;           // sequence numbers of this call's hand-offs are seq_base + 1, + 2, ... (never 0)
  int n_out;                   // T - 1
  int spin_limit;              // bound on the spin iterations of one hand-off
  int xspin_limit;             // ... of the CROSS-RANK part of a hand-off: ranks of a job never launch at the same instant (a late rank is
                               // normal, not a fault), so this bound is seconds, not the fraction of a second workgroups of ONE device get
  int spin_first;              // ... of the FIRST grid hand-off of a launch: the residency check (see grid_reduce_rank)
  int sleep_first, sleep_poll; // back-off (units of 64 clocks): before the first poll / between polls
  // tuple states (mi_ode_desc.n_segments > 1): component k owns workgroups seg_blk[k] .. seg_blk[k + 1] - 1 and seg_rows[k] rows
  int nseg;
  int seg_blk[kMaxSeg + 1];
  long long seg_rows[kMaxSeg];
  int seg_tol;                 // 1: per-component tolerances below
  double seg_rtol[kMaxSeg], seg_atol[kMaxSeg];
};

// thread 0: the scalar state mi_ode_begin would have uploaded
__device__ __forceinline__ void persist_init_ctl(Ctl& c, const PersistArgs& A) {
  int* w = (int*)&c;
  for (int i = 0; i < (int)(sizeof(Ctl) / sizeof(int)); ++i) w[i] = 0;
  c.t0 = c.t1 = A.t0;
  c.dt = A.s.cp.auto_first_step ? 0.0 : A.first_dt;
  c.idx_y0 = 0; c.idx_y1 = 1;
  for (int j = 0; j < kMaxK; ++j) c.idx_k[j] = 2 + j;
  c.clk_cycles = -(long long)__builtin_readcyclecounter();    // closed by persist_write_back: the shader clock this launch ran at
  c.clk_ticks = -(long long)wall_clock64();
}

// every thread: output times into LDS when they fit (kernel arguments for tiny T, else the uploaded array)
__device__ __forceinline__ const double* persist_stage_tout(const PersistArgs& A, double* lds_tout, int lds_cap = kPersistTout) {
  if (A.n_out <= kPersistTSmall) {
    if ((int)threadIdx.x < A.n_out) lds_tout[threadIdx.x] = A.t_small[threadIdx.x];
    return lds_tout;
  }
  if (A.n_out <= lds_cap) {
    for (int i = threadIdx.x; i < A.n_out; i += blockDim.x) lds_tout[i] = A.s.t_out[i];
    return lds_tout;
  }
  return A.s.t_out;
}

static_assert(sizeof(Ctl) % sizeof(long long) == 0, "Ctl is copied to the host in 8-byte words");

__device__ __forceinline__ void persist_write_back(const PersistArgs& A, Ctl& c) {
  c.clk_cycles += (long long)__builtin_readcyclecounter();     // shader cycles / 10 ns ticks of this launch (mi_ode_stats.clock_mhz)
  c.clk_ticks += (long long)wall_clock64();
  *A.s.ctl = c;                                               // device copy (mi_ode_get_state / get_stats)
  if (A.ctl_host != nullptr) {
    const long long* src = (const long long*)&c;
    long long* dst = (long long*)A.ctl_host;
    for (int i = 0; i < (int)(sizeof(Ctl) / sizeof(long long)); ++i)
      __hip_atomic_store(dst + i, src[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// k_set_outputs as a device function (solvers.py:33-34 + the entry assertions of dopri5.py:98-100)
__device__ __forceinline__ void set_outputs_apply(Ctl* c, int n_out) {
  c->next_out = 0; c->n_out = n_out; c->n_steps_out = 0; c->emit_lo = c->emit_hi = 0;
  c->done = 0;
  if (c->status != 0) { c->done = 1; return; }
  if (n_out <= 0) { c->done = 1; return; }
  if (c->y0_nonfinite && c->n_attempt == 0) { c->status |= MI_ODE_ST_NONFINITE; c->done = 1; return; }
  if (!(c->t1 + c->dt > c->t1)) { c->status |= MI_ODE_ST_DT_UNDERFLOW; c->done = 1; }
}

// ---- grid hand-off records -------------------------------------------------------------------------------------
// "LL" encoding (as in collective libraries): every 8-byte word carries 4 bytes of payload and a 4-byte sequence
// number, and an aligned 8-byte access is single-copy atomic for every agent - so a reader that finds the expected
// sequence number in all words of a record has the record, whatever order the writes became visible in.  No flag
// published after a drain, no atomics, no fences.  A double travels as two words (written with one 16-byte store, read
// with one 16-byte load: only the 8-byte halves need to be atomic).  A record is 5 values = 10 words (128-byte stride).
// Sequence numbers of a call are seq_base + 1, + 2, ...; the host advances seq_base past those of the previous call.
typedef unsigned long long u64x2_t __attribute__((ext_vector_type(2)));
constexpr int kPRec = 16;                                     // 8-byte words per hand-off record (10 used)
constexpr int kPersistMaxGrid = kMaxBlocks * kRec / (2 * kPRec);   // two parity buffers inside the `partials` allocation

__device__ __forceinline__ void store_ll_sc1(double* p, double value, unsigned seq) {
  const unsigned long long b = (unsigned long long)__double_as_longlong(value);
  u64x2_t v = {(b & 0xffffffffull) | ((unsigned long long)seq << 32), (b >> 32) | ((unsigned long long)seq << 32)};
  asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
}

// all five values of one record: five 16-byte loads in flight, one wait; false while some word lacks `seq`
__device__ __forceinline__ bool load_record_sc1(const double* p, unsigned seq, double (&val)[5]) {
  u64x2_t v[5];
  asm volatile(
      "global_load_dwordx4 %0, %5, off sc1\n\t"
      "global_load_dwordx4 %1, %5, off offset:16 sc1\n\t"
      "global_load_dwordx4 %2, %5, off offset:32 sc1\n\t"
      "global_load_dwordx4 %3, %5, off offset:48 sc1\n\t"
      "global_load_dwordx4 %4, %5, off offset:64 sc1\n\t"
      "s_waitcnt vmcnt(0)"
      : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4])
      : "v"(p)
      : "memory");
  bool ok = true;
#pragma unroll
  for (int i = 0; i < 5; ++i) {
    ok = ok && ((unsigned)(v[i].x >> 32) == seq) && ((unsigned)(v[i].y >> 32) == seq);
    val[i] = __longlong_as_double((long long)((v[i].x & 0xffffffffull) | (v[i].y << 32)));
  }
  return ok;
}
// one value (the sixth word pair of the cross-rank broadcast record)
__device__ __forceinline__ bool load_ll_sc1(const double* p, unsigned seq, double& val) {
  const unsigned long long w0 = (unsigned long long)__hip_atomic_load((const long long*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const unsigned long long w1 = (unsigned long long)__hip_atomic_load((const long long*)p + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  val = __longlong_as_double((long long)((w0 & 0xffffffffull) | (w1 << 32)));
  return (unsigned)(w0 >> 32) == seq && (unsigned)(w1 >> 32) == seq;
}

// Values read from LDS land in VGPRs even when every lane reads the same word; the tile passes have no VGPRs to
// spare, so wave-uniform scalars are moved to SGPRs explicitly.
__device__ __forceinline__ int uniform_i(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ double uniform_d(double v) {
  const long long b = __double_as_longlong(v);
  const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)b);
  const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)((unsigned long long)b >> 32));
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
template <typename P>
__device__ __forceinline__ P* uniform_p(P* p) {
  const unsigned long long b = (unsigned long long)p;
  const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)b);
  const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(b >> 32));
  return (P*)(((unsigned long long)hi << 32) | lo);
}

// what the controller (thread 0, registers) tells the rest of the workgroup after an attempt
struct PersistPub {
  double dt, t1, emit_t0, emit_t1, emit_dt;
  int accepted, emit_lo, emit_hi, done;
};

constexpr int kXRec = 16;                                       // 8-byte words per rank record in the host segment (12 used)
constexpr int kXMaxWorld = 64;

// MAXG: workgroups whose records are staged for the fold, TOUT: output times cached in LDS - parameters because a kernel that
// fills the LDS with its own data cannot afford the 50 KB of the general case
template <int MAXG, int TOUT>
struct PersistSharedT {
  static constexpr int kMaxGrid = MAXG, kTout = TOUT;
  Ctl c;                                                      // prologue (before_integrate) and the final write-back
  PersistPub pub;
  AttemptState st;                                            // MFMA kernel: the loop's scalar state rests here between attempts
  double red[80];
  double coef[kLinCoefMax];                                   // linear tile kernels: dt * tableau products of the attempt (LinCoef)
  double vals[5][MAXG];                                       // every workgroup's record, staged for the fixed-order fold
  double tout[TOUT];                                          // the requested output times, when they fit
  double xr[6][kXMaxWorld + 1];                               // cross-rank hand-off: every rank's record (+ one staging column)
  double seg_rec[kMaxSeg][kRec];                              // tuple states: the combined record of every component
  SegState seg;
  int ok;                                                     // 1 until a hand-off times out
};
using PersistShared = PersistSharedT<kPersistMaxGrid, kPersistTout>;

// ---- cross-rank hand-off through host memory ------------------------------------------------------------------
// Encoding (the "LL" idea of collective libraries): every 8-byte word carries 4 bytes of payload and a 4-byte
// sequence number, and an aligned 8-byte store is single-copy atomic for every agent of the system.  A double travels as
// two such words; a reader accepts a record when all of its words show the expected sequence number.  Nothing depends
// on the ORDER in which PCIe / xGMI deliver the writes, and there is no fence.
__device__ __forceinline__ void ll_store(unsigned long long* p, double v, unsigned seq) {
  const unsigned long long b = (unsigned long long)__double_as_longlong(v);
  __hip_atomic_store(p, (b & 0xffffffffull) | ((unsigned long long)seq << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  __hip_atomic_store(p + 1, (b >> 32) | ((unsigned long long)seq << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
// all 6 values of one rank record; false if some word does not carry `seq` yet
__device__ __forceinline__ bool ll_load_record(const unsigned long long* p, unsigned seq, double (&v)[6]) {
  unsigned long long w[12];
#pragma unroll
  for (int i = 0; i < 12; ++i) w[i] = __hip_atomic_load(p + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  bool ok = true;
#pragma unroll
  for (int i = 0; i < 12; ++i) ok = ok && ((unsigned)(w[i] >> 32) == seq);
#pragma unroll
  for (int i = 0; i < 6; ++i) v[i] = __longlong_as_double((long long)((w[2 * i] & 0xffffffffull) | (w[2 * i + 1] << 32)));
  return ok;
}

// Cross-rank hand-off (batch-sharded runs).  On entry thread 0 of EVERY workgroup holds this rank's record r[0..4];
// on exit it holds the record combined over all ranks (rank order, the fold of k_controller) and n_tot = sum of the
// ranks' element counts.  Workgroup 0 is the gateway: it publishes the rank record, lane q of its first wavefront polls
// rank q's record, thread 0 folds and broadcasts the result through device memory (same word encoding) to the other
// workgroups.  Bounded waits, as everywhere.  Two transports, same "LL" word encoding:
//   * peer device memory (A.xpeers): every rank owns a mailbox in its own HBM (uncached / fine-grained allocation,
//     exported with hipIpcGetMemHandle); lane q PUSHES this rank's record into rank q's mailbox - one posted store
//     stream over the xGMI link to that peer - and then polls slot q of its OWN mailbox, i.e. local memory.  No poll
//     ever crosses a link, no host memory, no PCIe;
//   * a host segment shared by the ranks (A.xrank): one slot per rank, written once, polled by everybody over PCIe.
template <class SH>
__device__ __forceinline__ void cross_rank(const PersistArgs& A, SH& sh, unsigned gen, double (&r)[5], double& n_tot,
                                           double n_local) {
  const int W = A.world;
  const unsigned seq = A.seq_base + gen + 1u;
  const unsigned bad = seq ^ 0x80000000u;                     // "the gateway gave up" (never a valid number of this hand-off)
  double* g = A.gbuf + (long long)(gen & 1u) * kPRec;
  if (blockIdx.x == 0) {
    const unsigned long long* poll = nullptr;
    if (A.xpeers != nullptr) {
      if (threadIdx.x == 0) {
#pragma unroll
        for (int i = 0; i < 5; ++i) sh.xr[i][kXMaxWorld] = r[i];              // hand the record to the pushing lanes
      }
      __syncthreads();
      if ((int)threadIdx.x < W) {
        unsigned long long* slot = (unsigned long long*)A.xpeers[threadIdx.x] + ((long long)(gen & 1u) * W + A.rank) * kXRec;
#pragma unroll
        for (int i = 0; i < 5; ++i) ll_store(slot + 2 * i, sh.xr[i][kXMaxWorld], seq);
        ll_store(slot + 10, n_local, seq);
        poll = (const unsigned long long*)A.xpeers[A.rank] + ((long long)(gen & 1u) * W + threadIdx.x) * kXRec;
      }
    } else {
      unsigned long long* seg = (unsigned long long*)A.xrank;
      if (threadIdx.x == 0) {
        unsigned long long* slot = seg + ((long long)(gen & 1u) * W + A.rank) * kXRec;
#pragma unroll
        for (int i = 0; i < 5; ++i) ll_store(slot + 2 * i, r[i], seq);
        ll_store(slot + 10, n_local, seq);
      }
      if ((int)threadIdx.x < W) poll = seg + ((long long)(gen & 1u) * W + threadIdx.x) * kXRec;
    }
    if ((int)threadIdx.x < W) {
      double v[6];
      int spins = 0;
      while (!ll_load_record(poll, seq, v)) {
        for (int i = 0; i < 4; ++i) __builtin_amdgcn_s_sleep(1);
        if (++spins > A.xspin_limit) { sh.ok = 0; break; }
      }
#pragma unroll
      for (int i = 0; i < 6; ++i) sh.xr[i][threadIdx.x] = v[i];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      double m0 = 0, m1 = 0, s0 = 0, s1 = 0, fl = 0, n = 0;
      for (int q = 0; q < W; ++q) {                           // k_controller's fold over the gathered rank records
        m0 = fmax(m0, sh.xr[0][q]); m1 = fmax(m1, sh.xr[1][q]); s0 += sh.xr[2][q]; s1 += sh.xr[3][q];
        fl = fmax(fl, sh.xr[4][q]); n += sh.xr[5][q];
      }
      const unsigned st = sh.ok ? seq : bad;                  // a failed gateway tells everyone (they stop polling)
      store_ll_sc1(g + 0, m0, st); store_ll_sc1(g + 2, m1, st); store_ll_sc1(g + 4, s0, st);
      store_ll_sc1(g + 6, s1, st); store_ll_sc1(g + 8, fl, st); store_ll_sc1(g + 10, n, st);
    }
  }
  if (threadIdx.x == 0) {
    double v[5], nv = 0.0;
    long long spins = 0;
    for (;;) {
      if (load_record_sc1(g, seq, v) && load_ll_sc1(g + 10, seq, nv)) { n_tot = nv; break; }
      double dummy;
      if (load_ll_sc1(g, bad, dummy)) { sh.ok = 0; break; }
      __builtin_amdgcn_s_sleep(2);
      if (++spins > 8LL * A.xspin_limit) { sh.ok = 0; break; }  // (outlasts the gateway's own bounded wait: it reports failures)
    }
#pragma unroll
    for (int i = 0; i < 5; ++i) r[i] = v[i];
  }
  __syncthreads();
}

// Block record -> (grid hand-off) -> combined record {max a, max b, sum a, sum b, flag} in THREAD 0's registers.
// Returns false (to every thread) on a hand-off timeout.  `gen` counts hand-offs (uniform over the grid).  Thread i
// polls record i (one round trip once the slowest workgroup has published), wavefront 0 folds in
// reduce_block_records' fixed order.
template <class SH>
__device__ __forceinline__ bool grid_reduce_rank(const PersistArgs& A, const Acc& acc, SH& sh, unsigned gen,
                                                 double (&r)[5]) {
  const int G = (int)gridDim.x;
  block_reduce_thread0(acc, sh.red, r);
  if (G == 1) return true;                                    // one record: folding it with zeros is exact
  double* buf = A.s.partials + (long long)(gen & 1u) * G * kPRec;
  const unsigned seq = A.seq_base + gen + 1u;
  if (threadIdx.x == 0) {
    double* mine = buf + (long long)blockIdx.x * kPRec;
#pragma unroll
    for (int i = 0; i < 5; ++i) store_ll_sc1(mine + 2 * i, r[i], seq);
  }
  // Co-residency is not something a launch can promise (another stream's kernel, a second process, an occupancy query
  // that is one block per CU too optimistic - MI355X_MICROARCH.md "Residency and cooperative launch").  The FIRST
  // hand-off of a launch doubles as the residency check: every workgroup that runs publishes within the duration of
  // one init pass, a workgroup that was not admitted can only start after a resident one exits - i.e. never.  So the
  // first hand-off gives up after milliseconds (the host then falls back to one launch per attempt, having lost almost
  // nothing); once every workgroup has been seen they stay resident (no pre-emption of running waves) and the later
  // hand-offs only absorb skew.
  const int limit = gen == 0 ? A.spin_first : A.spin_limit;
  for (int b = threadIdx.x; b < G; b += blockDim.x) {
    const double* p = buf + (long long)b * kPRec;
    double v[5];
    int spins = 0;
    for (int i = 0; i < A.sleep_first; ++i) __builtin_amdgcn_s_sleep(1);
    for (;;) {
      if (load_record_sc1(p, seq, v)) break;
      for (int i = 0; i < A.sleep_poll; ++i) __builtin_amdgcn_s_sleep(1);
      if (++spins > limit) { sh.ok = 0; break; }
    }
#pragma unroll
    for (int i = 0; i < 5; ++i) sh.vals[i][b] = v[i];
  }
  __syncthreads();
  const bool ok = sh.ok != 0;
  if (A.nseg > 1) {                                           // one fold per component: sh.seg_rec (grid_reduce_seg)
    if (threadIdx.x < 64 && ok) {
      for (int k = 0; k < A.nseg; ++k) {
        const int b0 = A.seg_blk[k], b1 = A.seg_blk[k + 1];
        double v0 = 0, v1 = 0, v2 = 0, v3 = 0, v4 = 0;
        for (int b = b0 + (int)threadIdx.x; b < b1; b += 64) {
          v0 = fmax(v0, sh.vals[R_MAXA][b]); v1 = fmax(v1, sh.vals[R_MAXB][b]);
          v2 += sh.vals[R_SUMA][b]; v3 += sh.vals[R_SUMB][b]; v4 = fmax(v4, sh.vals[R_FLAG][b]);
        }
        v0 = wave_max(v0); v1 = wave_max(v1); v2 = wave_sum(v2); v3 = wave_sum(v3); v4 = wave_max(v4);
        if (threadIdx.x == 0) {
          double* o = sh.seg_rec[k];
          o[R_MAXA] = v0; o[R_MAXB] = v1; o[R_SUMA] = v2; o[R_SUMB] = v3; o[R_FLAG] = v4; o[6] = 0; o[7] = 0;
        }
      }
    }
    return ok;
  }
  if (threadIdx.x < 64 && ok) {
    double v0 = 0, v1 = 0, v2 = 0, v3 = 0, v4 = 0;
    for (int b = threadIdx.x; b < G; b += 64) {
      v0 = fmax(v0, sh.vals[R_MAXA][b]); v1 = fmax(v1, sh.vals[R_MAXB][b]);
      v2 += sh.vals[R_SUMA][b]; v3 += sh.vals[R_SUMB][b]; v4 = fmax(v4, sh.vals[R_FLAG][b]);
    }
    r[0] = wave_max(v0); r[1] = wave_max(v1); r[2] = wave_sum(v2); r[3] = wave_sum(v3); r[4] = wave_max(v4);
  }
  return ok;
}

// ... and over all ranks when the run is batch-sharded.  n_tot (thread 0): elements behind the combined record.
template <class SH>
__device__ __forceinline__ bool grid_reduce(const PersistArgs& A, const Acc& acc, SH& sh, unsigned gen,
                                            double (&r)[5], double& n_tot) {
  bool ok = grid_reduce_rank(A, acc, sh, gen, r);
  n_tot = (double)A.s.cp.n_local;
  if (A.world > 1 || A.xrank != nullptr || A.xpeers != nullptr) {
    if (ok) cross_rank(A, sh, gen, r, n_tot, (double)A.s.cp.n_local);
    ok = ok && sh.ok != 0;
  }
  return ok;
}

__device__ __forceinline__ void fill_record(double (&rec)[kRec], const double (&r)[5], double n) {
  rec[R_MAXA] = r[0]; rec[R_MAXB] = r[1]; rec[R_SUMA] = r[2]; rec[R_SUMB] = r[3]; rec[R_FLAG] = r[4];
  rec[R_N] = n; rec[6] = 0; rec[7] = 0;
}

template <typename T, int S, bool TS, class RHS, bool FSAL = true>
__global__ __launch_bounds__(256) void k_persist_rowlocal(PersistArgs A) {
  constexpr int D = RHS::D;
  using Row = RowVec<T, D>;
  __shared__ PersistShared sh;
  Ctl& s_c = sh.c;

  const RHS rhs(A.s.rhs);
  const T sign = (T)A.s.rhs.sign;
  CtrlParams cp = A.s.cp;
  cp.t_out = persist_stage_tout(A, sh.tout);                  // the output cursor and the dense output read t from LDS
  const double* t_out = cp.t_out;
  const int nseg = A.nseg;                                    // > 1: tuple state, one component per range of workgroups
  // this thread's first element: a trajectory per thread (row * D) - or, for the cooperative right-hand sides (RhsMlpCoop: round 5),
  // ONE state element per thread, min(256 / dim, ..) trajectories per workgroup, f evaluated by the trajectory's threads together
  long long e0, n_unused;
  bool live;
  rowmap<RHS>(A.s.batch, A.s.dim, A.s.rhs, e0, live, n_unused);
  if (nseg > 1) {
    int sg = 0;
    for (int k = 1; k < nseg; ++k)
      if ((int)blockIdx.x >= A.seg_blk[k]) sg = k;
    live = (long long)((int)blockIdx.x - A.seg_blk[sg]) * blockDim.x + threadIdx.x < A.seg_rows[sg];
  }
  auto seg_counts = [&]() {                                   // thread 0: element counts into the components' records
    for (int k = 0; k < nseg; ++k) sh.seg_rec[k][R_N] = (double)(A.seg_rows[k] * (long long)D);
  };
  unsigned gen = 0;
  double r[5], rec[kRec], n_tot = 0.0;

  if (threadIdx.x == 0) { persist_init_ctl(s_c, A); sh.ok = 1; }
  Row y;
#pragma unroll
  for (int d = 0; d < D; ++d) y.v[d] = (T)0;
  if (live) {
    y = *(const Row*)((const T*)A.y0 + e0);
    *(Row*)((T*)A.out0 + e0) = y;                             // solution[0] = y0 (solvers.py:30)
  }
  __syncthreads();
  const T t_first = (T)s_c.t1;

  // ---- before_integrate: f0 and the norms of misc._select_initial_step (k_stage_rowlocal<M_F0>) ----
  T f0[D];
  bool ok;
  {
    Acc acc;
    T ys[D];
#pragma unroll
    for (int d = 0; d < D; ++d) ys[d] = y.v[d];
    rhs(sign * t_first, ys, f0);
#pragma unroll
    for (int d = 0; d < D; ++d) f0[d] = sign * f0[d];
    if (live) {
#pragma unroll
      for (int d = 0; d < D; ++d) {
        const T sc = (T)cp.atol + fabs(y.v[d]) * (T)cp.rtol;  // misc.py:225
        const double q0 = (double)(y.v[d] / sc);
        acc.suma += q0 * q0;
        if (!finite_(y.v[d])) acc.flag = 1;
      }
#pragma unroll
      for (int d = 0; d < D; ++d) {
        const T sc = (T)cp.atol + fabs(y.v[d]) * (T)cp.rtol;
        const double q1 = (double)(f0[d] / sc);
        acc.sumb += q1 * q1;                                  // misc.py:228
      }
    }
    ok = grid_reduce(A, acc, sh, gen++, r, n_tot);
    if (threadIdx.x == 0 && ok) {
      if (nseg > 1) { seg_counts(); controller_apply_seg(&s_c, &sh.seg, sh.seg_rec, nseg, PH_F0, cp); }
      else { fill_record(rec, r, n_tot); controller_apply(&s_c, rec, PH_F0, cp); }
    }
    __syncthreads();
  }
  if (cp.auto_first_step && ok) {                             // k_stage_rowlocal<M_INITB> (misc.py:235-245)
    Acc acc;
    const T h0 = (T)s_c.h0;
    T ys[D], f1[D];
#pragma unroll
    for (int d = 0; d < D; ++d) ys[d] = y.v[d] + h0 * f0[d];
    rhs(sign * (t_first + (T)1.0 * h0), ys, f1);
    if (live) {
#pragma unroll
      for (int d = 0; d < D; ++d) {
        const T kn = sign * f1[d];
        const T sc = (T)cp.atol + fabs(y.v[d]) * (T)cp.rtol;
        const double q = (double)((kn - f0[d]) / sc);         // misc.py:237
        acc.suma += q * q;
      }
    }
    ok = grid_reduce(A, acc, sh, gen++, r, n_tot);
    if (threadIdx.x == 0 && ok) {
      if (nseg > 1) { seg_counts(); controller_apply_seg(&s_c, &sh.seg, sh.seg_rec, nseg, PH_INITB, cp); }
      else { fill_record(rec, r, n_tot); controller_apply(&s_c, rec, PH_INITB, cp); }
    }
  }
  // thread 0 keeps the scalar state of the loop in registers from here on and publishes what the others need
  AttemptState st;
  if (threadIdx.x == 0) {
    if (!ok) { s_c.status |= MI_ODE_ST_SYNC_TIMEOUT; s_c.done = 1; }
    else set_outputs_apply(&s_c, A.n_out);
    st.load(s_c);
    sh.pub.dt = st.dt; sh.pub.t1 = st.t1; sh.pub.done = st.done; sh.pub.accepted = 0;
  }
  __syncthreads();

  // ---- the adaptive loop (dopri5.py:82-121); the attempt is k_step_rowlocal's ----
  T k[S + 1][D];
#pragma unroll
  for (int d = 0; d < D; ++d) k[0][d] = f0[d];
  while (!sh.pub.done) {
    MI_TICK(tk0);
    T hs = (T)sh.pub.dt;                                      // rk_common.py:46
    asm volatile("" : "+v"(hs));                              // keep the dt*coefficient products out of the loop-invariant set
    const T t0 = (T)sh.pub.t1;                                // rk_common.py:45
    T ys[D];
    auto stage = [&](auto sg_c) {
      constexpr int SG = decltype(sg_c)::value;
#pragma unroll
      for (int d = 0; d < D; ++d) {
        T kk[SG];
#pragma unroll
        for (int j = 0; j < SG; ++j) kk[j] = k[j][d];
        ys[d] = step_combine<T, SG>(y.v[d], kk, hs, A.s);
      }
      T kn[D];
      rhs(sign * (t0 + (T)A.s.alpha[SG - 1] * hs), ys, kn);
#pragma unroll
      for (int d = 0; d < D; ++d) k[SG][d] = sign * kn[d];
    };
    for_stages<1, S>(stage);
    Acc acc;
#pragma unroll
    for (int d = 0; d < D; ++d) {
      T kk[S + 1];
#pragma unroll
      for (int j = 0; j <= S; ++j) kk[j] = k[j][d];
      T err, unused_mid;
      step_finish<T, S>(y.v[d], kk, hs, A.s, err, unused_mid, false);      // y_mid only if an output falls into the step (below)
      if constexpr (!FSAL) ys[d] = step_y1_general<T, S>(y.v[d], kk, hs, A.s);  // rk_common.py:55-56
      if (live) {
        acc.maxa = fmax(acc.maxa, (double)fabs(y.v[d]));
        acc.maxb = fmax(acc.maxb, (double)fabs(ys[d]));
        acc.suma += (double)err * (double)err;
      }
    }
    MI_TICK(tk1);
    ok = grid_reduce(A, acc, sh, gen++, r, n_tot);                   // (its barriers also fence the reads of sh.pub above)
    MI_TICK(tk2);
    if (threadIdx.x == 0) {
      if (!ok) { st.status |= MI_ODE_ST_SYNC_TIMEOUT; st.done = 1; st.accepted = 0; }
      else if (nseg > 1) { seg_counts(); attempt_core_seg(st, sh.seg_rec, nseg, cp, A.seg_tol ? A.seg_rtol : nullptr, A.seg_atol); }
      else { fill_record(rec, r, n_tot); attempt_core(st, rec, cp); }
      sh.pub.dt = st.dt; sh.pub.t1 = st.t1; sh.pub.emit_t0 = st.emit_t0; sh.pub.emit_t1 = st.emit_t1;
      sh.pub.emit_dt = st.emit_dt; sh.pub.accepted = st.accepted; sh.pub.emit_lo = st.emit_lo;
      sh.pub.emit_hi = st.emit_hi; sh.pub.done = st.done;
    }
    MI_TICK(tk3);
    MI_TOCK(0, tk0, tk1); MI_TOCK(1, tk1, tk2); MI_TOCK(2, tk2, tk3);
    __syncthreads();
    if (sh.pub.accepted) {
      // A handful of trajectories with MANY output times per step (the reference's published workloads: one trajectory, 1000 - 10 000
      // outputs): the owner lane would evaluate the interpolant once per output, one after the other (~0.4 us each: a dependent chain of
      // fp64 instructions on one lane).  Here the quartic's coefficients are formed by the owner lane as always and BROADCAST; the 64
      // lanes of wavefront 0 then take one output time each.  Same operations per output, same bits.  (round 6)
      bool emitted = false;
      if constexpr (!TS && !rhs_is_coop<RHS>::value) {
        if (nseg <= 1 && A.s.batch <= kEmitShareRows && sh.pub.emit_hi - sh.pub.emit_lo >= 4) {
          emitted = true;
          if (blockIdx.x == 0 && threadIdx.x < 64) {
            const int j_lo = sh.pub.emit_lo, j_hi = sh.pub.emit_hi;
            const double e_t0 = sh.pub.emit_t0, e_t1 = sh.pub.emit_t1;
            T* const out = (T*)A.s.out;
#pragma unroll
            for (int d = 0; d < D; ++d) {
              T kk[S + 1];
#pragma unroll
              for (int j = 0; j <= S; ++j) kk[j] = k[j][d];
              T err_unused, ymid = y.v[d];
              step_finish<T, S>(y.v[d], kk, hs, A.s, err_unused, ymid, true);
              T co[5];
              quartic_from_mid<T>(y.v[d], ys[d], ymid, kk[0], kk[S], (T)sh.pub.emit_dt, co);
              for (int r_ = 0; r_ < (int)A.s.batch; ++r_) {   // rows 0 .. batch - 1 are lanes 0 .. batch - 1 of this wavefront
                T cb[5];
#pragma unroll
                for (int i = 0; i < 5; ++i) cb[i] = __shfl(co[i], r_);
                for (int j = j_lo + (int)threadIdx.x; j < j_hi; j += 64)
                  out[(long long)j * A.s.n_plane + (long long)r_ * D + d] = quartic_eval<T>(cb, interp_x<T>(e_t0, e_t1, t_out[j]));
              }
            }
          }
        }
      }
      if (!emitted && live && sh.pub.emit_hi > sh.pub.emit_lo) {
        StepPlanes<T, S> P;
        P.j_lo = sh.pub.emit_lo; P.j_hi = sh.pub.emit_hi;
        P.t_start = sh.pub.emit_t0; P.t_new = sh.pub.emit_t1; P.dt64 = sh.pub.emit_dt;
#pragma unroll
        for (int d = 0; d < D; ++d) {
          T kk[S + 1];
#pragma unroll
          for (int j = 0; j <= S; ++j) kk[j] = k[j][d];
          T err_unused, ymid = y.v[d];
          if constexpr (!TS) step_finish<T, S>(y.v[d], kk, hs, A.s, err_unused, ymid, true);
          step_emit<T, S, TS>(A.s, P, y.v[d], ys[d], kk, ymid, e0 + d, t_out);
        }
      }
#pragma unroll
      for (int d = 0; d < D; ++d) { y.v[d] = ys[d]; k[0][d] = k[S][d]; }     // FSAL (rk_common.py:58)
    }
    MI_TICK(tk4);
    MI_TOCK(3, tk3, tk4);
  }

  // ---- hand the final state back: planes idx_y0 / idx_k[0] (mi_ode_get_state; never rotated here), scalars to the host ----
  if (live) {
    *(Row*)((T*)(A.s.planes + (long long)s_c.idx_y0 * A.s.stride) + e0) = y;
    Row f;
#pragma unroll
    for (int d = 0; d < D; ++d) f.v[d] = k[0][d];
    *(Row*)((T*)(A.s.planes + (long long)s_c.idx_k[0] * A.s.stride) + e0) = f;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    st.store(s_c);
    persist_write_back(A, s_c);
  }
}

// ------------------------------------------------------------------------------------------------
// Row-local systems, ANY batch size: the whole call in one launch with the state in HBM planes.
// k_persist_rowlocal keeps a trajectory in its thread's registers for the whole integration, so its grid is batch / 256
// workgroups and stops where those are no longer co-resident (or the all-to-all hand-off no longer cheap): 131 072
// trajectories.  Here a co-resident grid walks the batch with a grid-stride loop per pass (a thread always meets the same
// rows), exactly the passes of k_step_rowlocal / k_stage_rowlocal<M_F0 / M_INITB>: y and f planes alternate on accept
// like the linear tile kernel's below, dense output is emitted speculatively inside the attempt pass (the output cursor only
// moves on accept, so a rejected attempt's rows are overwritten).  Traffic per attempt: y0, f0 in, y1, f1 out.
// ------------------------------------------------------------------------------------------------
template <typename T, int S, bool TS, class RHS, bool FSAL = true>
__global__ __launch_bounds__(512) void k_persist_rowlocal_planes(PersistArgs A) {
  constexpr int D = RHS::D;
  using Row = RowVec<T, D>;
  __shared__ PersistShared sh;
  Ctl& s_c = sh.c;
  const RHS rhs(A.s.rhs);
  const T sign = (T)A.s.rhs.sign;
  CtrlParams cp = A.s.cp;
  cp.t_out = persist_stage_tout(A, sh.tout);
  const double* t_out = cp.t_out;
  unsigned gen = 0;
  double r[5], rec[kRec], n_tot = 0.0;
  if (threadIdx.x == 0) { persist_init_ctl(s_c, A); sh.ok = 1; }
  __syncthreads();
  // this thread's rows: first, first + pitch, ... < batch.  Tuple state (nseg > 1): the workgroup belongs to ONE component - rows
  // seg_off .. seg_off + seg_rows of the packed buffer (components padded to MI_ODE_SEGMENT_ALIGN rows), walked by that
  // component's workgroups only, so a workgroup's record is a record of its component (grid_reduce folds per seg_blk range)
  // Cooperative right-hand sides (a thread per state ELEMENT, round 5): the loop variable is the first trajectory of the workgroup's
  // current group of tpw - the same for every thread of the workgroup, since rhs() contains barriers - and `locate` gives a thread its
  // element of that group (or none: it then runs the loop without loads and stores).
  constexpr bool COOP = rhs_is_coop<RHS>::value;
  const int nseg = A.nseg;
  const int tpw = coop_tpw<RHS>(A.s.rhs, A.s.dim);
  const int c_slot = COOP ? (int)threadIdx.x / A.s.dim : 0, c_col = COOP ? (int)threadIdx.x - c_slot * A.s.dim : 0;
  long long first = COOP ? (long long)blockIdx.x * tpw : (long long)blockIdx.x * blockDim.x + threadIdx.x;
  long long pitch = COOP ? (long long)gridDim.x * tpw : (long long)gridDim.x * blockDim.x;
  long long batch = A.s.batch;
  auto locate = [&](long long it, long long& e0) -> bool {    // e0: this thread's first element for loop position `it`; returns "it exists"
    if constexpr (COOP) {
      const long long traj = it + c_slot;
      e0 = traj * A.s.dim + c_col;
      return c_slot < tpw && traj < batch;
    } else {
      e0 = it * D;
      return true;
    }
  };
  if (nseg > 1) {
    int sg = 0;
    long long off = 0;
    for (int k = 1; k < nseg; ++k)
      if ((int)blockIdx.x >= A.seg_blk[k]) {
        off += (A.seg_rows[k - 1] + MI_ODE_SEGMENT_ALIGN - 1) / MI_ODE_SEGMENT_ALIGN * MI_ODE_SEGMENT_ALIGN;
        sg = k;
      }
    first = off + (long long)((int)blockIdx.x - A.seg_blk[sg]) * blockDim.x + threadIdx.x;
    pitch = (long long)(A.seg_blk[sg + 1] - A.seg_blk[sg]) * blockDim.x;
    batch = off + A.seg_rows[sg];
  }
  auto seg_counts = [&]() {                                   // thread 0: element counts into the components' records
    for (int k = 0; k < nseg; ++k) sh.seg_rec[k][R_N] = (double)(A.seg_rows[k] * (long long)D);
  };

  T* const ya = (T*)(A.s.planes);
  T* const yb = (T*)(A.s.planes + A.s.stride);
  T* const fa = (T*)(A.s.planes + 2 * A.s.stride);
  T* const fb = (T*)(A.s.planes + (long long)(2 + S) * A.s.stride);
  const T* const y_user = (const T*)A.y0;
  auto load_row = [](const T* plane, long long e0) {         // planes rewritten inside this launch: skip this CU's L1 (sc0)
    Row v;
#pragma unroll
    for (int d = 0; d < D; ++d) v.v[d] = __hip_atomic_load(plane + e0 + d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    return v;
  };
  auto zero_row = []() {
    Row v;
#pragma unroll
    for (int d = 0; d < D; ++d) v.v[d] = (T)0;
    return v;
  };

  bool ok;
  const T t_first = (T)uniform_d(s_c.t1);
  {                                                           // before_integrate, first half (misc.py:225-233)
    Acc acc;
    for (long long row = first; row < batch; row += pitch) {
      long long e0;
      const bool live = locate(row, e0);
      Row y = zero_row();
      if (live) {
        y = *(const Row*)(y_user + e0);
        *(Row*)((T*)A.out0 + e0) = y;                         // solution[0] = y0 (solvers.py:30)
      }
      T ys[D], f0[D];
#pragma unroll
      for (int d = 0; d < D; ++d) ys[d] = y.v[d];
      rhs(sign * t_first, ys, f0);
      Row f;
#pragma unroll
      for (int d = 0; d < D; ++d) f.v[d] = sign * f0[d];
      if (!live) continue;                                    // (after rhs(): its barriers are behind every thread)
      *(Row*)(fa + e0) = f;
#pragma unroll
      for (int d = 0; d < D; ++d) {
        const T sc = (T)cp.atol + fabs(y.v[d]) * (T)cp.rtol;  // misc.py:225
        const double q0 = (double)(y.v[d] / sc);
        acc.suma += q0 * q0;
        if (!finite_(y.v[d])) acc.flag = 1;
      }
#pragma unroll
      for (int d = 0; d < D; ++d) {
        const T sc = (T)cp.atol + fabs(y.v[d]) * (T)cp.rtol;
        const double q1 = (double)(f.v[d] / sc);
        acc.sumb += q1 * q1;                                  // misc.py:228
      }
    }
    ok = grid_reduce(A, acc, sh, gen++, r, n_tot);
    if (threadIdx.x == 0 && ok) {
      if (nseg > 1) { seg_counts(); controller_apply_seg(&s_c, &sh.seg, sh.seg_rec, nseg, PH_F0, cp); }
      else { fill_record(rec, r, n_tot); controller_apply(&s_c, rec, PH_F0, cp); }
    }
    __syncthreads();
  }
  if (cp.auto_first_step && ok) {                             // second half (misc.py:235-245)
    Acc acc;
    const T h0 = (T)uniform_d(s_c.h0);
    for (long long row = first; row < batch; row += pitch) {
      long long e0;
      const bool live = locate(row, e0);
      const Row y = live ? *(const Row*)(y_user + e0) : zero_row();
      const Row f0 = live ? load_row(fa, e0) : zero_row();
      T ys[D], f1[D];
#pragma unroll
      for (int d = 0; d < D; ++d) ys[d] = y.v[d] + h0 * f0.v[d];
      rhs(sign * (t_first + (T)1.0 * h0), ys, f1);
      if (!live) continue;
#pragma unroll
      for (int d = 0; d < D; ++d) {
        const T kn = sign * f1[d];
        const T sc = (T)cp.atol + fabs(y.v[d]) * (T)cp.rtol;
        const double q = (double)((kn - f0.v[d]) / sc);       // misc.py:237
        acc.suma += q * q;
      }
    }
    ok = grid_reduce(A, acc, sh, gen++, r, n_tot);
    if (threadIdx.x == 0 && ok) {
      if (nseg > 1) { seg_counts(); controller_apply_seg(&s_c, &sh.seg, sh.seg_rec, nseg, PH_INITB, cp); }
      else { fill_record(rec, r, n_tot); controller_apply(&s_c, rec, PH_INITB, cp); }
    }
  }
  auto publish = [&](const AttemptState& st) {                // thread 0: what the next attempt needs
    sh.pub.dt = st.dt; sh.pub.t1 = st.t1; sh.pub.accepted = st.accepted; sh.pub.done = st.done;
    int j = st.next_out;                                      // speculative output range of the NEXT attempt (resolve_step)
    const double t_new = st.t1 + st.dt;
    while (j < st.n_out && !(t_out[j] > t_new)) ++j;
    sh.pub.emit_lo = st.next_out; sh.pub.emit_hi = j;
  };
  if (threadIdx.x == 0) {
    if (!ok) { s_c.status |= MI_ODE_ST_SYNC_TIMEOUT; s_c.done = 1; }
    else set_outputs_apply(&s_c, A.n_out);
    AttemptState st;
    st.load(s_c);
    st.accepted = 0;
    publish(st);
    sh.st = st;
  }
  __syncthreads();

  const T* cur_y = y_user;
  T* cur_f = fa;
  while (!uniform_i(sh.pub.done)) {                           // the adaptive loop (dopri5.py:82-121); the attempt is k_step_rowlocal's
    StepPlanes<T, S> P;
    const double dt_u = uniform_d(sh.pub.dt), t1_u = uniform_d(sh.pub.t1);
    P.y0 = cur_y; P.f0 = cur_f;
    P.y1 = (cur_y == ya) ? yb : ya;
    P.f1 = (cur_f == fa) ? fb : fa;
    P.hs = (T)dt_u; P.t0 = (T)t1_u;
    P.t_start = t1_u; P.dt64 = dt_u; P.t_new = t1_u + dt_u;
    P.j_lo = uniform_i(sh.pub.emit_lo); P.j_hi = uniform_i(sh.pub.emit_hi);
    Acc acc;
    Row y0n, f0n;                                             // the next row of this thread, loaded one iteration ahead
    auto fetch = [&](long long row) {
      long long en;
      const bool have = row < batch && locate(row, en);
      if (have) { y0n = (cur_y == y_user) ? *(const Row*)(y_user + en) : load_row(P.y0, en); f0n = load_row(P.f0, en); }
      else if constexpr (COOP) { y0n = zero_row(); f0n = zero_row(); }
    };
    fetch(first);
    for (long long row = first; row < batch; row += pitch) {
      long long e0;
      const bool live = locate(row, e0);
      const Row y0 = y0n;
      T hs = P.hs;
      asm volatile("" : "+v"(hs));                            // dt * coefficient products are formed per row, not kept (and spilled) across rows
      T k[S + 1][D];
#pragma unroll
      for (int d = 0; d < D; ++d) k[0][d] = f0n.v[d];
      fetch(row + pitch);
      T ys[D];
      auto stage = [&](auto sg_c) {
        constexpr int SG = decltype(sg_c)::value;
#pragma unroll
        for (int d = 0; d < D; ++d) {
          T kk[SG];
#pragma unroll
          for (int j = 0; j < SG; ++j) kk[j] = k[j][d];
          ys[d] = step_combine<T, SG>(y0.v[d], kk, hs, A.s);
        }
        T kn[D];
        rhs(sign * (P.t0 + (T)A.s.alpha[SG - 1] * hs), ys, kn);
#pragma unroll
        for (int d = 0; d < D; ++d) k[SG][d] = sign * kn[d];
      };
      for_stages<1, S>(stage);
      if (!live) continue;
      Row y1, f1;
#pragma unroll
      for (int d = 0; d < D; ++d) {
        T kk[S + 1];
#pragma unroll
        for (int j = 0; j <= S; ++j) kk[j] = k[j][d];
        T err, ymid;
        step_finish<T, S>(y0.v[d], kk, hs, A.s, err, ymid, !TS && P.j_hi > P.j_lo);
        if constexpr (!FSAL) ys[d] = step_y1_general<T, S>(y0.v[d], kk, hs, A.s);   // rk_common.py:55-56
        y1.v[d] = ys[d];                                      // FSAL: y1 = y_S (rk_common.py:58)
        f1.v[d] = k[S][d];
        acc.maxa = fmax(acc.maxa, (double)fabs(y0.v[d]));
        acc.maxb = fmax(acc.maxb, (double)fabs(ys[d]));
        acc.suma += (double)err * (double)err;
        step_emit<T, S, TS>(A.s, P, y0.v[d], ys[d], kk, ymid, e0 + d, t_out);
      }
      *(Row*)(P.y1 + e0) = y1;
      *(Row*)(P.f1 + e0) = f1;
    }
    ok = grid_reduce(A, acc, sh, gen++, r, n_tot);            // (its barriers also fence the reads of sh.pub above)
    if (threadIdx.x == 0) {
      AttemptState st = sh.st;
      if (!ok) { st.status |= MI_ODE_ST_SYNC_TIMEOUT; st.done = 1; st.accepted = 0; }
      else if (nseg > 1) { seg_counts(); attempt_core_seg(st, sh.seg_rec, nseg, cp, A.seg_tol ? A.seg_rtol : nullptr, A.seg_atol); }
      else { fill_record(rec, r, n_tot); attempt_core(st, rec, cp); }
      publish(st);
      sh.st = st;
    }
    __syncthreads();
    if (uniform_i(sh.pub.accepted)) { cur_y = P.y1; cur_f = P.f1; }
  }

  // final state for mi_ode_get_state: plane indices as controller_apply's rotation would have left them
  if (cur_y == y_user) {                                      // no accepted step (error exit): seed plane 0 with y0
    for (long long row = first; row < batch; row += pitch) {
      long long e0;
      if (locate(row, e0)) *(Row*)(ya + e0) = *(const Row*)(y_user + e0);
    }
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    sh.st.store(s_c);
    s_c.idx_y0 = (cur_y == yb) ? 1 : 0; s_c.idx_y1 = (cur_y == yb) ? 0 : 1;
    s_c.idx_k[0] = (cur_f == fa) ? 2 : 2 + S; s_c.idx_k[S] = (cur_f == fa) ? 2 + S : 2;
    persist_write_back(A, s_c);
  }
}

// ------------------------------------------------------------------------------------------------
// Linear RHS, dim in {16, 32, 64, 128}: the whole call in one launch on the persistent MFMA grid.
// Same hand-off and redundant controller as above; the state stays in HBM planes (a workgroup only ever touches its own
// tiles, so planes written in one attempt are re-read by the SAME workgroup in the next - sc0 loads skip its L1),
// the W slices are loaded once per call, before_integrate uses the same 16-row tile passes as k_init_linear_mfma.
// The first attempt reads y0 straight from the caller's buffer (no seed copy); y planes 0/1 and f planes 2 / 2+S
// alternate on accept exactly like the plane rotation of controller_apply.
// ------------------------------------------------------------------------------------------------
template <typename T, int D, int S, bool TS>
__global__ __launch_bounds__(D * 4) void k_persist_linear_mfma(PersistArgs A) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  __shared__ PersistShared sh;
  Ctl& s_c = sh.c;
  LinCtx<T, D> cx;
  cx.init(A.s.rhs, (T*)smem_raw, A.s.dim);
  CtrlParams cp = A.s.cp;
  cp.t_out = persist_stage_tout(A, sh.tout);
  const double* t_out = cp.t_out;
  unsigned gen = 0;
  double r[5], rec[kRec], n_tot = 0.0;
  if (threadIdx.x == 0) { persist_init_ctl(s_c, A); sh.ok = 1; }
  __syncthreads();

  T* const ya = (T*)(A.s.planes);
  T* const yb = (T*)(A.s.planes + A.s.stride);
  T* const fa = (T*)(A.s.planes + 2 * A.s.stride);
  T* const fb = (T*)(A.s.planes + (long long)(2 + S) * A.s.stride);
  const T* const y_user = (const T*)A.y0;

  bool ok;
  {
    Acc acc;
    MI_TICK(tf0);
    lin_f0_pass<T, D, true>(A.s, y_user, fa, (T*)nullptr, (T*)A.out0, cx, acc);
    MI_TICK(tf1);
    MI_TOCK(0, tf0, tf1);
    MI_SKEW(gen, 0);
    MI_HANDOFF_T0();
    ok = grid_reduce(A, acc, sh, gen++, r, n_tot);
    MI_HANDOFF_T1();
    MI_SKEW(gen - 1u, 1);
    if (threadIdx.x == 0 && ok) { fill_record(rec, r, n_tot); controller_apply(&s_c, rec, PH_F0, cp); }
    __syncthreads();
  }
  if (cp.auto_first_step && ok) {
    Acc acc;
    MI_TICK(ti0);
    lin_initb_pass<T, D, true>(A.s, y_user, fa, (T)uniform_d(s_c.h0), cx, acc);
    MI_TICK(ti1);
    MI_TOCK(1, ti0, ti1);
    MI_SKEW(gen, 0);
    MI_HANDOFF_T0();
    ok = grid_reduce(A, acc, sh, gen++, r, n_tot);
    MI_HANDOFF_T1();
    MI_SKEW(gen - 1u, 1);
    if (threadIdx.x == 0 && ok) { fill_record(rec, r, n_tot); controller_apply(&s_c, rec, PH_INITB, cp); }
  }
  // The scalar state of the loop rests in LDS between attempts (thread 0 pulls it into registers only around
  // attempt_core): the tile passes need the whole register file.
  auto publish = [&](const AttemptState& st) {                // thread 0: what the next attempt needs
    sh.pub.dt = st.dt; sh.pub.t1 = st.t1; sh.pub.accepted = st.accepted; sh.pub.done = st.done;
    int j = st.next_out;                                      // speculative output range of the NEXT attempt (resolve_step)
    const double t_new = st.t1 + st.dt;
    while (j < st.n_out && !(t_out[j] > t_new)) ++j;
    sh.pub.emit_lo = st.next_out; sh.pub.emit_hi = j;
  };
  if (threadIdx.x == 0) {
    if (!ok) { s_c.status |= MI_ODE_ST_SYNC_TIMEOUT; s_c.done = 1; }
    else set_outputs_apply(&s_c, A.n_out);
    AttemptState st;
    st.load(s_c);
    st.accepted = 0;
    publish(st);
    sh.st = st;
  }
  __syncthreads();

  const T* cur_y = y_user;
  T* cur_f = fa;
  while (!uniform_i(sh.pub.done)) {
    StepPlanes<T, S> P;
    const double dt_u = uniform_d(sh.pub.dt), t1_u = uniform_d(sh.pub.t1);
    P.y0 = cur_y; P.f0 = cur_f;
    P.y1 = (cur_y == ya) ? yb : ya;
    P.f1 = (cur_f == fa) ? fb : fa;
    P.hs = (T)dt_u; P.t0 = (T)t1_u;
    P.t_start = t1_u; P.dt64 = dt_u; P.t_new = t1_u + dt_u;
    P.j_lo = uniform_i(sh.pub.emit_lo); P.j_hi = uniform_i(sh.pub.emit_hi);
    Acc acc;
    MI_TICK(ta0);
#ifdef MI_TRACE
    if (blockIdx.x == 0 && (threadIdx.x == 0 || threadIdx.x == 256) && gen == 3u) { cx.tr = (long long*)(A.s.partials + 8192) + (threadIdx.x ? 256 : 0); cx.tn = 0; }
    else cx.tr = nullptr;
#endif
    lin_attempt_pass<T, D, S, TS, true>(A.s, P, cx, acc, t_out, (T*)sh.coef);
#ifdef MI_TRACE
    if (cx.tr != nullptr) {
      for (int q = 0; q + 3 < cx.tn && q < 4 * 14; q += 4)
        printf("[trace] wave %d eval %2d: write+barrier %5lld  mfma %5lld  barrier2 %5lld  | since prev eval end %5lld\n", (int)threadIdx.x >> 6, q / 4,
               cx.tr[q + 1] - cx.tr[q], cx.tr[q + 2] - cx.tr[q + 1], cx.tr[q + 3] - cx.tr[q + 2], q ? cx.tr[q] - cx.tr[q - 1] : 0LL);
    }
#endif
    MI_TICK(ta1);
    MI_SKEW(gen, 0);
    MI_HANDOFF_T0();
    ok = grid_reduce(A, acc, sh, gen++, r, n_tot);                   // (its barriers also fence the reads of sh.pub above)
    MI_HANDOFF_T1();
    MI_SKEW(gen - 1u, 1);
    MI_TICK(ta2);
    MI_TOCK(2, ta0, ta1); MI_TOCK(3, ta1, ta2);
    if (threadIdx.x == 0) {
      AttemptState st = sh.st;
      if (!ok) { st.status |= MI_ODE_ST_SYNC_TIMEOUT; st.done = 1; st.accepted = 0; }
      else { fill_record(rec, r, n_tot); attempt_core(st, rec, cp); }
      publish(st);
      sh.st = st;
    }
    __syncthreads();
    if (uniform_i(sh.pub.accepted)) { cur_y = P.y1; cur_f = P.f1; }
  }

  // final state for mi_ode_get_state: plane indices as controller_apply's rotation would have left them
  if (cur_y == y_user) {                                      // no accepted step (error exit): seed plane 0 with y0
    const long long n = A.s.batch * A.s.dim;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) ya[i] = y_user[i];
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    sh.st.store(s_c);
    s_c.idx_y0 = (cur_y == yb) ? 1 : 0; s_c.idx_y1 = (cur_y == yb) ? 0 : 1;
    s_c.idx_k[0] = (cur_f == fa) ? 2 : 2 + S; s_c.idx_k[S] = (cur_f == fa) ? 2 + S : 2;
    persist_write_back(A, s_c);
  }
}

template <typename T, int D>
constexpr size_t persist_linear_lds_bytes() {
  return (size_t)2 * 16 * lin_ld<T>(D) * sizeof(T);
}

// Self-test of the cross-rank hand-off: `rounds` exchanges of synthetic records through the host segment; *result = 1
// when every round delivered every rank's values intact (one workgroup, launched by every rank at the same point).
template <int UNUSED>          // (a template only so that every translation unit including this header may see it)
__global__ __launch_bounds__(64) void k_xrank_selftest(PersistArgs A, int rounds, int* result) {
  __shared__ PersistShared sh;
  if (threadIdx.x == 0) sh.ok = 1;
  __syncthreads();
  bool good = true;
  for (int round = 0; round < rounds; ++round) {
    double r[5], n_tot = 0.0;
#pragma unroll
    for (int i = 0; i < 5; ++i) r[i] = (double)(A.rank * 10 + i + round);
    cross_rank(A, sh, (unsigned)round, r, n_tot, (double)(A.rank + 1));
    if (threadIdx.x == 0) {
      const int W = A.world;
      double s0 = 0, s1 = 0, n = 0;
      for (int q = 0; q < W; ++q) { s0 += (double)(q * 10 + 2 + round); s1 += (double)(q * 10 + 3 + round); n += (double)(q + 1); }
      good = good && sh.ok && r[0] == (double)((W - 1) * 10 + round) && r[1] == (double)((W - 1) * 10 + 1 + round) && r[2] == s0 &&
             r[3] == s1 && r[4] == (double)((W - 1) * 10 + 4 + round) && n_tot == n;
    }
    __syncthreads();
    if (!sh.ok) break;
  }
  if (threadIdx.x == 0) *result = good ? 1 : 0;
}

}  // namespace mi

