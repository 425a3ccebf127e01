// Two tiles in flight per workgroup for the attempt pass of the whole-call linear kernel (k_persist_linear_mfma<.., TWO = true>).
//
// Why.  In the one-tile pass every RHS evaluation is  combine -> LDS tile -> barrier -> 32 MFMAs -> barrier  with all eight
// wavefronts of the workgroup in the same phase: while they combine and exchange the stage tile the matrix pipe idles, and the
// two wavefronts that share a SIMD cannot cover for each other (profiles/r02_config4_ablation.txt: 62 of 221 us per attempt pass).
// Here a workgroup works on TWO 16-row tiles, A and B, half an evaluation apart, with ONE barrier per evaluation:
//
//     interval 2s-1 :   M_A(s)  = the 32 MFMAs of stage s on tile A          V = E_B(s-1), C_B(s)
//     interval 2s   :   M_B(s)                                                V = E_A(s),   C_A(s+1)
//
//     E_X(s): k_{s+1} of tile X from the accumulators of M_X(s);  C_X(s): y_s = y0 + sum (dt beta_sj) k_j -> LDS stage tile X
//
// Between two barriers every wavefront owes one block of MFMAs (tile X) and one block of vector work (tile Y != X) that do not
// depend on each other; the two wavefronts of a SIMD run them in OPPOSITE order (`mfirst`), so one feeds the matrix pipe while
// the other combines, writes LDS and waits out the latencies.  The end of a tile (error, norms, y1 / f1 stores, dense output)
// and the start of the next one are vector work of the same kind and ride in the same slots.
//
// Registers: W slice 64 + tile A's k_1..k_S 48 + y0 8 + two accumulators + the prefetched next pair; tile B's k_1..k_S and y0 live
// in LDS - in wave-private slots (a lane only ever reads back what it wrote: [plane][chunk][thread] of 16-byte chunks,
// conflict-free), i.e. no barrier is involved.  k_{S+1} of either tile is consumed where it is produced.
// LDS at D = 128 fp64: two stage tiles 33 KB + 6 + 1 private planes 112 KB = 145 KB; the staging arrays of the grid hand-off
// (PersistSharedSplitT) alias the stage tiles - they are only used between passes.
//
// Arithmetic, MFMA accumulation order, element -> thread map, tile -> workgroup map and the order in which a thread
// accumulates its norms are those of lin_attempt_pass: the results are bit-identical to the one-tile pass and to fusion='step'.
#pragma once
// (included by mi_ode_persist.h, after the hand-off machinery and before the kernel that uses it)

namespace mi {

#ifdef LIN2_FULL_BARRIER
#define lds_barrier __syncthreads
#endif

// PersistSharedT with the big staging arrays (only live inside a grid hand-off) placed by the kernel, e.g. on top of buffers
// that are only live inside a tile pass.
template <int MAXG, int TOUT>
struct PersistSharedSplitT {
  static constexpr int kMaxGrid = MAXG, kTout = TOUT;
  Ctl c;
  PersistPub pub;
  AttemptState st;
  double red[80];
  double tout[TOUT];
  SegState seg;
  int ok;
  double coef[64];                                            // two-tile pass: dt * tableau products of the attempt (Lin2Coef)
  double (*vals)[MAXG];
  double (*xr)[kXMaxWorld + 1];
  double (*seg_rec)[kRec];
  static constexpr size_t kStagingBytes = sizeof(double) * ((size_t)5 * MAXG + 6 * (kXMaxWorld + 1) + kMaxSeg * kRec);
  __device__ __forceinline__ void bind(char* base) {
    vals = (double (*)[MAXG])base;
    xr = (double (*)[kXMaxWorld + 1])(base + sizeof(double) * 5 * MAXG);
    seg_rec = (double (*)[kRec])(base + sizeof(double) * (5 * MAXG + 6 * (kXMaxWorld + 1)));
  }
};

constexpr int kLin2MaxGrid = 512;
constexpr int kLin2Tout = 64;
using Lin2Shared = PersistSharedSplitT<kLin2MaxGrid, kLin2Tout>;

template <typename T, int D, int S>
struct Lin2Layout {
  using TR = MfmaTraits<T>;
  static constexpr int VEC = TR::VEC;                       // elements per 16-byte chunk
  static constexpr int NCH = 4 / VEC;                       // chunks per lane and plane (a lane owns 4 elements of a tile)
  static constexpr int NT = D * 4;
  static constexpr int LD = D + VEC;
  static constexpr size_t kTile = (size_t)16 * LD;          // elements of one stage tile
  static constexpr size_t kPriv = (size_t)4 * NT;           // elements of one wave-private plane
  static constexpr size_t kElems = 2 * kTile + (S + 1) * kPriv;     // tiles A, B | k_1..k_S of tile B | y0 of tile B
  static constexpr size_t kPassBytes = kElems * sizeof(T);
  static constexpr size_t kBytes = kPassBytes > Lin2Shared::kStagingBytes ? kPassBytes : Lin2Shared::kStagingBytes;
};

// which of the wavefronts sharing a SIMD starts an interval with its MFMAs.  order: 0 = by hardware SIMD id (every second
// wavefront of a SIMD), 1 = nobody (all waves vector-first: the two-tile schedule without the opposite order),
// 2 = wavefronts 4..7, 3 = odd wavefronts
__device__ __forceinline__ bool lin2_mfirst(int order, int* s_simd /* LDS, >= 16 ints */) {
  const int wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  bool m = false;
  if (order == 1) m = false;
  else if (order == 2) m = wave >= nw / 2;
  else if (order == 3) m = (wave & 1) != 0;
  else {
    const int simd = (int)__builtin_amdgcn_s_getreg((1 << 11) | (4 << 6) | 4) & 3;      // HW_REG_HW_ID, SIMD_ID = bits 5:4
    if ((threadIdx.x & 63) == 0) s_simd[wave] = simd;
    __syncthreads();
    int before = 0;
    for (int w = 0; w < wave; ++w) before += (s_simd[w] == simd) ? 1 : 0;
    m = (before & 1) != 0;
    __syncthreads();
  }
  return __builtin_amdgcn_readfirstlane(m ? 1 : 0) != 0;
}

// dt * coefficient products of an attempt, staged in LDS once per pass ((T)dt * (T)c, the very products step_combine /
// step_finish form): rows of beta back to back (row s-1 at s(s-1)/2), then c_error, then c_mid.  Reading one back is an LDS
// instruction; forming it in place is a v_mul_f64 (plus the scalar-register traffic of ~35 tableau entries) per use, and on this
// part every vector instruction of either wavefront of a SIMD takes its issue time away from the matrix pipe
// (scripts/micro/mfma_pair.hip, profiles/r03_mfma_pair.txt).
template <int S>
struct Lin2Coef {
  static constexpr int kBeta = 0, kErr = S * (S + 1) / 2, kMid = kErr + S + 1, kCount = kMid + S + 1;
  static constexpr int row(int s) { return s * (s - 1) / 2; }          // beta row of stage s (1-based)
};

template <typename T, int D, int S, bool SC0>
__device__ __forceinline__ void lin_attempt_pass2(const StepArgs& A, const StepPlanes<T, S>& P, LinCtx<T, D>& cx, T* lds, T* coef, Acc& acc,
                                                  const double* t_out, const bool mfirst) {
  using L = Lin2Layout<T, D, S>;
  using TR = MfmaTraits<T>;
  using acc_t = typename TR::acc_t;
  using CH = Chunk<T, L::VEC>;
  using CF = Lin2Coef<S>;
  constexpr int VEC = L::VEC, NCH = L::NCH, NT = L::NT, LD = L::LD, KS = D / 4;
  constexpr int RS = TR::RSTEP;                              // rows between a lane's consecutive accumulator elements
  T* const tileA = lds;
  T* const tileB = lds + L::kTile;
  T* const kB = tileB + L::kTile;                            // [S][NCH][NT] chunks
  T* const y0B = kB + S * L::kPriv;
  const long long ntiles = (A.batch + 15) / 16;
  const long long G = gridDim.x;
  const bool plain = A.rhs.b[0] == nullptr && A.rhs.sign == 1.0;      // (kernel arguments: a scalar branch)
  const int d = cx.d;
#ifdef LIN2_NO_FAST
  const bool wide = false;
#else
  const bool wide = d == D;
#endif
  //                                 // the state fills the tile's columns (no column mask)
  const int tid = threadIdx.x;
  const int r0 = TR::acc_row(cx.lane, 0);
  const int eoff = r0 * d + cx.col;                          // element offset of the lane's first element inside a tile of the planes
  const int tslot = r0 * LD + cx.col;                        // ... inside a stage tile
  const T hs = P.hs;

  // the coefficient table (every pass: dt changes)
  if (tid < CF::kCount) {
    double c;
    if (tid < CF::kErr) {
      int s = 1;
      while (CF::row(s + 1) <= tid) ++s;
      c = A.beta[s - 1][tid - CF::row(s)];
    } else if (tid < CF::kMid) c = A.e[tid - CF::kErr];
    else c = A.cmid[tid - CF::kMid];
    coef[tid] = hs * (T)c;
  }

  // ---- helpers ---------------------------------------------------------------------------------------------------------
  auto put_priv = [&](T* plane, const T (&v)[4]) {
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      CH ch;
#pragma unroll
      for (int e = 0; e < VEC; ++e) ch.v[e] = v[c * VEC + e];
      *(CH*)(plane + (c * NT + tid) * VEC) = ch;
    }
  };
  auto get_priv = [&](const T* plane, T (&v)[4]) {
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const CH ch = *(const CH*)(plane + (c * NT + tid) * VEC);
#pragma unroll
      for (int e = 0; e < VEC; ++e) v[c * VEC + e] = ch.v[e];
    }
  };
  auto write_tile = [&](T* tile, const T (&ys)[4]) {
    T* q = tile + tslot;
#pragma unroll
    for (int i = 0; i < 4; ++i) q[i * RS * LD] = ys[i];
  };
  auto read_tile = [&](const T* tile, T (&ys)[4]) {
    const T* q = tile + tslot;
#pragma unroll
    for (int i = 0; i < 4; ++i) ys[i] = q[i * RS * LD];
  };
  // The KS MFMA steps of LinCtx::rhs_eval on a stage tile (same operand order), software pipelined: PF operand reads in flight
  // ahead of the MFMAs that use them.  With the two wavefronts of a SIMD in opposite phases a wavefront has to keep the
  // matrix pipe fed on its own - "read, wait, two MFMAs" (what the compiler emits by itself) leaves it idle for an LDS
  // round trip out of every three.  The sched_group_barriers pin the interleave; the sched_barriers fence the block off.
  auto mfma_tile = [&](const T* tile) -> acc_t {
    constexpr int NCHK = KS / VEC, PF = NCHK < 4 ? NCHK : 4;
    acc_t c0 = {0, 0, 0, 0};
#ifdef LIN2_NO_PIPE
    { const T* ap_ = tile + cx.li * LD + cx.lg * KS;
      for (int m = 0; m < NCHK; ++m) { const CH a0 = *(const CH*)(ap_ + m * VEC); for (int v = 0; v < VEC; ++v) c0 = TR::mfma(a0.v[v], cx.bf[m * VEC + v], c0); }
      return c0; }
#endif
    const T* ap = tile + cx.li * LD + cx.lg * KS;
    __builtin_amdgcn_sched_barrier(0);
    CH a[NCHK];
#pragma unroll
    for (int m = 0; m < NCHK; ++m) a[m] = *(const CH*)(ap + m * VEC);
#pragma unroll
    for (int m = 0; m < NCHK; ++m) {
#pragma unroll
      for (int v = 0; v < VEC; ++v) c0 = TR::mfma(a[m].v[v], cx.bf[m * VEC + v], c0);
    }
    __builtin_amdgcn_sched_group_barrier(0x100, PF, 0);      // PF LDS reads
#pragma unroll
    for (int m = 0; m < NCHK - PF; ++m) {
      __builtin_amdgcn_sched_group_barrier(0x008, VEC, 0);   // the MFMAs of one chunk
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);     // the read PF chunks ahead
    }
    __builtin_amdgcn_sched_group_barrier(0x008, PF * VEC, 0);
    __builtin_amdgcn_sched_barrier(0);
    return c0;
  };
  // ... and its epilogue: bias, reversed-time sign (multiplying by +1 and skipping the bias are exact no-ops)
  auto eval_k = [&](const acc_t& c0, T (&kn)[4]) {
    if (plain) {
#pragma unroll
      for (int i = 0; i < 4; ++i) kn[i] = c0[i];
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        T k_ = c0[i];
        if (cx.has_bias) k_ = k_ + cx.bias_v;
        kn[i] = cx.sign * k_;
      }
    }
  };
  // y0 / f0 of tile t_i -> registers.  ONLY the loads: anything that touched the values here would make the compiler wait for
  // them here - the prefetch would be a blocking load.  mask4() zeroes the elements beyond the batch / the state's width when
  // the tile is adopted (a full tile needs neither clamped addresses nor a mask).
  auto fetch = [&](long long t_i, T (&y)[4], T (&f)[4]) {
    const long long row0 = t_i * 16;
    const long long left = A.batch - row0;                   // (uniform)
    if (left <= 0) return;
    const T* yb = P.y0 + row0 * d;
    const T* fb = P.f0 + row0 * d;
    if (left >= 16 && wide) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        y[i] = stream_load<SC0>(yb + eoff + i * RS * d);
        f[i] = stream_load<SC0>(fb + eoff + i * RS * d);
      }
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const bool ok = cx.colok && (long long)(r0 + i * RS) < left;
        const int off = ok ? eoff + i * RS * d : 0;
        y[i] = stream_load<SC0>(yb + off);
        f[i] = stream_load<SC0>(fb + off);
      }
    }
  };
  auto mask4 = [&](long long t_i, T (&y)[4], T (&f)[4]) {
    const long long left = A.batch - t_i * 16;
    if (left >= 16 && wide) return;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const bool ok = cx.colok && (long long)(r0 + i * RS) < left;
      y[i] = ok ? y[i] : (T)0;
      f[i] = ok ? f[i] : (T)0;
    }
  };
  // sum_{j < N} coef[c0 + j] * k_j per element, in step_combine's order: (c_0 k_0) + (c_1 k_1) + ...; getk(j, out), j < N - 1,
  // the last term's k comes in registers
  auto lincomb = [&](auto nc, int c0, auto&& getk, const T (&klast)[4], T (&a)[4]) {
    constexpr int N = decltype(nc)::value;
    for_stages<0, N - 1>([&](auto jc) {
      constexpr int J = decltype(jc)::value;
      T kj[4];
      if constexpr (J < N - 1) getk(J, kj);
#ifdef LIN2_NO_COEF
      const double* ct_ = (c0 >= CF::kMid) ? &A.cmid[c0 + J - CF::kMid] : (c0 >= CF::kErr) ? &A.e[c0 + J - CF::kErr] : &A.beta[N - 1][J];
      const T cb = hs * (T)*ct_;
#else
      const T cb = coef[c0 + J];
#endif
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const T kv = (J < N - 1) ? kj[i] : klast[i];
        a[i] = (J == 0) ? cb * kv : a[i] + cb * kv;
      }
    });
  };

  // end of a tile: err (rk_common.py:60), y1 / f1, norms.  getk(j, out): k_{j+1} of the tile, j < S; kS = k_{S+1}; y1 = y_S is
  // read back from the lane's own slots of the stage tile.  (No dense output here: an attempt that contains a requested time
  // takes the one-tile pass - same bits - so that its y_mid / quartic code does not weigh on this pass's registers.)
  auto finish = [&](long long tile_i, const T (&y0e)[4], auto&& getk, const T (&kS)[4], const T* tile) {
    const long long row0 = tile_i * 16;
    const long long left = A.batch - row0;
    if (left <= 0) return;                                   // (the odd tile of the last pair)
    T err[4], y1[4];
    lincomb(std::integral_constant<int, S + 1>{}, CF::kErr, getk, kS, err);
    read_tile(tile, y1);
    T* const y1p = P.y1 + row0 * d;
    T* const f1p = P.f1 + row0 * d;
    if (left >= 16 && wide) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        y1p[eoff + i * RS * d] = y1[i];
        f1p[eoff + i * RS * d] = kS[i];
        acc.maxa = fmax(acc.maxa, (double)fabs(y0e[i]));
        acc.maxb = fmax(acc.maxb, (double)fabs(y1[i]));
        acc.suma += (double)err[i] * (double)err[i];
      }
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (cx.colok && (long long)(r0 + i * RS) < left) {
          y1p[eoff + i * RS * d] = y1[i];
          f1p[eoff + i * RS * d] = kS[i];
          acc.maxa = fmax(acc.maxa, (double)fabs(y0e[i]));
          acc.maxb = fmax(acc.maxb, (double)fabs(y1[i]));
          acc.suma += (double)err[i] * (double)err[i];
        }
      }
    }
  };

  // ---- state -----------------------------------------------------------------------------------------------------------
  T y0A[4], kA[S][4];                                        // tile A: registers
  T py[4] = {0, 0, 0, 0}, pf[4] = {0, 0, 0, 0};              // ONE prefetch buffer, used alternately for the next A and the next B
  acc_t accA = {0, 0, 0, 0}, accB = {0, 0, 0, 0};
  const long long tA0 = blockIdx.x;
  if (tA0 >= ntiles) return;                                 // (workgroup uniform)
  fetch(tA0, py, pf);
  lds_barrier();                                             // the coefficient table

  auto adoptA = [&](long long t_i) {                         // registers <- prefetch, C_A(1)
    T ys[4];
    mask4(t_i, py, pf);
    const T cb = coef[CF::row(1)];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      y0A[i] = py[i]; kA[0][i] = pf[i];
      ys[i] = y0A[i] + cb * kA[0][i];
    }
    write_tile(tileA, ys);
  };
  auto getkA = [&](int j, T (&out)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) out[i] = kA[j][i];
  };
  auto getkB = [&](int j, T (&out)[4]) { get_priv(kB + j * (int)L::kPriv, out); };

  adoptA(tA0);
  fetch(tA0 + G, py, pf);                                    // the first B
  lds_barrier();

  // the pair loop, once per order of the two blocks of an interval (two copies: no register shuffling where the orders would meet)
  auto run = [&](auto mf_c) {
    constexpr bool MF = decltype(mf_c)::value;
    long long tA = tA0;
    bool prevB = false;
    for (;;) {
      const bool active = tA < ntiles;                       // false: only the last pair's B is left to finish
      // ---- interval 1:  M_A(1)  |  end of the previous pair's B, start of this pair's B, prefetch of the next A
      {
        auto Mp = [&]() { if (active) accA = mfma_tile(tileA); };
        auto Vp = [&]() {
          if (prevB) {
            T kn[4], yb[4];
            eval_k(accB, kn);
            get_priv(y0B, yb);
            finish(tA - G, yb, getkB, kn, tileB);
          }
          if (active) {
            T ys[4];
            mask4(tA + G, py, pf);
            put_priv(y0B, py);
            put_priv(kB, pf);
            const T cb = coef[CF::row(1)];
#pragma unroll
            for (int i = 0; i < 4; ++i) ys[i] = py[i] + cb * pf[i];
            write_tile(tileB, ys);
            fetch(tA + 2 * G, py, pf);
          }
        };
        if constexpr (MF) { Mp(); Vp(); } else { Vp(); Mp(); }
        if (!active) break;
        lds_barrier();
      }
      for_stages<1, S>([&](auto sg_c) {
        constexpr int SG = decltype(sg_c)::value;
        // ---- interval 2 SG - 1 (SG > 1):  M_A(SG)  |  E_B(SG-1), C_B(SG)
        if constexpr (SG > 1) {
          auto Mp = [&]() { accA = mfma_tile(tileA); };
          auto Vp = [&]() {
            T kn[4], a[4], yb[4];
            eval_k(accB, kn);
            put_priv(kB + (SG - 1) * (int)L::kPriv, kn);
            lincomb(std::integral_constant<int, SG>{}, CF::row(SG), getkB, kn, a);
            get_priv(y0B, yb);
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = yb[i] + a[i];
            write_tile(tileB, a);
          };
          if constexpr (MF) { Mp(); Vp(); } else { Vp(); Mp(); }
          lds_barrier();
        }
        // ---- interval 2 SG:  M_B(SG)  |  E_A(SG), C_A(SG+1)  or  end of A, start of the next pair's A, prefetch of the next B
        {
          auto Mp = [&]() { accB = mfma_tile(tileB); };
          auto Vp = [&]() {
            T kn[4];
            eval_k(accA, kn);
            if constexpr (SG < S) {
              T a[4];
#pragma unroll
              for (int i = 0; i < 4; ++i) kA[SG][i] = kn[i];
              lincomb(std::integral_constant<int, SG + 1>{}, CF::row(SG + 1), getkA, kn, a);
#pragma unroll
              for (int i = 0; i < 4; ++i) a[i] = y0A[i] + a[i];
              write_tile(tileA, a);
            } else {
              finish(tA, y0A, getkA, kn, tileA);
              if (tA + 2 * G < ntiles) adoptA(tA + 2 * G);
              fetch(tA + 3 * G, py, pf);                     // the next pair's B
            }
          };
          if constexpr (MF) { Mp(); Vp(); } else { Vp(); Mp(); }
          lds_barrier();
        }
      });
      prevB = true;
      tA += 2 * G;
    }
  };
  if (mfirst) run(std::true_type{}); else run(std::false_type{});
}

}  // namespace mi
