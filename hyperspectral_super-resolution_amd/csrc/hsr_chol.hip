// Cholesky solve of the ridge system of variant a9: (Phi_c^T Phi_c + alpha I) W = Phi_c^T Y_c, 285 x 285 float64
// with up to 285 right-hand sides (legacy_notebooks/Spectral_matching.ipynb, Ridge(alpha=1) at raw lines 475-490).
//
// The matrix is tiny for a GPU (650 KB) and the factorisation is a chain of 9 dependent panel steps, so a
// library that launches a kernel per sub-step pays mostly launch and dependency latency: rocSOLVER through
// torch.linalg took 0.93 ms of a 1.35 ms fit (potf2_kernel_small 3 x 122 us, four substitution kernels 500 us).
// Here:
//   chol_factor_kernel  ONE workgroup of 1024 threads runs the whole right-looking blocked factorisation
//                       (block 32): diagonal block factored in LDS with one barrier per column, the panel below
//                       staged in LDS and solved four lanes per row, the trailing matrix updated from the
//                       LDS-resident panel in 16 x 16 tiles on v_mfma_f64_16x16x4_f64.
//                       The inverse of every diagonal block rides along with its factorisation (r03) and is what the
//                       panel is multiplied with; the inverses also go to the workspace, so that
//   chol_solve_kernel   (forward and backward substitution, blocked: one workgroup per slab of 32 right-hand sides held
//                       in LDS) needs a 32 x 32 product with the block inverse instead of a serial 32-step chain, and
//                       updates the other rows with 16 x 16 tiles of L on v_mfma_f64_16x16x4_f64.
// n must be a multiple of 32 (the caller pads with an identity block), n <= 512.
#include <math.h>
#include <stdlib.h>

#include "hsr_common.h"

namespace hsr {

#ifdef HSR_CHOL_STAMPS          // diagnostic build only (tools/chol_stamps.hip): s_memtime per phase of the factor kernel
unsigned long long* g_chol_stamps = nullptr;   // [16 blocks][8 phases], device memory
#define CHOL_STAMP_PARAM , unsigned long long* stamps
#define CHOL_STAMP_ARG , hsr::g_chol_stamps
#define CHOL_STAMP(k) do { if (tid == 0 && stamps) { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory"); stamps[(kb / kCb) * 8 + (k)] = t_; } } while (0)
#define CHOL_STAMP_W(k) do { if (tid == 64 && stamps) { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory"); stamps[(kb / kCb) * 8 + (k)] = t_; } } while (0)
#else
#define CHOL_STAMP_W(k)
#define CHOL_STAMP_PARAM
#define CHOL_STAMP_ARG
#define CHOL_STAMP(k)
#endif
constexpr int kCb = 32;        // block size
constexpr int kCs = kCb + 1;   // LDS row stride (doubles)
constexpr int kTrailTiles = 5; // trailing-update tiles a wave loads before its first MFMA (9 = one pass at n = 288 spilled 75 VGPRs
                               // under the 128-register cap of a 1024-thread workgroup and was 35 % slower; two alternating
                               // batches of 2 / 3 / 4 tiles, the next one's loads in flight under the MFMAs: 112 k / 113 k / 136 k
                               // cycles for the phase against 105 k)

typedef double chol_f64x4 __attribute__((ext_vector_type(4)));
typedef double chol_f64x2 __attribute__((ext_vector_type(2)));
constexpr int kDs = kCb + 2;  // LDS row stride of chol_factor_res_kernel: rows start 16-byte aligned, and the 16 rows x 4 k-groups a wave reads for
                              // one MFMA operand fall on distinct banks (with 33 doubles lanes (col, kk) and (col + 1, kk - 1) share a bank: 4-way)

// 1 / d to the last bit or two: v_rcp_f64 and two Newton steps (a full IEEE division is ~3x the instructions, and the
// factorisation's column steps are latency chains)
__device__ __forceinline__ double rcp_nr(double d) {
  double r = __builtin_amdgcn_rcp(d);
  r = fma(fma(-d, r, 1.0), r, r);
  r = fma(fma(-d, r, 1.0), r, r);
  return r;
}

// Barrier that orders LDS traffic only: __syncthreads() also waits for vmcnt(0), i.e. for global loads a wave has issued
// ahead of time on purpose (chol_solve_kernel's tile of the next step).
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// sum over the 4 lanes of a quad (two DPP quad permutes per half), in every lane
__device__ __forceinline__ double quad_sum(double v) {
  v += __hiloint2double(__builtin_amdgcn_update_dpp(0, __double2hiint(v), 0xB1, 0xf, 0xf, false),
                        __builtin_amdgcn_update_dpp(0, __double2loint(v), 0xB1, 0xf, 0xf, false));
  v += __hiloint2double(__builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x4E, 0xf, 0xf, false),
                        __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x4E, 0xf, 0xf, false));
  return v;
}

// Where the first version (one panel row per thread, 4 x 4 register tiles for the trailing update) spent its 236 us
// (cycle stamps, n = 288): diagonal block 26 %, panel 30 %, trailing update 42 %.  The panel row is a 32-step chain of
// up to 31 dependent fma + a division; the update ran at 22 % of the CU's float64 rate because its 4 x 4 tiles read
// 4 bytes of LDS per fma.  Now: the panel is staged in LDS and every row is solved by FOUR lanes (each owns a quarter
// of the row's x, partial dot products joined by two DPP quad adds, reciprocals of the diagonal from LDS); the
// trailing update is 16 x 16 tiles on v_mfma_f64_16x16x4_f64 with both operands read from the LDS-resident panel
// (1 byte of LDS per fma; same peak as the vector unit on CDNA4, but reachable), a wave's tiles loaded from global
// memory up front so that their latency hides under the matrix instructions; the diagonal block multiplies by a
// Newton-refined reciprocal instead of dividing.
__global__ __launch_bounds__(1024) void chol_factor_kernel(double* __restrict__ A, int64_t lda, int n, int* info,
                                                           double* __restrict__ dinv /* [n/32][32][32]: inverses of the diagonal blocks of L */
                                                           CHOL_STAMP_PARAM) {
  extern __shared__ __attribute__((aligned(16))) double chol_lds[];
  double (*D)[kCs] = reinterpret_cast<double (*)[kCs]>(chol_lds);                      // diagonal block
  double (*P)[kCs] = reinterpret_cast<double (*)[kCs]>(chol_lds + kCb * kCs);           // panel below, (n - 32) rows
  __shared__ int bad_pivot;
  // r03: the inverse of the diagonal block rides along with its factorisation.  The column steps below are Gaussian
  // elimination with multipliers f_ij = D[i][j] / D[j][j]; applying the same row operations to the identity (E, done by the
  // threads of the upper triangle, which were idle) gives the inverse of the unit-lower factor, and L_kk^-1 =
  // diag(1 / L_ii) E.  With it the panel below is ONE small matrix product on the matrix cores (X = P L_kk^-T) instead of
  // a 32-step substitution chain per row (30 % of the kernel in r02's cycle stamps), and chol_diag_inverse_kernel - which
  // recomputed the same inverses for the substitution - is gone.
  __shared__ double E[kCb][kCs];       // E[i][c], i > c: strictly lower part of the unit-lower inverse (diagonal = 1)
  __shared__ double Minv[kCb][kCs];    // L_kk^-1, lower triangular, upper part zero
  const int tid = threadIdx.x, ti = tid >> 5, tj = tid & 31;
  const int lane = tid & 63, wave = tid >> 6;
  if (tid == 0) {
    *info = 0;
    bad_pivot = 0x7fffffff;
  }
  // tile t of the trailing update's lower triangle -> its tile row (t = bi (bi + 1) / 2 + bj); the same for every block step
  __shared__ unsigned char tile_row[480];                   // n <= 512: at most 30 tile rows = 465 tiles
  if (tid < 480) {
    int bi = (int)((sqrtf(8.0f * (float)tid + 1.0f) - 1.0f) * 0.5f);
    while ((bi + 1) * (bi + 2) / 2 <= tid) ++bi;
    while (bi * (bi + 1) / 2 > tid) --bi;
    tile_row[tid] = (unsigned char)bi;
  }
  for (int kb = 0; kb < n; kb += kCb) {
    const int m = n - kb - kCb;   // rows below the diagonal block
    CHOL_STAMP(0);
    D[ti][tj] = tj <= ti ? A[(int64_t)(kb + ti) * lda + kb + tj] : 0.0;
    E[ti][tj] = 0.0;
    for (int e = tid; e < m * kCb; e += 1024) P[e >> 5][e & 31] = A[(int64_t)(kb + kCb + (e >> 5)) * lda + kb + (e & 31)];
    __syncthreads();
    CHOL_STAMP(1);
    // 32 x 32 block, one barrier per column: the Schur update of step j is applied with the UNSCALED column j
    // (D[i][l] -= D[i][j] D[l][j] / D[j][j]); D[j][j] is then the squared pivot and the columns are scaled once at
    // the end.  (Scaling each column first needs three barriers per step: 96 instead of 33 per block.)
    // r03 cycle stamps: this loop is ~48 % of the kernel at ~620 cycles per column step.  Two restructurings were built to parity
    // and measured slower, so the loop stays as it is: two columns per barrier with a rank-2 update (1 350 cycles per pair), and
    // the next pivot's reciprocal computed one step ahead next to the update (+6 %).  The step is not bound by the reciprocal
    // chain or by the barrier count but by the LDS round trips of the 16 waves.  Later in r03 the block was moved out of LDS
    // altogether - every thread's element of (D | E^T) in a register, only the column the next step needs published (32
    // doubles, double-buffered) and read back with three loads per thread; the panel arriving by LDS-DMA meanwhile - bit-identical
    // and SLOWER: 706 cycles per step fully unrolled, 772 rolled (601 here); waves 0-3 alone with a 2 x 2 register tile per
    // thread: 880.  The step's floor is the write -> barrier -> read round trip plus the reciprocal chain, not the LDS volume.
    for (int j = 0; j < kCb - 1; ++j) {
      if (ti > j && tj > j && tj <= ti) {
        D[ti][tj] -= D[ti][j] * D[tj][j] * rcp_nr(D[j][j]);
      } else if (tj > j && ti <= j) {                 // upper-triangle thread (ti, tj): element E[tj][ti], updated for ti <= j < tj
        const double f = D[tj][j] * rcp_nr(D[j][j]);  // multiplier of row tj in step j (column j is final by now)
        E[tj][ti] -= f * (ti == j ? 1.0 : E[j][ti]);  // row j of E is final (only steps < j touch it)
      }
      __syncthreads();
    }
    CHOL_STAMP(2);
    double piv = 1.0, rpiv_row = 1.0;
    if (tj <= ti) {
      const double d = D[tj][tj];
      if (ti == tj && !(d > 0.0)) atomicMin(&bad_pivot, kb + tj + 1);   // LAPACK: index of the first non-positive pivot
      piv = sqrt(d);
      rpiv_row = 1.0 / sqrt(D[ti][ti]);
    }
    __syncthreads();
    if (tid == 0 && *info == 0 && bad_pivot != 0x7fffffff) *info = bad_pivot;
    {
      double mi = 0.0;
      if (tj <= ti) {
        const double l = ti == tj ? piv : D[ti][tj] / piv;
        A[(int64_t)(kb + ti) * lda + kb + tj] = l;
        mi = (ti == tj ? 1.0 : E[ti][tj]) * rpiv_row;                   // L_kk^-1 = diag(1 / L_ii) E
      }
      Minv[ti][tj] = mi;
      dinv[((size_t)(kb / kCb) * kCb + ti) * kCb + tj] = mi;
    }
    __syncthreads();
    CHOL_STAMP(3);
    // panel below: X = P L_kk^-T, i.e. X[r][c] = sum_k P[r][k] Minv[c][k], on the float64 matrix cores.  One wave per 16 rows
    // does both 16-column halves: all of its A operands are in registers before the first result is written back to P.
    {
      const int col = lane & 15, kk = lane >> 4;
      for (int I = wave; I < (m >> 4); I += 16) {
        double av[kCb / 4];
#pragma unroll
        for (int st = 0; st < kCb / 4; ++st) av[st] = P[16 * I + col][4 * st + kk];
        chol_f64x4 x0 = {0.0, 0.0, 0.0, 0.0}, x1 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int st = 0; st < kCb / 4; ++st) {
          x0 = __builtin_amdgcn_mfma_f64_16x16x4f64(av[st], Minv[col][4 * st + kk], x0, 0, 0, 0);
          x1 = __builtin_amdgcn_mfma_f64_16x16x4f64(av[st], Minv[16 + col][4 * st + kk], x1, 0, 0, 0);
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          P[16 * I + kk + 4 * g][col] = x0[g];
          P[16 * I + kk + 4 * g][16 + col] = x1[g];
        }
      }
    }
    __syncthreads();
    CHOL_STAMP(4);
    for (int e = tid; e < m * kCb; e += 1024) A[(int64_t)(kb + kCb + (e >> 5)) * lda + kb + (e & 31)] = P[e >> 5][e & 31];
    CHOL_STAMP(5);
    // trailing update, lower triangle in 16 x 16 tiles on the float64 matrix cores: C[I][J] -= P_I P_J^T.
    // Lane (col = lane & 15, kk = lane >> 4): A operand P[16 I + col][4 s + kk], B operand P[16 J + col][4 s + kk],
    // accumulator register g = element (row kk + 4 g, column col) of the tile (layout as in csrc/hsr_ridge.hip).
    {
      const int mt = m >> 4;                       // m is a multiple of 32
      const int ntile = mt * (mt + 1) / 2;
      const int col = lane & 15, kk = lane >> 4;
      for (int t0 = wave; t0 < ntile; t0 += 16 * kTrailTiles) {
        double creg[kTrailTiles][4];
        int bis[kTrailTiles], bjs[kTrailTiles];
#pragma unroll
        for (int u = 0; u < kTrailTiles; ++u) {
          const int t = t0 + 16 * u;
          const int bi = tile_row[t < ntile ? t : 0];      // table built once per launch (r03: a float64 sqrt and two
          bis[u] = bi;                                     // correction loops per tile were part of the 39 % this phase took)
          bjs[u] = (t < ntile ? t : 0) - bi * (bi + 1) / 2;
          if (t < ntile) {
            const double* src = A + (int64_t)(kb + kCb + 16 * bi + kk) * lda + kb + kCb + 16 * bjs[u] + col;
#pragma unroll
            for (int g = 0; g < 4; ++g) creg[u][g] = src[(int64_t)(4 * g) * lda];
          }
        }
#pragma unroll
        for (int u = 0; u < kTrailTiles; ++u) {
          const int t = t0 + 16 * u;
          if (t < ntile) {
            chol_f64x4 acc = {0.0, 0.0, 0.0, 0.0};
            const double* pa = &P[16 * bis[u] + col][kk];
            const double* pb = &P[16 * bjs[u] + col][kk];
#pragma unroll
            for (int st = 0; st < kCb / 4; ++st) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(pa[4 * st], pb[4 * st], acc, 0, 0, 0);
            double* dst = A + (int64_t)(kb + kCb + 16 * bis[u] + kk) * lda + kb + kCb + 16 * bjs[u] + col;
#pragma unroll
            for (int g = 0; g < 4; ++g) dst[(int64_t)(4 * g) * lda] = creg[u][g] - acc[g];
          }
        }
      }
    }
    __syncthreads();
    CHOL_STAMP(6);
  }
}

// ---- r04: the 32 x 32 diagonal block on ONE wave, four columns per step, rank-4 updates on the matrix cores -------------------------
// r03 stamps: 47 % of the factorisation in the 31 column steps of a block (600 cycles each: LDS write -> barrier -> read of 16 waves
// plus a reciprocal chain), and every restructuring that kept the step inside LDS and barriers measured slower.  Here the block
// never meets a barrier: wave 0 holds its three 16 x 16 tiles (and the three of the inverse) as MFMA accumulators and takes FOUR
// columns per step:
//   S    = the 4 x 4 pivot block, read out of the accumulators with v_readlane; every lane runs its LDL^T (unit-lower M, pivots d;
//          four reciprocal chains instead of 4 x 16 waves of them) and has L_S^-1 = diag(d^-1/2) M^-1 in uniform registers;
//   Lp   = a L_S^-T for the rows below (a = the four raw columns, which cross from accumulator to operand layout through LDS - the
//          one transposition of the step; lane (i, k) reads row i's four values and combines them with its column k of L_S^-T);
//   C   -= Lp Lp^T on the three tiles: one v_mfma_f64_16x16x4 each (rows already finished are masked out of the A operand);
//   X    = L_S^-1 (rows of the unit-lower inverse E so far): ONE MFMA - the B operand is accumulator register s mod 4 of the E
//          tile as it stands, and the result lands in the B-operand layout of the next product; X is rows 4 s .. 4 s + 3 of
//          L_kk^-1 (-> Minv); E -= Lp X on the tiles below.
// No sqrt or division on the way of a step except the chain of the pivot block itself.
__device__ __forceinline__ double lane_bcast(double v, int l) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}

// a wave-uniform double into scalar registers (two v_readfirstlane_b32): the step's ten uniform results would otherwise sit in
// twenty vector registers next to 136 of resident tiles and 48 of accumulators
__device__ __forceinline__ double uni(double v) {
  return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)), __builtin_amdgcn_readfirstlane(__double2loint(v)));
}

struct DiagTiles {
  chol_f64x4 c00, c10, c11, e00, e10, e11;
};

// 1 / sqrt(d) to the last bit or two: v_rsq_f64 and two Newton steps (sqrt + reciprocal are ~3x the instructions)
__device__ __forceinline__ double rsqrt_nr(double d) {
  double y = __builtin_amdgcn_rsq(d);
  y = fma(0.5 * y, fma(-(d * y), y, 1.0), y);
  y = fma(0.5 * y, fma(-(d * y), y, 1.0), y);
  return y;
}

// what step s leaves for its inverse side, which runs INSIDE step s + 1 (the 4 x 4 unit-lower inverse as an MFMA operand, the lane's
// d^-1/2, the two masked -Lp operands)
struct DiagPend {
  double Li, rsk, A0, A1;
};

// Step S = 0 .. 7 does the Cholesky side of columns 4 S .. 4 S + 3 and, in two places, the inverse side of step S - 1 (S = 8: only
// that).  One step behind because `x = d^-1/2 y` needs the MFMA that makes y, queued behind the three of the Cholesky side, and the
// wave waited for it in step order (worth ~40 cycles of a step's ~1 450: 12.0 k -> 11.7 k per block).  Now y of step S - 1 is issued right after step S's pivot block has left the
// accumulators and completes under step S's reciprocal chain; x, the rows of L_kk^-1 and the two updates of the inverse's tiles follow
// step S's own three MFMAs and run under the NEXT chain.
template <int S>
__device__ __forceinline__ void diag_step(DiagTiles& t, double (*D)[kDs], double (*Minv)[kDs], int col, int kk, int kb, int& bad,
                                          DiagPend& pe) {
  constexpr bool HAS_D = S < 8, HAS_E = S >= 1;
  constexpr int SD = HAS_D ? S : 7;                              // (constants of a step that is not there are never used)
  constexpr int j0 = 4 * SD, Jp = SD / 4, q0 = j0 % 16, g0 = SD % 4;
  constexpr int SE = HAS_E ? S - 1 : 0;
  constexpr int ej0 = 4 * SE, eJp = SE / 4, eg0 = SE % 4;
  double s00 = 0, s10 = 0, s11 = 0, s20 = 0, s21 = 0, s22 = 0, s30 = 0, s31 = 0, s32 = 0, s33 = 0;
  chol_f64x2 a0l = {0.0, 0.0}, a0h = {0.0, 0.0}, a1l = {0.0, 0.0}, a1h = {0.0, 0.0};
  if (HAS_D) {
    const chol_f64x4& dt = Jp == 0 ? t.c00 : t.c11;
    // pivot block, lower triangle: element (a, b) sits in lane (col = q0 + b, kk = a), register g0 of the diagonal tile
    s00 = lane_bcast(dt[g0], q0), s10 = lane_bcast(dt[g0], q0 + 16), s11 = lane_bcast(dt[g0], q0 + 17);
    s20 = lane_bcast(dt[g0], q0 + 32), s21 = lane_bcast(dt[g0], q0 + 33), s22 = lane_bcast(dt[g0], q0 + 34);
    s30 = lane_bcast(dt[g0], q0 + 48), s31 = lane_bcast(dt[g0], q0 + 49), s32 = lane_bcast(dt[g0], q0 + 50);
    s33 = lane_bcast(dt[g0], q0 + 51);
    // the four raw columns go to their places in D (step 0: still there from the deposit)
    if (SD > 0 && (col >> 2) == g0) {
      if (Jp == 0) {
#pragma unroll
        for (int g = g0; g < 4; ++g) D[kk + 4 * g][col] = t.c00[g];
#pragma unroll
        for (int g = 0; g < 4; ++g) D[16 + kk + 4 * g][col] = t.c10[g];
      } else {
#pragma unroll
        for (int g = g0; g < 4; ++g) D[16 + kk + 4 * g][16 + col] = t.c11[g];
      }
    }
    __builtin_amdgcn_wave_barrier();
    // lane (i = col, k = kk) reads the four raw values of rows col and 16 + col (16-byte loads: rows of 34 doubles)
    if (SD <= 3) {
      a0l = *reinterpret_cast<const chol_f64x2*>(&D[col][j0]);
      a0h = *reinterpret_cast<const chol_f64x2*>(&D[col][j0 + 2]);
    }
    a1l = *reinterpret_cast<const chol_f64x2*>(&D[16 + col][j0]);
    a1h = *reinterpret_cast<const chol_f64x2*>(&D[16 + col][j0 + 2]);
  }
  // inverse side of the previous step, first half: Y = M^-1 E[rows of that step][:] - A operand lane (col = k', kk = a) =
  // (M^-1)[k'][a] (unit lower), rows k' >= 4 zero; X = diag(d^-1/2) Y
  const chol_f64x4 zero = {0.0, 0.0, 0.0, 0.0};
  chol_f64x4 y0 = zero, y1 = zero;
  if (HAS_E) {
    y0 = __builtin_amdgcn_mfma_f64_16x16x4f64(pe.Li, (eJp == 0 ? t.e00 : t.e10)[eg0], zero, 0, 0, 0);
    if (eJp == 1) y1 = __builtin_amdgcn_mfma_f64_16x16x4f64(pe.Li, t.e11[eg0], zero, 0, 0, 0);
  }
  __builtin_amdgcn_sched_barrier(0);
  DiagPend next = pe;
  if (HAS_D) {
    // LDL^T of the pivot block (unit lower M, pivots d), every lane the same
    const double d0 = s00, r0 = rcp_nr(d0);
    const double m10 = s10 * r0, m20 = s20 * r0, m30 = s30 * r0;
    const double d1 = fma(-m10, s10, s11), r1 = rcp_nr(d1);
    const double u21 = fma(-m20, s10, s21), u31 = fma(-m30, s10, s31);
    const double m21 = u21 * r1, m31 = u31 * r1;
    const double d2 = fma(-m21, u21, fma(-m20, s20, s22)), r2 = rcp_nr(d2);
    const double u32 = fma(-m31, u21, fma(-m30, s20, s32));
    const double m32 = u32 * r2;
    const double d3 = fma(-m32, u32, fma(-m31, u31, fma(-m30, s30, s33)));
    if (!(d0 > 0.0 && d1 > 0.0 && d2 > 0.0 && d3 > 0.0)) {          // wave-uniform, rare
      const int first = !(d0 > 0.0) ? 1 : (!(d1 > 0.0) ? 2 : (!(d2 > 0.0) ? 3 : 4));
      bad = min(bad, kb + j0 + first);
    }
    const double rsk = rsqrt_nr(kk == 0 ? d0 : (kk == 1 ? d1 : (kk == 2 ? d2 : d3)));   // d_k^-1/2 of the lane's column k
    // t = a M^-T by forward substitution (all four), the lane keeps t_k: Lp[i][k] = t_k d_k^-1/2
    double lp0 = 0.0;
    if (SD <= 3) {
      const double t0 = a0l[0], t1 = fma(-m10, t0, a0l[1]), t2 = fma(-m21, t1, fma(-m20, t0, a0h[0]));
      const double t3 = fma(-m32, t2, fma(-m31, t1, fma(-m30, t0, a0h[1])));
      lp0 = rsk * (kk == 0 ? t0 : (kk == 1 ? t1 : (kk == 2 ? t2 : t3)));
      if (col >= j0 + kk) D[col][j0 + kk] = lp0;
    }
    double lp1;
    {
      const double t0 = a1l[0], t1 = fma(-m10, t0, a1l[1]), t2 = fma(-m21, t1, fma(-m20, t0, a1h[0]));
      const double t3 = fma(-m32, t2, fma(-m31, t1, fma(-m30, t0, a1h[1])));
      lp1 = rsk * (kk == 0 ? t0 : (kk == 1 ? t1 : (kk == 2 ? t2 : t3)));
      if (16 + col >= j0 + kk) D[16 + col][j0 + kk] = lp1;
    }
    const double A0 = col > j0 + 3 ? -lp0 : 0.0;                  // rows that are finished take no part
    const double A1 = 16 + col > j0 + 3 ? -lp1 : 0.0;
    if (SD <= 2) {
      t.c00 = __builtin_amdgcn_mfma_f64_16x16x4f64(A0, lp0, t.c00, 0, 0, 0);
      t.c10 = __builtin_amdgcn_mfma_f64_16x16x4f64(A1, lp0, t.c10, 0, 0, 0);
    }
    if (SD <= 6) t.c11 = __builtin_amdgcn_mfma_f64_16x16x4f64(A1, lp1, t.c11, 0, 0, 0);
    // the 4 x 4 unit-lower inverse as the operand of the NEXT step's first MFMA
    const double n10 = -m10, n21 = -m21, n32 = -m32;
    const double n20 = fma(m21, m10, -m20), n31 = fma(m32, m21, -m31);
    const double n30 = -fma(n32, m20, fma(n31, m10, m30));
    const int lane = col + 16 * kk;
    // (built with selects: six v_writelane_b32 through inline assembly - this compiler has no builtin for it - gave an inverse that was
    // off by 1e-9, the low words of single entries; not understood, not used)
    double Li = (lane == 0 || lane == 17 || lane == 34 || lane == 51) ? 1.0 : 0.0;
    Li = lane == 1 ? n10 : Li;
    Li = lane == 2 ? n20 : Li;
    Li = lane == 3 ? n30 : Li;
    Li = lane == 18 ? n21 : Li;
    Li = lane == 19 ? n31 : Li;
    Li = lane == 35 ? n32 : Li;
    next = DiagPend{Li, rsk, A0, A1};
  }
  __builtin_amdgcn_sched_barrier(0);
  if (HAS_E) {
    // inverse side of the previous step, second half
    const double x0 = pe.rsk * y0[0];                             // lane (c, k): X[ej0 + k][c]
    Minv[ej0 + kk][col] = x0;                                     // rows ej0 .. ej0 + 3 of L_kk^-1
    const double x1 = eJp == 1 ? pe.rsk * y1[0] : 0.0;
    Minv[ej0 + kk][16 + col] = x1;
    if (SE <= 2) t.e00 = __builtin_amdgcn_mfma_f64_16x16x4f64(pe.A0, x0, t.e00, 0, 0, 0);
    if (SE <= 6) t.e10 = __builtin_amdgcn_mfma_f64_16x16x4f64(pe.A1, x0, t.e10, 0, 0, 0);
    if (SE >= 4 && SE <= 6) t.e11 = __builtin_amdgcn_mfma_f64_16x16x4f64(pe.A1, x1, t.e11, 0, 0, 0);
  }
  pe = next;
}

// Wave 0 of the workgroup: D (lower triangle, upper zero) -> L_kk in D's lower triangle, L_kk^-1 in Minv
__device__ __forceinline__ void diag_block_wave(double (*D)[kDs], double (*Minv)[kDs], int lane, int kb, int* bad_pivot) {
  const int col = lane & 15, kk = lane >> 4;
  DiagTiles t;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    t.c00[g] = D[kk + 4 * g][col];
    t.c10[g] = D[16 + kk + 4 * g][col];
    t.c11[g] = D[16 + kk + 4 * g][16 + col];
    t.e00[g] = kk + 4 * g == col ? 1.0 : 0.0;
    t.e10[g] = 0.0;
    t.e11[g] = t.e00[g];
  }
  int bad = 0x7fffffff;
  DiagPend pe{0.0, 0.0, 0.0, 0.0};
  diag_step<0>(t, D, Minv, col, kk, kb, bad, pe);
  diag_step<1>(t, D, Minv, col, kk, kb, bad, pe);
  diag_step<2>(t, D, Minv, col, kk, kb, bad, pe);
  diag_step<3>(t, D, Minv, col, kk, kb, bad, pe);
  diag_step<4>(t, D, Minv, col, kk, kb, bad, pe);
  diag_step<5>(t, D, Minv, col, kk, kb, bad, pe);
  diag_step<6>(t, D, Minv, col, kk, kb, bad, pe);
  diag_step<7>(t, D, Minv, col, kk, kb, bad, pe);
  diag_step<8>(t, D, Minv, col, kk, kb, bad, pe);           // the inverse side of step 7
  if (lane == 0 && bad != 0x7fffffff) atomicMin(bad_pivot, bad);
}

// ---- r04, n <= 288: the trailing matrix never returns to memory -----------------------------------------------------------------
// r03 stamps of the kernel above: 30 % in the trailing update (load a tile, 8 MFMAs, store it: all L2 latency) and 8 % loading D and
// P.  Here the 16 x 16 tiles of the lower triangle behind the first block column - 136 at n = 288 - are dealt out to the waves in
// column-major order (tile t to wave t mod 8: the tiles still alive at any block step are spread evenly) and live in their owners'
// registers as MFMA accumulators.  When a tile's block column comes up its owner writes it into D / P in LDS.  136 tiles are 72 registers per thread of a 1024-thread workgroup, which the 128
// registers of such a thread do not have beside the rest of the kernel (100 spilled); 8 waves have 256 each.
constexpr int kResWaves = 8;
constexpr int kResThreads = 64 * kResWaves;
constexpr int kResMaxN = 288;
constexpr int kResTiles = 20;     // ceil(136 / 7): wave 0 owns none

__global__ __launch_bounds__(kResThreads) void chol_factor_res_kernel(double* __restrict__ A, int64_t lda, int n, int* info,
                                                                      double* __restrict__ dinv CHOL_STAMP_PARAM) {
  extern __shared__ __attribute__((aligned(16))) double chol_lds[];
  // two (D | P) buffers, block columns alternate: buffer 0 has n rows, buffer 1 n - 32 (the columns only get shorter)
  typedef double (*Rows)[kDs];
  Rows buf0 = reinterpret_cast<Rows>(chol_lds), buf1 = reinterpret_cast<Rows>(chol_lds + (size_t)n * kDs);
  __shared__ int bad_pivot;
  __shared__ __attribute__((aligned(16))) double Minv[kCb][kDs];    // L_kk^-1
  constexpr int kWorkers = kResWaves - 1, kWorkThreads = 64 * kWorkers;
  const int tid = threadIdx.x, ti = tid >> 5, tj = tid & 31;                            // thread (ti, tj): elements (ti, tj), (ti + 16, tj)
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int col = lane & 15, kk = lane >> 4;
  __shared__ unsigned short res_tile[144];                  // resident tile t: bi | bj << 8
  const int mt = n >> 4, nres = mt > 2 ? (mt - 2) * (mt - 1) / 2 : 0;
  if (tid == 0) {
    *info = 0;
    bad_pivot = 0x7fffffff;
  }
  if (tid < nres) {
    int t = tid, bj = 2;
    while (t >= mt - bj) {
      t -= mt - bj;
      ++bj;
    }
    res_tile[tid] = (unsigned short)((bj + t) | (bj << 8));
  }
  // loads in the order of need (they return in order): the first diagonal block, the first panel (the workers; wave 0 is busy with
  // the block), then the workers' resident tiles by slot
  double d_in[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) d_in[h] = tj <= ti + 16 * h ? A[(int64_t)(ti + 16 * h) * lda + tj] : 0.0;
  {
    const int kb = 0;
    (void)kb;
    CHOL_STAMP(0);
  }
  // one 16-row tile of the panel times L_kk^-T, both 16-column halves
  // (k runs over a lane's PAIRS: lane (col, kk) supplies k = 8 p + 2 kk, 8 p + 2 kk + 1 to MFMAs 2 p, 2 p + 1 of a chain - for both
  // operands, so the sum is the same - and reads them with one 16-byte load)
  auto panel_tile = [&](Rows P, int I) {
    chol_f64x2 av[kCb / 8], b0[kCb / 8], b1[kCb / 8];
#pragma unroll
    for (int p = 0; p < kCb / 8; ++p) {
      av[p] = *reinterpret_cast<const chol_f64x2*>(&P[16 * I + col][8 * p + 2 * kk]);
      b0[p] = *reinterpret_cast<const chol_f64x2*>(&Minv[col][8 * p + 2 * kk]);
      b1[p] = *reinterpret_cast<const chol_f64x2*>(&Minv[16 + col][8 * p + 2 * kk]);
    }
    chol_f64x4 x0 = {0.0, 0.0, 0.0, 0.0}, x1 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int p = 0; p < kCb / 8; ++p) {
      x0 = __builtin_amdgcn_mfma_f64_16x16x4f64(av[p][0], b0[p][0], x0, 0, 0, 0);
      x1 = __builtin_amdgcn_mfma_f64_16x16x4f64(av[p][0], b1[p][0], x1, 0, 0, 0);
      x0 = __builtin_amdgcn_mfma_f64_16x16x4f64(av[p][1], b0[p][1], x0, 0, 0, 0);
      x1 = __builtin_amdgcn_mfma_f64_16x16x4f64(av[p][1], b1[p][1], x1, 0, 0, 0);
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      P[16 * I + kk + 4 * g][col] = x0[g];
      P[16 * I + kk + 4 * g][16 + col] = x1[g];
    }
  };
  // what every wave does at the top of block step kb: L_kk and its inverse to memory, then its share of the panel's row tiles
  auto block_out_and_panel = [&](Rows D, Rows P, int kb, int m) {
    if (tid == 0 && *info == 0 && bad_pivot != 0x7fffffff) *info = bad_pivot;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int r = ti + 16 * h;
      if (tj <= r) A[(int64_t)(kb + r) * lda + kb + tj] = D[r][tj];
      dinv[((size_t)(kb / kCb) * kCb + r) * kCb + tj] = Minv[r][tj];
    }
    for (int I = wave; I < (m >> 4); I += kResWaves) panel_tile(P, I);
  };
  if (wave == 0) {
    // ---- wave 0: the diagonal blocks.  It owns no tile, so that nothing but a block's six accumulator tiles is live in its registers
    __syncthreads();                                         // (the workers wait for their loads of the first panel here)
    {
      const int kb = 23 * kCb;                              // (stamps 189 .. 191: block 0's slots 5, 6 belong to the worker's stamps)
      (void)kb;
      CHOL_STAMP(5);
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) buf0[ti + 16 * h][tj] = d_in[h];
    lds_barrier();
    {
      const int kb = 23 * kCb;                              // (stamps 189 .. 191: block 0's slots 5, 6 belong to the worker's stamps)
      (void)kb;
      CHOL_STAMP(6);
    }
    diag_block_wave(buf0, Minv, lane, 0, &bad_pivot);
    {
      const int kb = 23 * kCb;                              // (stamps 189 .. 191: block 0's slots 5, 6 belong to the worker's stamps)
      (void)kb;
      CHOL_STAMP(7);
    }
    lds_barrier();
    for (int kb = 0; kb < n; kb += kCb) {
      const int m = n - kb - kCb;
      Rows D = ((kb >> 5) & 1) ? buf1 : buf0, Dn = ((kb >> 5) & 1) ? buf0 : buf1;
      CHOL_STAMP(1);
      block_out_and_panel(D, D + kCb, kb, m);
      lds_barrier();
      CHOL_STAMP(2);
      if (m == 0) break;
      Dn[lane >> 2][16 + 4 * (lane & 3) + 0] = 0.0;          // the tile above the next block's diagonal belongs to nobody
      Dn[lane >> 2][16 + 4 * (lane & 3) + 1] = 0.0;
      Dn[lane >> 2][16 + 4 * (lane & 3) + 2] = 0.0;
      Dn[lane >> 2][16 + 4 * (lane & 3) + 3] = 0.0;
      lds_barrier();
      CHOL_STAMP(3);
      diag_block_wave(Dn, Minv, lane, kb + kCb, &bad_pivot);
      lds_barrier();
    }
    return;
  }
  // ---- waves 1 .. 7: the panel's loads, the resident tiles
  const int wt = tid - 64, ww = wave - 1;
  constexpr int kPanelLoads = ((kResMaxN - kCb) * kCb + kWorkThreads - 1) / kWorkThreads;
  double p_in[kPanelLoads];
#pragma unroll
  for (int i = 0; i < kPanelLoads; ++i) {
    const int e = wt + i * kWorkThreads;
    p_in[i] = e < (n - kCb) * kCb ? A[(int64_t)(kCb + (e >> 5)) * lda + (e & 31)] : 0.0;
  }
  __syncthreads();                                           // waits for the loads of the first block and panel
#pragma unroll
  for (int h = 0; h < 2; ++h) buf0[ti + 16 * h][tj] = d_in[h];
#pragma unroll
  for (int i = 0; i < kPanelLoads; ++i) {
    const int e = wt + i * kWorkThreads;
    if (e < (n - kCb) * kCb) buf0[kCb + (e >> 5)][e & 31] = p_in[i];
  }
  lds_barrier();                                             // wave 0 starts on block 0
  // the resident tiles: 80 loads per thread, in flight under block 0 and the first panel product.  (Issued BEFORE the panel's loads
  // they held its way into LDS up - a wait can only name the 63 youngest loads - and so did negating them on arrival: the first
  // block took 59 k and 38 k cycles instead of 19 k.)
  chol_f64x4 creg[kResTiles];
  int tij[kResTiles];
#pragma unroll
  for (int u = 0; u < kResTiles; ++u) {
    const int t = ww + kWorkers * u;
    tij[u] = t < nres ? __builtin_amdgcn_readfirstlane((int)res_tile[t]) : -1;
    creg[u] = chol_f64x4{0.0, 0.0, 0.0, 0.0};
    if (tij[u] >= 0) {
      const double* src = A + (int64_t)(16 * (tij[u] & 255) + kk) * lda + 16 * (tij[u] >> 8) + col;
#pragma unroll
      for (int g = 0; g < 4; ++g) creg[u][g] = src[(int64_t)(4 * g) * lda];   // (no arithmetic on them here: the wave would wait)
    }
  }
  lds_barrier();
  // Which slots a phase has to visit is asked ONCE per phase, by the lanes: lane u holds slot u's tile, the phase's condition is one
  // vector compare and a ballot, and every slot's test is one scalar bit test (the slots are registers: static indices only; a
  // while-loop over the set bits with a switch on the index put all 160 registers into scratch).
  // Walking all 20 slots with scalar compares - the tile codes live in spilled SGPRs, 15 instructions per slot - cost 1.4 k cycles per
  // walk, three walks per step; in phase (2), where a worker has one tile or none, that was most of its 3.5 k cycles.
  const int slot_tile = (lane < kResTiles && ww + kWorkers * lane < nres) ? (int)res_tile[ww + kWorkers * lane] : -1;
  const int slot_bi = slot_tile & 255, slot_bj = slot_tile >> 8;
  auto for_slots = [&](unsigned long long mask, auto&& f) __attribute__((always_inline)) {
#pragma unroll
    for (int u = 0; u < kResTiles; ++u)
      if (mask & (1ull << u)) f(creg[u], tij[u]);            // one scalar bit test per slot
  };
  // Block step kb: D holds L_kk and Minv its inverse, P the raw panel.  (1) L_kk and the inverse go to memory, the panel is
  // multiplied by L_kk^-T; (2) the three tiles of the NEXT diagonal block are updated and leave the registers for the other buffer;
  // (3) while wave 0 factors that block, the workers store the panel, update the next panel's tiles and put them beside it, and
  // update the rest of the trailing matrix.  (Measured and dropped: the three tiles of a diagonal block on ONE worker, which also
  // multiplies the panel's first two row tiles and so has the next block in LDS when the panel is - phase (2) and its barrier gone -
  // made that worker's 4.5 k cycles of matrix work the length of phase (1) in every step: 154 us against 134.)
  // (the loop is rotated - phase (1) of step 0 stands in front of it, phase (1) of step k + 1 ends iteration k - so that the resident
  // tiles can be "touched" once between the first panel product and the first update: the compiler then waits for their loads THERE.
  // Left to itself it put s_waitcnt vmcnt(2) in front of a tile's first MFMA inside the loop, where in every later step it waits for
  // nothing but the step's own stores of L_kk and the panel - 3.5 k cycles per step in phase (2), the diagonal tile's 1 k aside.)
  block_out_and_panel(buf0, buf0 + kCb, 0, n - kCb);
  lds_barrier();
#pragma unroll
  for (int u = 0; u < kResTiles; ++u)
#pragma unroll
    for (int g = 0; g < 4; ++g) asm volatile("" : "+v"(creg[u][g]));
  for (int kb = 0; kb < n - kCb; kb += kCb) {
    const int m = n - kb - kCb;   // rows below the diagonal block (> 0)
    Rows D = ((kb >> 5) & 1) ? buf1 : buf0, P = D + kCb;
    Rows Dn = ((kb >> 5) & 1) ? buf0 : buf1, Pn = Dn + kCb;
    const int k2 = (kb >> 4) + 2;                   // first tile row / column behind this block column
    // trailing update of resident tiles: C[I][J] -= P_I P_J^T.  Lane (col, kk): A operand -P[16 I + col][k], B operand
    // P[16 J + col][k] (k in pairs as above), accumulator register g = element (row kk + 4 g, column col) of the tile.
    // (Two tiles at a time with alternating chains - so that one wave alone keeps the matrix pipe busy - needs the operands of both:
    // the kernel went from 244 registers to 256 + scratch and from 134 to 155 us.)
    auto update = [&](chol_f64x4& acc, int bi, int bj) __attribute__((always_inline)) {
      const chol_f64x2* pa = reinterpret_cast<const chol_f64x2*>(&P[16 * (bi - k2) + col][2 * kk]);
      const chol_f64x2* pb = reinterpret_cast<const chol_f64x2*>(&P[16 * (bj - k2) + col][2 * kk]);
      chol_f64x2 va[kCb / 8], vb[kCb / 8];          // (filled here, not in a helper taking the arrays by reference: that left them in scratch)
#pragma unroll
      for (int p = 0; p < kCb / 8; ++p) {
        va[p] = -pa[4 * p];
        vb[p] = pb[4 * p];
      }
      // all eight loads before the first MFMA: left alone, the compiler reuses two operand registers and runs load - wait - two MFMAs
      // four times over (a tile's update 1.5 k cycles, the three tiles of phase (2) 3.5 k)
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int p = 0; p < kCb / 8; ++p) {
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(va[p][0], vb[p][0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(va[p][1], vb[p][1], acc, 0, 0, 0);
      }
    };
    (void)D;
    CHOL_STAMP_W(4);
    // (2) the next diagonal block's tiles
    const bool in_col = slot_tile >= 0 && (slot_bj >> 1) == (k2 >> 1);
    for_slots(__ballot(in_col && slot_bi < k2 + 2), [&](chol_f64x4& acc, int t) __attribute__((always_inline)) {
      const int bi = t & 255, bj = t >> 8;
      update(acc, bi, bj);
      const int c0 = 16 * (bj - k2) + col;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int r = 16 * (bi - k2) + kk + 4 * g;
        Dn[r][c0] = c0 <= r ? acc[g] : 0.0;
      }
    });
    CHOL_STAMP_W(5);
    lds_barrier();
    CHOL_STAMP_W(6);
    // (3)
    for (int e = wt; e < m * kCb; e += kWorkThreads) A[(int64_t)(kb + kCb + (e >> 5)) * lda + kb + (e & 31)] = P[e >> 5][e & 31];
    for_slots(__ballot(in_col && slot_bi >= k2 + 2), [&](chol_f64x4& acc, int t) __attribute__((always_inline)) {
      const int bi = t & 255, bj = t >> 8;
      update(acc, bi, bj);
      const int c0 = 16 * (bj - k2) + col;
#pragma unroll
      for (int g = 0; g < 4; ++g) Pn[16 * (bi - k2 - 2) + kk + 4 * g][c0] = acc[g];
    });
    for_slots(__ballot(slot_tile >= 0 && slot_bj >= k2 + 2), [&](chol_f64x4& acc, int t) __attribute__((always_inline)) { update(acc, t & 255, t >> 8); });
    lds_barrier();
    block_out_and_panel(Dn, Pn, kb + kCb, m - kCb);      // (1) of the next step
    lds_barrier();
  }
}

// L y = b, then L^T x = y, in place in B - blocked, on the float64 matrix cores.  One workgroup of 1024 threads per slab
// of 16 right-hand sides; the slab lives in LDS (Y, n x 16) for the whole solve.  Per 32-row block k, forward:
//     X   = Linv_kk  Y_k                      two 16 x 16 tiles (waves 0, 1), 8 MFMA steps each
//     Y_r -= L_rk X   for the rows r below    (n - kb - 32) / 16 tiles, one per wave, L read from global (L2)
// and backward the same with Linv_kk^T and L^T (whose tiles are read along the coalesced direction of the row-major
// factor).  Round 2 ran one wave per right-hand side with matrix-VECTOR products: every wave re-read all of L and the
// block inverses with one 8-byte load per multiply-add - 92 us for 32 right-hand sides (r03 rocprofv3), most of it load
// latency.  The first r03 version (slabs of 32, two tiles per wave) took 67 us: 18 block steps of 3.7 us, each of them
// one exposed L2 round trip for the wave's tiles of L (issued at the top of the step, needed after the block solve) on top
// of two chains of 8 dependent MFMAs.  With slabs of 16 a wave has ONE tile per step, so that the tile of the NEXT step
// fits in registers beside it (2 x 8 doubles) and is fetched a whole step ahead; a 32-target fit also runs on two CUs.
// MFMA operand layout (as in chol_factor_kernel): lane (col = lane & 15, kk = lane >> 4) supplies A[16 I + col][4 s + kk]
// and B^T[col][4 s + kk]; accumulator register g is element (row kk + 4 g, column col) of the tile.
constexpr int kSw = 16;        // right-hand sides per slab
constexpr int kYs = kSw + 1;   // LDS row stride of the slab
__global__ __launch_bounds__(1024) void chol_solve_kernel(const double* __restrict__ L, int64_t lda, int n,
                                                          const double* __restrict__ dinv, double* __restrict__ B,
                                                          int64_t ldb, int T, int dinv_in_lds CHOL_STAMP_PARAM) {
  extern __shared__ __attribute__((aligned(16))) double solve_lds[];
#ifdef HSR_CHOL_STAMPS
#define SOLVE_STAMP(k) do { if (threadIdx.x == 0 && blockIdx.x == 0 && stamps) { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory"); stamps[200 + (k)] = t_; } } while (0)
#else
#define SOLVE_STAMP(k)
#endif
  SOLVE_STAMP(0);
  double (*Y)[kYs] = reinterpret_cast<double (*)[kYs]>(solve_lds);
  double (*X)[kYs] = reinterpret_cast<double (*)[kYs]>(solve_lds + (size_t)n * kYs);
  double* dl = solve_lds + (size_t)(n + kCb) * kYs;        // all block inverses, when they fit
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int col = lane & 15, kk = lane >> 4;
  const int c0 = blockIdx.x * kSw;
  for (int e = tid; e < n * kSw; e += 1024) {
    const int r = e >> 4, c = e & 15;
    Y[r][c] = c0 + c < T ? B[(int64_t)r * ldb + c0 + c] : 0.0;
  }
  // rows of the block inverses padded to 33 doubles in LDS: with the dense stride of 32 (= 64 banks) the 16 rows a wave reads
  // for one MFMA operand start in the same bank
  if (dinv_in_lds)
    for (int e = tid; e < n * kCb; e += 1024) dl[(e >> 5) * kCs + (e & 31)] = dinv[e];
  const int ds = dinv_in_lds ? kCs : kCb;                                // row stride of a block inverse
  const int nsteps = n / kCb;
  // step q = 0 .. 2 nsteps - 1: forward with L over the blocks top down, then backward with L^T bottom up
  struct Step {
    int kb, r0, ntile;
    int64_t sr, sc;      // element (r, c) of L (forward) or L^T (backward)
  };
  auto step_of = [&](int q) {
    Step s;
    const bool fwd = q < nsteps;
    s.kb = fwd ? q * kCb : n - kCb - (q - nsteps) * kCb;
    s.r0 = fwd ? s.kb + kCb : 0;
    s.ntile = (fwd ? n - s.kb - kCb : s.kb) / 16;
    s.sr = fwd ? lda : 1;
    s.sc = fwd ? 1 : lda;
    return s;
  };
  auto load_tile = [&](double (&av)[kCb / 4], const Step& s, int t) {   // this wave's 16 x 32 tile of L for the update of step s
    const int row = s.r0 + 16 * t + col;
#pragma unroll
    for (int st = 0; st < kCb / 4; ++st) av[st] = L[(int64_t)row * s.sr + (int64_t)(s.kb + 4 * st + kk) * s.sc];
  };
  double av[kCb / 4], an[kCb / 4];
#pragma unroll
  for (int st = 0; st < kCb / 4; ++st) av[st] = an[st] = 0.0;
  {
    const Step s0 = step_of(0);
    if (wave < s0.ntile) load_tile(av, s0, wave);
  }
  __syncthreads();
  SOLVE_STAMP(1);
  // one step; `cur` holds this wave's tile of L for step q, `nxt` receives the one for step q + 1 (the two arrays swap roles
  // from step to step: a register copy at the end of the step would wait for the loads just issued)
  auto run_step = [&](double (&cur)[kCb / 4], double (&nxt)[kCb / 4], int q) {
    const Step s = step_of(q);
    const bool fwd = q < nsteps;
    const int dr = fwd ? ds : 1, dc = fwd ? 1 : ds;                     // element (r, c) of the block inverse / its transpose
    // (1) the wave's tile of the NEXT step: does not depend on anything computed here, a whole step of lead
    if (q + 1 < 2 * nsteps) {
      const Step sn = step_of(q + 1);
      if (wave < sn.ntile) load_tile(nxt, sn, wave);
    }
    // (2) X = Linv_kk Y_k (forward) / Linv_kk^T Y_k (backward)
    // (two copies on purpose: through ONE pointer that may be LDS or global the operand loads become flat loads, and a flat
    // load waits for vmcnt(0) - i.e. for the tile fetched ahead in (1); measured: 3.5 k cycles per block solve instead of 1 k)
    if (wave < 2) {
      chol_f64x4 acc = {0.0, 0.0, 0.0, 0.0};
      const size_t blk = (size_t)(s.kb / kCb) * kCb * ds;                // Linv_kk, row-major
      if (dinv_in_lds) {
#pragma unroll
        for (int st = 0; st < kCb / 4; ++st) {
          const int p = 4 * st + kk;
          acc = __builtin_amdgcn_mfma_f64_16x16x4f64(dl[blk + (16 * wave + col) * dr + p * dc], Y[s.kb + p][col], acc, 0, 0, 0);
        }
      } else {
#pragma unroll
        for (int st = 0; st < kCb / 4; ++st) {
          const int p = 4 * st + kk;
          acc = __builtin_amdgcn_mfma_f64_16x16x4f64(dinv[blk + (16 * wave + col) * dr + p * dc], Y[s.kb + p][col], acc, 0, 0, 0);
        }
      }
#pragma unroll
      for (int g = 0; g < 4; ++g) X[16 * wave + kk + 4 * g][col] = acc[g];
    }
    lds_barrier();
    SOLVE_STAMP(2 + 2 * q);
    if (tid < kCb * kSw) Y[s.kb + (tid >> 4)][tid & 15] = X[tid >> 4][tid & 15];
    // (3) the other rows: below the block (forward) / above it (backward)
    if (wave < s.ntile) {
      chol_f64x4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int st = 0; st < kCb / 4; ++st) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(cur[st], X[4 * st + kk][col], acc, 0, 0, 0);
#pragma unroll
      for (int g = 0; g < 4; ++g) Y[s.r0 + 16 * wave + kk + 4 * g][col] -= acc[g];
    }
    for (int t = wave + 16; t < s.ntile; t += 16) {                      // n > 288: the tiles beyond one per wave
      const int row = s.r0 + 16 * t + col;
      chol_f64x4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int st = 0; st < kCb / 4; ++st) {
        const int p = 4 * st + kk;
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(L[(int64_t)row * s.sr + (int64_t)(s.kb + p) * s.sc], X[p][col], acc, 0, 0, 0);
      }
#pragma unroll
      for (int g = 0; g < 4; ++g) Y[s.r0 + 16 * t + kk + 4 * g][col] -= acc[g];
    }
    lds_barrier();
    SOLVE_STAMP(3 + 2 * q);
  };
#pragma unroll 1
  for (int q = 0; q < 2 * nsteps; q += 2) {
    run_step(av, an, q);
    run_step(an, av, q + 1);
  }
  for (int e = tid; e < n * kSw; e += 1024) {
    const int r = e >> 4, c = e & 15;
    if (c0 + c < T) B[(int64_t)r * ldb + c0 + c] = Y[r][c];
  }
  SOLVE_STAMP(2 + 4 * nsteps);
}

}  // namespace hsr

extern "C" size_t hsr_chol_work_bytes(int32_t n) { return n >= 32 ? (size_t)n * hsr::kCb * sizeof(double) : 0; }

extern "C" int hsr_chol_solve_f64(double* a_dev, int64_t lda, int32_t n, double* b_dev, int64_t ldb, int32_t nrhs,
                                  double* work_dev, int32_t* info_dev, hsr_stream_t stream) {
  using namespace hsr;
  HSR_REQUIRE(a_dev && b_dev && work_dev && info_dev, HSR_ERR_INVALID, "hsr_chol_solve_f64: NULL pointer");
  HSR_REQUIRE(n >= kCb && n <= 512 && n % kCb == 0, HSR_ERR_UNSUPPORTED,
              "hsr_chol_solve_f64: n=%d must be a multiple of 32 in [32, 512] (pad with an identity block)", n);
  HSR_REQUIRE(lda >= n && nrhs >= 1 && ldb >= nrhs, HSR_ERR_INVALID, "hsr_chol_solve_f64: bad leading dimension");
  hipStream_t s = (hipStream_t)stream;
  const size_t lds_f = ((size_t)kCb + (size_t)(n - kCb)) * kCs * sizeof(double);
  static thread_local size_t configured = 0;
  if (lds_f > configured) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(chol_factor_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_f);
    (void)hipGetLastError();
    configured = lds_f;
  }
  static const bool no_res = getenv("HSR_CHOL_NO_RES") != nullptr;     // A/B switch of tools/chol_stamps
  if (n <= kResMaxN && !no_res) {
    const size_t lds_r = (size_t)(2 * n - kCb) * kDs * sizeof(double);   // two (D | P) buffers of n and n - 32 rows of 34 doubles
    static thread_local size_t configured_r = 0;
    if (lds_r > configured_r) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(chol_factor_res_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_r);
      (void)hipGetLastError();
      configured_r = lds_r;
    }
    hipLaunchKernelGGL(chol_factor_res_kernel, dim3(1), dim3(kResThreads), lds_r, s, a_dev, lda, n, info_dev, work_dev CHOL_STAMP_ARG);
  } else {
    hipLaunchKernelGGL(chol_factor_kernel, dim3(1), dim3(1024), lds_f, s, a_dev, lda, n, info_dev, work_dev CHOL_STAMP_ARG);
  }
  size_t lds_s = ((size_t)n + kCb) * kYs * sizeof(double);
  const int dinv_in_lds = lds_s + (size_t)n * kCs * sizeof(double) <= 160 * 1024 ? 1 : 0;
  if (dinv_in_lds) lds_s += (size_t)n * kCs * sizeof(double);
  static thread_local size_t configured_s = 0;
  if (lds_s > configured_s) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(chol_solve_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_s);
    (void)hipGetLastError();
    configured_s = lds_s;
  }
  hipLaunchKernelGGL(chol_solve_kernel, dim3((nrhs + kSw - 1) / kSw), dim3(1024), lds_s, s, a_dev, lda, n, work_dev, b_dev, ldb, nrhs, dinv_in_lds CHOL_STAMP_ARG);
  HSR_LAUNCH_CHECK("chol kernels");
  return HSR_OK;
}
