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
//   chol_diag_inverse_kernel  inverses of the diagonal blocks, so that
//   chol_solve_kernel   (forward and backward substitution, one wave per right-hand side) needs a 32 x 32
//                       matrix-vector product per block instead of a serial 32-step chain; the updates below /
//                       above are matrix-vector products read along rows of L (forward) / along columns, which is
//                       the coalesced direction of the row-major factor (backward).
// n must be a multiple of 32 (the caller pads with an identity block), n <= 512.
#include <math.h>

#include "hsr_common.h"

namespace hsr {

constexpr int kCb = 32;        // block size
constexpr int kCs = kCb + 1;   // LDS row stride (doubles)

typedef double chol_f64x4 __attribute__((ext_vector_type(4)));

// 1 / d to the last bit or two: v_rcp_f64 and two Newton steps (a full IEEE division is ~3x the instructions, and the
// factorisation's column steps are latency chains)
__device__ __forceinline__ double rcp_nr(double d) {
  double r = __builtin_amdgcn_rcp(d);
  r = fma(fma(-d, r, 1.0), r, r);
  r = fma(fma(-d, r, 1.0), r, r);
  return r;
}

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
__global__ __launch_bounds__(1024) void chol_factor_kernel(double* __restrict__ A, int64_t lda, int n, int* info) {
  extern __shared__ __attribute__((aligned(16))) double chol_lds[];
  double (*D)[kCs] = reinterpret_cast<double (*)[kCs]>(chol_lds);                      // diagonal block
  double (*P)[kCs] = reinterpret_cast<double (*)[kCs]>(chol_lds + kCb * kCs);           // panel below, (n - 32) rows
  __shared__ int bad_pivot;
  __shared__ double invd[kCb];
  const int tid = threadIdx.x, ti = tid >> 5, tj = tid & 31;
  const int lane = tid & 63, wave = tid >> 6;
  if (tid == 0) {
    *info = 0;
    bad_pivot = 0x7fffffff;
  }
  for (int kb = 0; kb < n; kb += kCb) {
    const int m = n - kb - kCb;   // rows below the diagonal block
    D[ti][tj] = tj <= ti ? A[(int64_t)(kb + ti) * lda + kb + tj] : 0.0;
    for (int e = tid; e < m * kCb; e += 1024) P[e >> 5][e & 31] = A[(int64_t)(kb + kCb + (e >> 5)) * lda + kb + (e & 31)];
    __syncthreads();
    // 32 x 32 block, one barrier per column: the Schur update of step j is applied with the UNSCALED column j
    // (D[i][l] -= D[i][j] D[l][j] / D[j][j]); D[j][j] is then the squared pivot and the columns are scaled once at
    // the end.  (Scaling each column first needs three barriers per step: 96 instead of 33 per block.)
    for (int j = 0; j < kCb - 1; ++j) {
      if (ti > j && tj > j && tj <= ti) D[ti][tj] -= D[ti][j] * D[tj][j] * rcp_nr(D[j][j]);
      __syncthreads();
    }
    double piv = 1.0;
    if (tj <= ti) {
      const double d = D[tj][tj];
      if (ti == tj && !(d > 0.0)) atomicMin(&bad_pivot, kb + tj + 1);   // LAPACK: index of the first non-positive pivot
      piv = sqrt(d);
    }
    __syncthreads();
    if (tid == 0 && *info == 0 && bad_pivot != 0x7fffffff) *info = bad_pivot;
    if (tj <= ti) {
      const double l = ti == tj ? piv : D[ti][tj] / piv;
      D[ti][tj] = l;
      A[(int64_t)(kb + ti) * lda + kb + tj] = l;
      if (ti == tj) invd[ti] = 1.0 / piv;
    }
    __syncthreads();
    // panel below: row r solves x L^T = a.  Four lanes per row; lane q owns x[q], x[q+4], ..  D's upper triangle is zero
    // and x starts at zero, so every partial dot product may run over all of a lane's columns below c.
    for (int r = tid >> 2; r < m; r += 256) {
      const int q = tid & 3;
      double x[kCb / 4];
#pragma unroll
      for (int k = 0; k < kCb / 4; ++k) x[k] = 0.0;
#pragma unroll
      for (int c = 0; c < kCb; ++c) {
        double part = 0.0;
#pragma unroll
        for (int k = 0; k < kCb / 4; ++k)
          if (4 * k < c) part = fma(x[k], D[c][q + 4 * k], part);
        part = quad_sum(part);
        const double xc = (P[r][c] - part) * invd[c];
        if (q == (c & 3)) {
          x[c >> 2] = xc;
          P[r][c] = xc;
        }
      }
    }
    __syncthreads();
    for (int e = tid; e < m * kCb; e += 1024) A[(int64_t)(kb + kCb + (e >> 5)) * lda + kb + (e & 31)] = P[e >> 5][e & 31];
    // trailing update, lower triangle in 16 x 16 tiles on the float64 matrix cores: C[I][J] -= P_I P_J^T.
    // Lane (col = lane & 15, kk = lane >> 4): A operand P[16 I + col][4 s + kk], B operand P[16 J + col][4 s + kk],
    // accumulator register g = element (row kk + 4 g, column col) of the tile (layout as in csrc/hsr_ridge.hip).
    {
      const int mt = m >> 4;                       // m is a multiple of 32
      const int ntile = mt * (mt + 1) / 2;
      const int col = lane & 15, kk = lane >> 4;
      constexpr int kMaxTiles = 30;                // n <= 512: 30 tile rows -> 465 tiles over 16 waves
      for (int t0 = wave; t0 < ntile; t0 += 16 * 5) {
        double creg[5][4];
        int bis[5], bjs[5];
#pragma unroll
        for (int u = 0; u < 5; ++u) {
          const int t = t0 + 16 * u;
          int bi = (int)((sqrt(8.0 * (double)(t < ntile ? t : 0) + 1.0) - 1.0) * 0.5);
          while ((bi + 1) * (bi + 2) / 2 <= t && bi < kMaxTiles) ++bi;
          while (bi * (bi + 1) / 2 > t && bi > 0) --bi;
          bis[u] = bi;
          bjs[u] = (t < ntile ? t : 0) - bi * (bi + 1) / 2;
          if (t < ntile) {
            const double* src = A + (int64_t)(kb + kCb + 16 * bi + kk) * lda + kb + kCb + 16 * bjs[u] + col;
#pragma unroll
            for (int g = 0; g < 4; ++g) creg[u][g] = src[(int64_t)(4 * g) * lda];
          }
        }
#pragma unroll
        for (int u = 0; u < 5; ++u) {
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
  }
}

// Inverses of the 32 x 32 diagonal blocks of L (one workgroup of 128 threads per block): with them the substitution below
// needs no serial 32-step chain per block, only a 32 x 32 matrix-vector product.  X L^T = I gives X = L^-T, solved row by
// row exactly like a panel row of the factorisation (four lanes per row, reciprocals of the diagonal), and stored
// transposed.  (First version: one thread per column, a division per step: 26 us.)
__global__ __launch_bounds__(128) void chol_diag_inverse_kernel(const double* __restrict__ L, int64_t lda,
                                                                double* __restrict__ dinv /* [n/32][32][32] */) {
  __shared__ double D[kCb][kCs], X[kCb][kCs], invd[kCb];
  const int kb = blockIdx.x * kCb, tid = threadIdx.x;
  for (int e = tid; e < kCb * kCb; e += 128) {
    const int r = e >> 5, c = e & 31;
    D[r][c] = c <= r ? L[(int64_t)(kb + r) * lda + kb + c] : 0.0;
    if (r == c) invd[r] = 1.0 / D[r][c];
  }
  __syncthreads();
  {
    const int r = tid >> 2, q = tid & 3;             // row r of X = row r of L^-T
    double x[kCb / 4];
#pragma unroll
    for (int k = 0; k < kCb / 4; ++k) x[k] = 0.0;
#pragma unroll
    for (int c = 0; c < kCb; ++c) {
      double part = 0.0;
#pragma unroll
      for (int k = 0; k < kCb / 4; ++k)
        if (4 * k < c) part = fma(x[k], D[c][q + 4 * k], part);
      part = quad_sum(part);
      const double xc = ((r == c ? 1.0 : 0.0) - part) * invd[c];
      if (q == (c & 3)) {
        x[c >> 2] = xc;
        X[r][c] = xc;
      }
    }
  }
  __syncthreads();
  for (int e = tid; e < kCb * kCb; e += 128) {
    const int r = e >> 5, c = e & 31;
    dinv[((size_t)blockIdx.x * kCb + r) * kCb + c] = X[c][r];    // L^-1 = (L^-T)^T
  }
}

// L y = b, then L^T x = y, in place in column `col` of B.  One wave per column; 4 waves per workgroup.
__global__ __launch_bounds__(256) void chol_solve_kernel(const double* __restrict__ L, int64_t lda, int n,
                                                         const double* __restrict__ dinv, double* __restrict__ B,
                                                         int64_t ldb, int T) {
  extern __shared__ __attribute__((aligned(16))) double solve_lds[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int col = blockIdx.x * 4 + wave;
  double* y = solve_lds + (size_t)wave * (n + kCb);
  double* xb = y + n;            // the block solution being formed (32 values)
  if (col >= T) return;          // no barriers below: every wave works on its own slice of LDS
  for (int r = lane; r < n; r += 64) y[r] = B[(int64_t)r * ldb + col];
  const int c = lane & 31;
  // ---- forward: blocks top to bottom
  for (int kb = 0; kb < n; kb += kCb) {
    const double* di = dinv + ((size_t)(kb / kCb) * kCb + c) * kCb;   // row c of the block's inverse
    double s = 0.0;
#pragma unroll
    for (int p = 0; p < kCb; ++p) s += di[p] * y[kb + p];             // x_c = sum_{p <= c} Linv[c][p] y_p (rest is 0)
    if (lane < kCb) xb[c] = s;
    if (lane < kCb) y[kb + c] = xb[c];
    // rows below: y[r] -= L[r][kb .. kb+31] . x
    for (int r = kb + kCb + lane; r < n; r += 64) {
      const double* lr = L + (int64_t)r * lda + kb;
      double t = 0.0;
#pragma unroll
      for (int p = 0; p < kCb; ++p) t += lr[p] * xb[p];
      y[r] -= t;
    }
  }
  // ---- backward: blocks bottom to top, with L^T:  x = Linv^T y
  for (int kb = n - kCb; kb >= 0; kb -= kCb) {
    const double* di = dinv + (size_t)(kb / kCb) * kCb * kCb;
    double s = 0.0;
#pragma unroll
    for (int p = 0; p < kCb; ++p) s += di[(size_t)p * kCb + c] * y[kb + p];   // (Linv^T)[c][p] = Linv[p][c]
    if (lane < kCb) xb[c] = s;
    if (lane < kCb) y[kb + c] = xb[c];
    // rows above: y[r] -= sum_p L[kb + p][r] x_p   (coalesced along r)
    for (int r = lane; r < kb; r += 64) {
      double t = 0.0;
#pragma unroll
      for (int p = 0; p < kCb; ++p) t += L[(int64_t)(kb + p) * lda + r] * xb[p];
      y[r] -= t;
    }
  }
  for (int r = lane; r < n; r += 64) B[(int64_t)r * ldb + col] = y[r];
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
  hipLaunchKernelGGL(chol_factor_kernel, dim3(1), dim3(1024), lds_f, s, a_dev, lda, n, info_dev);
  hipLaunchKernelGGL(chol_diag_inverse_kernel, dim3(n / kCb), dim3(128), 0, s, a_dev, lda, work_dev);
  hipLaunchKernelGGL(chol_solve_kernel, dim3((nrhs + 3) / 4), dim3(256), (size_t)4 * (n + kCb) * sizeof(double), s, a_dev, lda, n,
                     work_dev, b_dev, ldb, nrhs);
  HSR_LAUNCH_CHECK("chol kernels");
  return HSR_OK;
}
