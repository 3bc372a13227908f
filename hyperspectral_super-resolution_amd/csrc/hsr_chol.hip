// Cholesky solve of the ridge system of variant a9: (Phi_c^T Phi_c + alpha I) W = Phi_c^T Y_c, 285 x 285 float64
// with up to 285 right-hand sides (legacy_notebooks/Spectral_matching.ipynb, Ridge(alpha=1) at raw lines 475-490).
//
// The matrix is tiny for a GPU (650 KB) and the factorisation is a chain of 9 dependent panel steps, so a
// library that launches a kernel per sub-step pays mostly launch and dependency latency: rocSOLVER through
// torch.linalg took 0.93 ms of a 1.35 ms fit (potf2_kernel_small 3 x 122 us, four substitution kernels 500 us).
// Here:
//   chol_factor_kernel  ONE workgroup of 1024 threads runs the whole right-looking blocked factorisation
//                       (block 32): diagonal block factored in LDS with one barrier per column, the panel below
//                       solved one row per thread (row in registers, L broadcast from LDS), the trailing matrix
//                       updated from the LDS-resident panel with 4 x 4 register tiles.
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

__global__ __launch_bounds__(1024) void chol_factor_kernel(double* __restrict__ A, int64_t lda, int n, int* info) {
  extern __shared__ __attribute__((aligned(16))) double chol_lds[];
  double (*D)[kCs] = reinterpret_cast<double (*)[kCs]>(chol_lds);                      // diagonal block
  double (*P)[kCs] = reinterpret_cast<double (*)[kCs]>(chol_lds + kCb * kCs);           // panel below, (n - 32) rows
  __shared__ int bad_pivot;
  const int tid = threadIdx.x, ti = tid >> 5, tj = tid & 31;
  if (tid == 0) {
    *info = 0;
    bad_pivot = 0x7fffffff;
  }
  for (int kb = 0; kb < n; kb += kCb) {
    const int m = n - kb - kCb;   // rows below the diagonal block
    D[ti][tj] = tj <= ti ? A[(int64_t)(kb + ti) * lda + kb + tj] : 0.0;
    __syncthreads();
    // 32 x 32 block, one barrier per column: the Schur update of step j is applied with the UNSCALED column j
    // (D[i][l] -= D[i][j] D[l][j] / D[j][j]); D[j][j] is then the squared pivot and the columns are scaled once at
    // the end.  (Scaling each column first needs three barriers per step: 96 instead of 33 per block.)
    for (int j = 0; j < kCb - 1; ++j) {
      if (ti > j && tj > j && tj <= ti) D[ti][tj] -= D[ti][j] * D[tj][j] / D[j][j];
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
    }
    __syncthreads();
    // panel below: row r of the panel solves x L^T = a, one thread per row, the row in registers, L broadcast from LDS
    if (tid < m) {
      const int r = tid;
      double* arow = A + (int64_t)(kb + kCb + r) * lda + kb;
      double x[kCb];
#pragma unroll
      for (int c = 0; c < kCb; ++c) x[c] = arow[c];
#pragma unroll
      for (int c = 0; c < kCb; ++c) {
        double s = x[c];
#pragma unroll
        for (int p = 0; p < kCb; ++p)
          if (p < c) s -= x[p] * D[c][p];
        x[c] = s / D[c][c];
      }
#pragma unroll
      for (int c = 0; c < kCb; ++c) {
        arow[c] = x[c];
        P[r][c] = x[c];
      }
    }
    __syncthreads();
    // trailing update, lower triangle in 4 x 4 tiles: A[i][j] -= sum_p P[i][p] P[j][p]
    {
      const int m4 = m >> 2;
      const int ntile = m4 * (m4 + 1) / 2;
      for (int t = tid; t < ntile; t += 1024) {
        int bi = (int)((sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
        while ((bi + 1) * (bi + 2) / 2 <= t) ++bi;
        while (bi * (bi + 1) / 2 > t) --bi;
        const int bj = t - bi * (bi + 1) / 2;
        double acc[4][4];
#pragma unroll
        for (int x = 0; x < 4; ++x)
#pragma unroll
          for (int y = 0; y < 4; ++y) acc[x][y] = 0.0;
        for (int p = 0; p < kCb; ++p) {
          double a[4], b[4];
#pragma unroll
          for (int x = 0; x < 4; ++x) {
            a[x] = P[4 * bi + x][p];
            b[x] = P[4 * bj + x][p];
          }
#pragma unroll
          for (int x = 0; x < 4; ++x)
#pragma unroll
            for (int y = 0; y < 4; ++y) acc[x][y] += a[x] * b[y];
        }
        double* dst = A + (int64_t)(kb + kCb + 4 * bi) * lda + kb + kCb + 4 * bj;
#pragma unroll
        for (int x = 0; x < 4; ++x)
#pragma unroll
          for (int y = 0; y < 4; ++y) dst[(int64_t)x * lda + y] -= acc[x][y];
      }
    }
    __syncthreads();
  }
}

// Inverses of the 32 x 32 diagonal blocks of L (one workgroup of 32 threads per block, thread c solves L x = e_c):
// with them the substitution below needs no serial 32-step chain per block, only a 32 x 32 matrix-vector product.
__global__ __launch_bounds__(64) void chol_diag_inverse_kernel(const double* __restrict__ L, int64_t lda,
                                                               double* __restrict__ dinv /* [n/32][32][32] */) {
  __shared__ double D[kCb][kCs], Di[kCb][kCs];
  const int kb = blockIdx.x * kCb, c = threadIdx.x;
  if (c < kCb)
    for (int r = 0; r < kCb; ++r) {
      D[r][c] = c <= r ? L[(int64_t)(kb + r) * lda + kb + c] : 0.0;
      Di[r][c] = 0.0;
    }
  __syncthreads();
  if (c < kCb)
    for (int r = c; r < kCb; ++r) {
      double s = r == c ? 1.0 : 0.0;
      for (int p = c; p < r; ++p) s -= D[r][p] * Di[p][c];
      Di[r][c] = s / D[r][r];
    }
  __syncthreads();
  if (c < kCb)
    for (int r = 0; r < kCb; ++r) dinv[((size_t)blockIdx.x * kCb + r) * kCb + c] = Di[r][c];
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
  hipLaunchKernelGGL(chol_diag_inverse_kernel, dim3(n / kCb), dim3(64), 0, s, a_dev, lda, work_dev);
  hipLaunchKernelGGL(chol_solve_kernel, dim3((nrhs + 3) / 4), dim3(256), (size_t)4 * (n + kCb) * sizeof(double), s, a_dev, lda, n,
                     work_dev, b_dev, ldb, nrhs);
  HSR_LAUNCH_CHECK("chol kernels");
  return HSR_OK;
}
