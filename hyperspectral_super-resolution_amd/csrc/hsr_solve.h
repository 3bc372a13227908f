// np.polyfit from the moment sums: the per-band solve shared by the stand-alone solve / reduce+solve kernels
// (hsr_poly.hip), the host-side hsr_poly_solve_host and the fused tail of K1 (hsr_srf.hip).  One definition, so that
// every caller produces the same bits.
#pragma once
#include "hsr_common.h"

namespace hsr {

// ------------------------------------------------------------------------------------------------
// solve: np.polyfit from the moments
// ------------------------------------------------------------------------------------------------
// np.polyfit(x, y, deg): V = vander(x, deg+1) (highest power first), s_j = ||V[:,j]||, least squares
// of (V/s) c' = y by SVD with rcond = len(x)*eps, c = c'/s.  From the moments:
//   A_jk = S_{(d-j)+(d-k)} / (s_j s_k),  s_j = sqrt(S_{2(d-j)}),  rhs_j = T_{d-j} / s_j,
// eigen-decompose A (cyclic Jacobi, <= 5x5) and apply the pseudo-inverse keeping the eigenvalues
// above rcond^2 * max (singular values of V/s are the square roots).
// Rank-revealing path: symmetric cyclic Jacobi on the scaled Gram, pseudo-inverse with NumPy's cut-off.
// Storage is flat and caller-provided (rows of kSolveLd doubles): the stand-alone kernels and the host hand in local
// arrays, the fused tail of K1 hands in LDS so that the dynamically indexed matrices do not make K1 a scratch-using
// kernel.  Same arithmetic in the same order either way.
constexpr int kSolveLd = HSR_MAX_DEG + 1;
constexpr int kSolveWork = 2 * kSolveLd * kSolveLd + 2 * kSolveLd;   // doubles of work space per band: A, V, rhs, s
__host__ __device__ inline void solve_band_jacobi(double* A, double* V, const double* rhs, const double* s, int n,
                                                  double count, double* coef) {
  for (int j = 0; j < n; ++j)
    for (int k = 0; k < n; ++k) V[j * kSolveLd + k] = j == k ? 1.0 : 0.0;
  for (int sweep = 0; sweep < 40; ++sweep) {
    double off = 0.0, diag = 0.0;
    for (int p = 0; p < n; ++p) {
      diag += A[p * kSolveLd + p] * A[p * kSolveLd + p];
      for (int q = p + 1; q < n; ++q) off += A[p * kSolveLd + q] * A[p * kSolveLd + q];
    }
    if (off <= 1e-36 * diag) break;
    for (int p = 0; p < n; ++p) {
      for (int q = p + 1; q < n; ++q) {
        const double apq = A[p * kSolveLd + q];
        if (fabs(apq) < 1e-300) continue;
        const double theta = (A[q * kSolveLd + q] - A[p * kSolveLd + p]) / (2.0 * apq);
        const double tt = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
        const double c = 1.0 / sqrt(tt * tt + 1.0), sn = tt * c;
        for (int k = 0; k < n; ++k) {
          const double akp = A[k * kSolveLd + p], akq = A[k * kSolveLd + q];
          A[k * kSolveLd + p] = c * akp - sn * akq;
          A[k * kSolveLd + q] = sn * akp + c * akq;
        }
        for (int k = 0; k < n; ++k) {
          const double apk = A[p * kSolveLd + k], aqk = A[q * kSolveLd + k];
          A[p * kSolveLd + k] = c * apk - sn * aqk;
          A[q * kSolveLd + k] = sn * apk + c * aqk;
        }
        for (int k = 0; k < n; ++k) {
          const double vkp = V[k * kSolveLd + p], vkq = V[k * kSolveLd + q];
          V[k * kSolveLd + p] = c * vkp - sn * vkq;
          V[k * kSolveLd + q] = sn * vkp + c * vkq;
        }
      }
    }
  }
  double lmax = 0.0;
  for (int i = 0; i < n; ++i) lmax = A[i * kSolveLd + i] > lmax ? A[i * kSolveLd + i] : lmax;
  const double rcond = count * 2.220446049250313e-16;
  const double thresh = rcond * rcond * lmax;
  for (int j = 0; j < n; ++j) coef[j] = 0.0;
  for (int i = 0; i < n; ++i) {
    const double lam = A[i * kSolveLd + i];
    if (!(lam > thresh)) continue;
    double proj = 0.0;
    for (int k = 0; k < n; ++k) proj += V[k * kSolveLd + i] * rhs[k];
    proj /= lam;
    for (int j = 0; j < n; ++j) coef[j] += V[j * kSolveLd + i] * proj;
  }
  for (int j = 0; j < n; ++j) coef[j] /= s[j];
}

// Fast path: Cholesky of the scaled Gram with every loop unrolled (N is a template constant, so the
// whole factorisation lives in registers).  A pivot below 1e-13 (the scaled diagonal is exactly 1, so
// this is cond(V/s) > ~3e6) hands the band to the rank-revealing Jacobi path, which reproduces
// np.polyfit's singular-value cut-off; above it both paths agree to ~cond * eps.
// EXTWORK: the Jacobi path works in ``work`` (kSolveWork doubles, e.g. LDS) instead of local arrays.
template <int DEG, bool EXTWORK = false>
__host__ __device__ inline void solve_band_t(const double* mom, long long min_count, double* coef, double* work = nullptr) {
  constexpr int n = DEG + 1;
  const double* S = mom;
  const double* T = mom + 2 * DEG + 1;
  const double count = S[0];
  if (!(count >= (double)min_count) || count < 1.0) {  // reference fallback: identity polynomial
#pragma unroll
    for (int j = 0; j < n; ++j) coef[j] = j == n - 2 ? 1.0 : 0.0;
    return;
  }
  double s[HSR_MAX_DEG + 1], A[HSR_MAX_DEG + 1][HSR_MAX_DEG + 1], rhs[HSR_MAX_DEG + 1];
#pragma unroll
  for (int j = 0; j < n; ++j) {
    const double d = S[2 * (DEG - j)];
    s[j] = d > 0.0 ? sqrt(d) : 1.0;
  }
#pragma unroll
  for (int j = 0; j < n; ++j) {
    rhs[j] = T[DEG - j] / s[j];
#pragma unroll
    for (int k = 0; k < n; ++k) A[j][k] = S[(DEG - j) + (DEG - k)] / (s[j] * s[k]);
  }
  double L[n][n];
  bool ok = true;
#pragma unroll
  for (int j = 0; j < n; ++j) {
    double d = A[j][j];
#pragma unroll
    for (int k = 0; k < j; ++k) d -= L[j][k] * L[j][k];
    ok = ok && (d > 1e-13);
    const double ljj = sqrt(d > 1e-13 ? d : 1.0);
    L[j][j] = ljj;
#pragma unroll
    for (int i = j + 1; i < n; ++i) {
      double v = A[i][j];
#pragma unroll
      for (int k = 0; k < j; ++k) v -= L[i][k] * L[j][k];
      L[i][j] = v / ljj;
    }
  }
  if (ok) {
    double z[n];
#pragma unroll
    for (int i = 0; i < n; ++i) {
      double v = rhs[i];
#pragma unroll
      for (int k = 0; k < i; ++k) v -= L[i][k] * z[k];
      z[i] = v / L[i][i];
    }
#pragma unroll
    for (int i = n - 1; i >= 0; --i) {
      double v = z[i];
#pragma unroll
      for (int k = i + 1; k < n; ++k) v -= L[k][i] * z[k];
      z[i] = v / L[i][i];
    }
#pragma unroll
    for (int j = 0; j < n; ++j) coef[j] = z[j] / s[j];
    return;
  }
  // rank-revealing path: the matrices move to flat storage with static indices (no scratch when ``work`` is LDS)
  double* wk = work;
  double local[EXTWORK ? 1 : kSolveWork];
  if constexpr (!EXTWORK) wk = local;
  double* Af = wk;
  double* Vf = wk + kSolveLd * kSolveLd;
  double* rf = Vf + kSolveLd * kSolveLd;
  double* sf = rf + kSolveLd;
#pragma unroll
  for (int j = 0; j < n; ++j) {
    rf[j] = rhs[j];
    sf[j] = s[j];
#pragma unroll
    for (int k = 0; k < n; ++k) Af[j * kSolveLd + k] = A[j][k];
  }
  solve_band_jacobi(Af, Vf, rf, sf, n, count, coef);
}

// The same solve with the Jacobi matrices in caller-provided storage (kSolveWork doubles, e.g. LDS): keeps a kernel free
// of scratch memory.
__host__ __device__ inline void solve_band_work(const double* mom, int deg, long long min_count, double* coef, double* work) {
  switch (deg) {
    case 1: solve_band_t<1, true>(mom, min_count, coef, work); break;
    case 2: solve_band_t<2, true>(mom, min_count, coef, work); break;
    case 3: solve_band_t<3, true>(mom, min_count, coef, work); break;
    default: solve_band_t<4, true>(mom, min_count, coef, work); break;
  }
}

__host__ __device__ inline void solve_band(const double* mom, int deg, long long min_count, double* coef) {
  switch (deg) {
    case 1: solve_band_t<1>(mom, min_count, coef); break;
    case 2: solve_band_t<2>(mom, min_count, coef); break;
    case 3: solve_band_t<3>(mom, min_count, coef); break;
    default: solve_band_t<4>(mom, min_count, coef); break;
  }
}

}  // namespace hsr
