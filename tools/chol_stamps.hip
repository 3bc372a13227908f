// Where the one-workgroup Cholesky factorisation (csrc/hsr_chol.hip) spends its cycles: s_memtime at the phase boundaries
// of every 32-column block step.  Diagnostic build:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DHSR_CHOL_STAMPS tools/chol_stamps.hip \
//         hyperspectral_super-resolution_amd/csrc/hsr_chol.hip hyperspectral_super-resolution_amd/csrc/hsr_lib.hip -o tools/chol_stamps
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <vector>
#include "../include/hsr.h"
#ifdef HSR_CHOL_STAMPS
namespace hsr { extern unsigned long long* g_chol_stamps; }
#endif
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
int main(int argc, char** argv) {
  setvbuf(stdout, NULL, _IONBF, 0);
  const int n = argc > 1 ? atoi(argv[1]) : 288, T = argc > 2 ? atoi(argv[2]) : 32;
  std::vector<double> A((size_t)n * n), B((size_t)n * T, 1.0);
  uint32_t s = 7;
  std::vector<double> M((size_t)n * (n + 40));
  for (auto& x : M) { s = s * 1664525u + 1013904223u; x = ((s >> 8) / 16777216.0) - 0.5; }
  for (int i = 0; i < n; ++i) for (int j = 0; j <= i; ++j) { double v = 0; for (int k = 0; k < n + 40; ++k) v += M[(size_t)i * (n + 40) + k] * M[(size_t)j * (n + 40) + k]; v = v / n + (i == j ? 0.5 : 0.0); A[(size_t)i * n + j] = A[(size_t)j * n + i] = v; }
  double *dA, *dB, *dW; int* dI; unsigned long long* dS;
  CK(hipMalloc(&dA, A.size() * 8)); CK(hipMalloc(&dB, B.size() * 8)); CK(hipMalloc(&dW, hsr_chol_work_bytes(n))); CK(hipMalloc(&dI, 4)); CK(hipMalloc(&dS, 512 * 8));
#ifdef HSR_CHOL_STAMPS
  hsr::g_chol_stamps = dS;
#endif
  const char* nm[6] = {"load D, P -> LDS", "31 column steps", "sqrt, scale, inverse", "panel = P Linv^T (MFMA)", "panel -> global", "trailing update"};
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipMemcpy(dA, A.data(), A.size() * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), B.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemset(dS, 0, 512 * 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, 0));
    if (hsr_chol_solve_f64(dA, n, n, dB, T, T, dW, dI, 0)) { printf("error %s\n", hsr_last_error()); return 1; }
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long h[512]; CK(hipMemcpy(h, dS, sizeof h, hipMemcpyDeviceToHost));   // [0, 200) factor blocks, [200, 300) solve
    if (rep == 0) {   // the factor and the solution against a host Cholesky in long double
      std::vector<double> Lg((size_t)n * n), Xg((size_t)n * T);
      CK(hipMemcpy(Lg.data(), dA, Lg.size() * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(Xg.data(), dB, Xg.size() * 8, hipMemcpyDeviceToHost));
      std::vector<long double> Lh((size_t)n * n, 0.0L);
      for (int j = 0; j < n; ++j) {
        long double d = A[(size_t)j * n + j];
        for (int k = 0; k < j; ++k) d -= Lh[(size_t)j * n + k] * Lh[(size_t)j * n + k];
        d = sqrtl(d); Lh[(size_t)j * n + j] = d;
        for (int i = j + 1; i < n; ++i) { long double v = A[(size_t)i * n + j]; for (int k = 0; k < j; ++k) v -= Lh[(size_t)i * n + k] * Lh[(size_t)j * n + k]; Lh[(size_t)i * n + j] = v / d; }
      }
      double eL = 0, eX = 0;
      for (int i = 0; i < n; ++i) for (int j = 0; j <= i; ++j) { double e = fabs(Lg[(size_t)i * n + j] - (double)Lh[(size_t)i * n + j]); if (!(e <= eL)) eL = e; }
      for (int c = 0; c < T; ++c) {
        std::vector<long double> y(n);
        for (int i = 0; i < n; ++i) { long double v = 1.0L; for (int k = 0; k < i; ++k) v -= Lh[(size_t)i * n + k] * y[k]; y[i] = v / Lh[(size_t)i * n + i]; }
        for (int i = n - 1; i >= 0; --i) { long double v = y[i]; for (int k = i + 1; k < n; ++k) v -= Lh[(size_t)k * n + i] * y[k]; y[i] = v / Lh[(size_t)i * n + i]; }
        for (int i = 0; i < n; ++i) { double e = fabs(Xg[(size_t)i * T + c] - (double)y[i]) / (fabs((double)y[i]) + 1e-300); if (!(e <= eX)) eX = e; }
      }
      int info_h = -1; CK(hipMemcpy(&info_h, dI, 4, hipMemcpyDeviceToHost));
      printf("check: max |L - L_host| = %.3e, max rel |x - x_host| = %.3e, info = %d\n", eL, eX, info_h);
    }
    if (h[191] != 0) {     // chol_factor_res_kernel (n <= 288): stamps 0 (start), then 1, 2, 3 per block step
      const int nbk = n / 32;
      printf("rep %d: factor + solve %.1f us (events); resident kernel, cycles per block step:\n", rep, ms * 1e3);
      if (rep == 2) {
        printf("  first block (loads, block 0 on wave 0 | panel 0 -> LDS): %llu  (table + first loads %llu, D and P in LDS %llu, block 0 %llu, workers arrive %llu)\n", h[1] - h[0], h[189] - h[0], h[190] - h[189], h[191] - h[190], h[1] - h[191]);
        const char* rn[3] = {"L_kk, inverse -> memory; panel = P Linv^T", "panel -> memory; next block column updated and out of the registers", "next diagonal block on wave 0 | rest of the trailing update"};
        for (int k = 0; k < 3; ++k) {
          unsigned long long sum = 0;
          printf("  %-90s:", rn[k]);
          for (int b = 0; b < nbk; ++b) {
            const unsigned long long a0 = h[b * 8 + 1 + k], a1 = k < 2 ? h[b * 8 + 2 + k] : (b + 1 < nbk ? h[(b + 1) * 8 + 1] : 0);
            const unsigned long long d = (a0 && a1 > a0) ? a1 - a0 : 0;
            printf(" %llu", d); sum += d;
          }
          printf("  = %llu\n", sum);
        }
        printf("  wave 1 (worker 0) in phase 2, per block step: from wave 0's stamp after the panel barrier to its own | its slot loop | its wait at the barrier\n   ");
        for (int b = 0; b + 1 < nbk; ++b) printf(" %lld|%llu|%llu", (long long)(h[b * 8 + 4] - h[b * 8 + 2]), h[b * 8 + 5] - h[b * 8 + 4], h[b * 8 + 6] - h[b * 8 + 5]);
        printf("\n");
        printf("  whole kernel up to the last panel: %llu cycles\n", h[(nbk - 1) * 8 + 2] - h[0]);
      }
      continue;
    }
    double ph[6] = {0}; double tot = 0;
    for (int b = 0; b < n / 32; ++b) for (int k = 0; k < 6; ++k) { ph[k] += (double)(h[b * 8 + k + 1] - h[b * 8 + k]); }
    for (int k = 0; k < 6; ++k) tot += ph[k];
    printf("rep %d: factor + solve %.1f us (events); factor kernel %.0f cycles over %d block steps\n", rep, ms * 1e3, tot, n / 32);
#ifndef HSR_CHOL_STAMPS
    if (rep == 2) {   // plain build: factor and solve timed apart is not possible from here; five more back-to-back runs
      float best = 1e9f;
      for (int it = 0; it < 5; ++it) {
        CK(hipMemcpy(dA, A.data(), A.size() * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), B.size() * 8, hipMemcpyHostToDevice));
        CK(hipEventRecord(e0, 0)); hsr_chol_solve_f64(dA, n, n, dB, T, T, dW, dI, 0); CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float m2; CK(hipEventElapsedTime(&m2, e0, e1)); if (m2 < best) best = m2;
      }
      printf("plain build: best factor + solve %.1f us\n", best * 1e3);
      continue;
    }
#endif
    if (rep == 2) {
      for (int k = 0; k < 6; ++k) printf("  %-28s %9.0f cycles  %5.1f %%\n", nm[k], ph[k], 100 * ph[k] / tot);
      for (int k = 0; k < 6; ++k) { printf("    phase %d per block:", k); for (int b = 0; b < n / 32; ++b) printf(" %llu", h[b * 8 + k + 1] - h[b * 8 + k]); printf("\n"); }
      printf("  per block step (cycles):");
      for (int b = 0; b < n / 32; ++b) printf(" %llu", h[b * 8 + 6] - h[b * 8]);
      printf("\n");
      const int ns = n / 32;
      printf("  solve kernel (slab 0): load %llu cycles, %d steps, store %llu; per step (block solve + barrier | copy, update + barrier):\n   ",
             h[201] - h[200], 2 * ns, h[202 + 4 * ns] - h[201 + 4 * ns]);
      for (int q = 0; q < 2 * ns; ++q) printf(" %llu|%llu", h[202 + 2 * q] - h[201 + 2 * q], h[203 + 2 * q] - h[202 + 2 * q]);
      printf("\n");
    }
  }
  return 0;
}
