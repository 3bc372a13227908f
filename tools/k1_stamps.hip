// Diagnostic build of the PRODUCTION K1 source with per-phase s_memtime stamps (-DHSR_PHASE_STAMPS).
// Prints where a tile iteration spends its cycles (shares only - never quote this build's run time).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DHSR_PHASE_STAMPS tools/k1_stamps.hip \
//         hyperspectral_super-resolution_amd/csrc/hsr_srf.hip hyperspectral_super-resolution_amd/csrc/hsr_lib.hip -o tools/k1_stamps
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "../include/hsr.h"
namespace hsr { extern unsigned long long* g_stamp_buffer; }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

int main(int argc, char** argv) {
  const int H = argc > 1 ? atoi(argv[1]) : 1024, W = argc > 2 ? atoi(argv[2]) : 1024, B = 285, nb = 12;
  const int64_t npix = (int64_t)H * W;
  const int tilepx = argc > 3 ? atoi(argv[3]) : 64;
  const int u16 = argc > 4 ? atoi(argv[4]) : 0;      // 1: uint16 cube (ring kernel), 2: with the fast-arithmetic flag
  hsr_srf_options opts = {tilepx, 0, 0, u16 == 2 ? HSR_SRF_U16_FAST : 0};
  printf("tile pixels %d\n", tilepx);
  std::vector<float> wn((size_t)nb * B, 0.f);
  int k0[16], klen[16];
  const int centres[12] = {8, 15, 24, 38, 44, 48, 54, 62, 65, 76, 166, 244};
  const int widths[12] = {6, 17, 10, 9, 5, 5, 6, 30, 6, 6, 24, 48};
  for (int b = 0; b < nb; ++b) {
    int a = centres[b] - widths[b] / 2; if (a < 0) a = 0; int e = a + widths[b]; if (e > B) e = B;
    double sum = 0;
    for (int k = a; k < e; ++k) { double d = (k - centres[b]) / (widths[b] / 4.0 + 0.5); wn[b * B + k] = (float)exp(-0.5 * d * d); sum += wn[b * B + k]; }
    for (int k = a; k < e; ++k) wn[b * B + k] /= (float)sum;
    k0[b] = a; klen[b] = e - a;
  }
  float *d_cube, *d_wn, *d_planes, *d_real; double* d_part; unsigned long long* d_st;
  CK(hipMalloc(&d_cube, npix * B * 4)); CK(hipMalloc(&d_wn, wn.size() * 4)); CK(hipMalloc(&d_planes, (size_t)nb * npix * 4));
  CK(hipMalloc(&d_real, (size_t)nb * npix * 4)); CK(hipMalloc(&d_part, hsr_partials_bytes(nb, 4)));
  const int G = 1024, NW = 8; const size_t stn = (size_t)G * NW * 8;
  CK(hipMalloc(&d_st, stn * 8));
  { std::vector<float> h((size_t)npix * B); uint32_t s = 12345; for (auto& x : h) { s = s * 1664525u + 1013904223u; x = (s >> 8) * (0.6f / 16777216.f); }
    CK(hipMemcpy(d_cube, h.data(), h.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(d_real, h.data(), (size_t)nb * npix * 4, hipMemcpyHostToDevice)); }
  CK(hipMemcpy(d_wn, wn.data(), wn.size() * 4, hipMemcpyHostToDevice));
  uint16_t* d_cube16 = nullptr;
  if (u16) {
    std::vector<uint16_t> h((size_t)npix * B); uint32_t s = 777; for (auto& x : h) { s = s * 1664525u + 1013904223u; x = (uint16_t)((s >> 16) % 6000u); }
    CK(hipMalloc(&d_cube16, h.size() * 2)); CK(hipMemcpy(d_cube16, h.data(), h.size() * 2, hipMemcpyHostToDevice));
  }
  hsr::g_stamp_buffer = d_st;
  const char* nm32[6] = {"issue glds", "wait tile (sync1)", "scan", "compute+store+moments", "end barrier", "whole iteration"};
  const char* nm16[6] = {"wait own DMA (vmcnt)", "top barrier", "issue DMA k+1 + flush k-1", "sweep + barrier", "dots + moments", "whole iteration"};
  const char** nm = u16 ? nm16 : nm32;
  for (int deg = 0; deg <= 4; ++deg) {
    CK(hipMemset(d_st, 0, stn * 8));
    int slots = 0;
    for (int rep = 0; rep < 3; ++rep) {
      int rc;
      if (u16) {
        if (deg == 0) continue;
        rc = hsr_srf_integrate_moments_u16(d_cube16, npix, B, 1e-4f, 65535, d_wn, k0, klen, nb, d_planes, 1, 12, d_real, 1, 12, nullptr, 0.f, 0.f, deg, d_part, &slots, &opts, 0);
      } else {
        rc = deg == 0 ? hsr_srf_integrate(d_cube, npix, B, d_wn, k0, klen, nb, d_planes, 1, 12, &opts, 0)
                      : hsr_srf_integrate_moments(d_cube, npix, B, d_wn, k0, klen, nb, d_planes, 1, 12, d_real, 1, 12, nullptr, 0.f, 0.f, deg, d_part, &slots, &opts, 0);
      }
      if (rc) { printf("error: %s\n", hsr_last_error()); return 1; }
    }
    CK(hipDeviceSynchronize());
    std::vector<unsigned long long> ht(stn);
    CK(hipMemcpy(ht.data(), d_st, stn * 8, hipMemcpyDeviceToHost));
    if (u16 && deg == 0) continue;
    double sum[6] = {0}, nt = 0, wsum[8][6] = {{0}}, wn_[8] = {0};
    for (size_t i = 0; i < (size_t)G * NW; ++i) { if (!ht[i * 8 + 6]) continue; for (int k = 0; k < 6; ++k) { sum[k] += ht[i * 8 + k]; wsum[i % NW][k] += ht[i * 8 + k]; } nt += ht[i * 8 + 6]; wn_[i % NW] += ht[i * 8 + 6]; }
    printf("deg %d (last launch only): s_memtime ticks per group per wave (each stamp itself costs a scalar-memory round trip)\n", deg);
    for (int k = 0; k < 6; ++k) printf("  %-28s %9.1f  %5.1f %%\n", nm[k], sum[k] / nt, 100.0 * sum[k] / sum[5]);
    if (u16) for (int w = 0; w < NW; ++w) { printf("    wave %d:", w); for (int k = 0; k < 6; ++k) printf(" %7.1f", wsum[w][k] / wn_[w]); printf("\n"); }
  }
  return 0;
}
