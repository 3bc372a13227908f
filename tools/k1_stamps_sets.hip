// Phase stamps of the PRODUCTION K1 (deg 3) on N candidate operand sets of one process: which phase of a tile iteration
// grows when the operands lie in a "slow" stretch of device memory (profiles/r03_placement_mechanism.md).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DHSR_PHASE_STAMPS tools/k1_stamps_sets.hip \
//         hyperspectral_super-resolution_amd/csrc/hsr_srf.hip hyperspectral_super-resolution_amd/csrc/hsr_lib.hip -o tools/k1_stamps_sets
//   tools/k1_stamps_sets [sets=12] [pitch_gb=16]
// Shares only (the stamps themselves cost time); the event time per set classifies it.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
#include "../include/hsr.h"
namespace hsr { extern unsigned long long* g_stamp_buffer; extern unsigned long long* g_stamp_buffer2; extern uint32_t* g_stamp_buffer3; }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

int main(int argc, char** argv) {
  const int H = 1024, W = 1024, B = 285, nb = 12, deg = 3;
  const int64_t npix = (int64_t)H * W;
  const int nsets = argc > 1 ? atoi(argv[1]) : 12;
  const double pitch = argc > 2 ? atof(argv[2]) : 16.0;
  hsr_srf_options opts = {64, 0, 0, 0};
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
  float* d_wn; double* d_part; unsigned long long* d_st;
  CK(hipMalloc(&d_wn, wn.size() * 4)); CK(hipMalloc(&d_part, hsr_partials_bytes(nb, 4)));
  CK(hipMemcpy(d_wn, wn.data(), wn.size() * 4, hipMemcpyHostToDevice));
  const int G = 1024, NW = 8; const size_t stn = (size_t)G * NW * 8;
  CK(hipMalloc(&d_st, stn * 8));
  std::vector<float> h((size_t)npix * B);
  { uint32_t s = 12345; for (auto& x : h) { s = s * 1664525u + 1013904223u; x = (s >> 8) * (0.6f / 16777216.f); } }
  struct Set { float *cube, *real, *out; };
  std::vector<Set> sets;
  for (int i = 0; i < nsets; ++i) {
    if (i) { void* sp; if (hipMalloc(&sp, (size_t)(pitch * (1ull << 30))) != hipSuccess) { (void)hipGetLastError(); break; } }
    Set s;
    if (hipMalloc(&s.cube, npix * B * 4) != hipSuccess || hipMalloc(&s.real, (size_t)nb * npix * 4) != hipSuccess ||
        hipMalloc(&s.out, (size_t)nb * npix * 4) != hipSuccess) { (void)hipGetLastError(); break; }
    CK(hipMemcpy(s.cube, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(s.real, h.data(), (size_t)nb * npix * 4, hipMemcpyHostToDevice));
    sets.push_back(s);
  }
  hsr::g_stamp_buffer = d_st;
  unsigned long long* d_st2; CK(hipMalloc(&d_st2, (size_t)G * 4 * 8)); hsr::g_stamp_buffer2 = d_st2;
  uint32_t* d_st3; CK(hipMalloc(&d_st3, (size_t)G * 64 * 4)); CK(hipMemset(d_st3, 0, (size_t)G * 64 * 4)); hsr::g_stamp_buffer3 = d_st3;
  const char* nm[6] = {"issue glds", "wait tile", "scan", "compute+store+mom", "end barrier", "iteration"};
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  int slots = 0;
  auto k1 = [&](const Set& s) {
    int rc = hsr_srf_integrate_moments(s.cube, npix, B, d_wn, k0, klen, nb, s.out, 1, 12, s.real, 1, 12, nullptr, 0.f, 0.f, deg, d_part, &slots, &opts, 0);
    if (rc) { printf("error: %s\n", hsr_last_error()); exit(1); }
  };
  for (int i = 0; i < 300; ++i) k1(sets[0]);      // settle the power state
  CK(hipDeviceSynchronize());
  printf("set   ms(best of 3)");
  for (int k = 0; k < 6; ++k) printf(" %18s", nm[k]);
  printf("   (s_memtime ticks per group per wave, last launch)\n");
  for (size_t i = 0; i < sets.size(); ++i) {
    k1(sets[i]);
    float best = 1e9f;
    for (int r = 0; r < 3; ++r) {
      CK(hipMemsetAsync(d_st, 0, stn * 8, 0));
      CK(hipEventRecord(e0, 0)); k1(sets[i]); CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    std::vector<unsigned long long> ht(stn);
    CK(hipMemcpy(ht.data(), d_st, stn * 8, hipMemcpyDeviceToHost));
    double sum[6] = {0}, nt = 0;
    for (size_t j = 0; j < (size_t)G * NW; ++j) { if (!ht[j * 8 + 6]) continue; for (int k = 0; k < 6; ++k) sum[k] += ht[j * 8 + k]; nt += ht[j * 8 + 6]; }
    printf("%3zu   %.4f       ", i, best);
    for (int k = 0; k < 6; ++k) printf(" %18.1f", sum[k] / nt);
    printf("\n");
    {   // workgroup timeline of the last launch (100 MHz REFCLK ticks = 10 ns): entry / exit relative to the first entry
      std::vector<unsigned long long> h2((size_t)G * 4);
      CK(hipMemcpy(h2.data(), d_st2, h2.size() * 8, hipMemcpyDeviceToHost));
      unsigned long long t0 = ~0ull; int nwg = 0;
      for (int b = 0; b < G; ++b) if (h2[b * 4 + 1]) { if (h2[b * 4] < t0) t0 = h2[b * 4]; ++nwg; }
      std::vector<double> dur, endt; double xmax[8] = {0}, xsum[8] = {0}; int xn[8] = {0}; double begmax = 0;
      for (int b = 0; b < G; ++b) if (h2[b * 4 + 1]) {
        double bg = (h2[b * 4] - t0) * 0.01, en = (h2[b * 4 + 1] - t0) * 0.01; int x = (int)(h2[b * 4 + 2] & 7);
        dur.push_back(en - bg); endt.push_back(en); if (bg > begmax) begmax = bg;
        if (en > xmax[x]) xmax[x] = en; xsum[x] += en - bg; ++xn[x];
      }
      std::sort(dur.begin(), dur.end()); std::sort(endt.begin(), endt.end());
      printf("      %d WGs: last entry +%.1f us; WG duration min %.1f med %.1f p95 %.1f max %.1f us; exit med %.1f p95 %.1f last %.1f us\n", nwg, begmax,
             dur.front(), dur[dur.size() / 2], dur[dur.size() * 95 / 100], dur.back(), endt[endt.size() / 2], endt[endt.size() * 95 / 100], endt.back());
      printf("      per XCC (n, mean duration, last exit):");
      for (int x = 0; x < 8; ++x) printf("  %d:%d %.1f %.1f", x, xn[x], xn[x] ? xsum[x] / xn[x] : 0.0, xmax[x]);
      printf("\n");
      CK(hipMemset(d_st2, 0, (size_t)G * 4 * 8));
      // per round (= one contiguous 37 MB stretch of the cube: groups k*512 .. k*512+511): mean iteration time over the workgroups
      std::vector<uint32_t> h3((size_t)G * 64);
      CK(hipMemcpy(h3.data(), d_st3, h3.size() * 4, hipMemcpyDeviceToHost));
      printf("      us per round:");
      for (int k = 0; k < 32; ++k) { double sm = 0; int n = 0; for (int b = 0; b < 512; ++b) if (h3[(size_t)b * 64 + k]) { sm += h3[(size_t)b * 64 + k]; ++n; } printf(" %.2f", n ? sm / n * 0.01 : 0.0); }
      printf("\n");
      CK(hipMemset(d_st3, 0, (size_t)G * 64 * 4));
    }
  }
  return 0;
}
