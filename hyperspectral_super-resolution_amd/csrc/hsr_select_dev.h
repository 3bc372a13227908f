// Device-side pieces of the exact percentile select (csrc/hsr_select.hip) that a PRODUCER kernel needs to count the pass-1
// histogram of the values it writes (csrc/hsr_resample.hip: bilinear_up_hist_kernel): the radix key, the bin count and the
// ballot-aggregated LDS increment.
#ifndef HSR_SELECT_DEV_H_
#define HSR_SELECT_DEV_H_
#include "hsr_common.h"

namespace hsr {

constexpr int kBins1 = 2048, kBins2 = 2048, kBins3 = 1024, kQ = 4;
constexpr int kHist1 = kBins1 + 4;   // per channel: 2048 bins + [NaN count, pad, pad, pad]

__device__ __forceinline__ uint32_t f32_key(float v) {
  const uint32_t u = __float_as_uint(v);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float key_f32(uint32_t k) {
  return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}

constexpr uint32_t kNoBin = 0xffffffffu;

// Pass-1 increments.  Reflectance images put most of a wave's 64 samples into two or three of the 2048
// top-bit bins, and same-address LDS atomics serialise lane by lane; so the two most common bins of the wave are
// peeled off with a ballot + one add of the population count each, and only what is left goes out as plain
// atomics (spread-out data loses a dozen instructions and keeps its parallel atomics).
// `copies` > 1: what is left after the peel goes to the copy of the histogram chosen by the lane's low bits
// (h[copy * kBins1 + bin]) - lanes that still share a bin then hit different LDS words (spread-out data: ~20 distinct
// bins per wave, up to 8 lanes each, serialised lane by lane on one word).  The copies are added when the workgroup flushes.
template <int COPIES>
__device__ __forceinline__ void hist_add_wave(uint32_t* h, uint32_t bin) {
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const uint64_t act = __ballot(bin != kNoBin);
    if (act == 0) return;                       // wave-uniform
    const int leader = __ffsll((unsigned long long)act) - 1;
    const uint32_t lb = __shfl(bin, leader, 64);
    const uint64_t same = __ballot(bin == lb);
    if (lane == leader) atomicAdd(&h[lb], (uint32_t)__popcll(same));
    if (bin == lb) bin = kNoBin;
  }
  if (bin != kNoBin) atomicAdd(&h[(COPIES > 1 ? (lane & (COPIES - 1)) * kBins1 : 0) + bin], 1u);
}

}  // namespace hsr
#endif
