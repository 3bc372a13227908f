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

// ---- state of a select in progress and the increments of passes 2 / 3 (shared with the producer kernels of hsr_resample.hip)
struct SelState {      // one per channel
  uint32_t n;          // masked sample count
  uint32_t nan_count;  // masked NaNs (copied from the pass-1 histogram tail by scan 1)
  uint32_t prefix[kQ]; // key prefix fixed so far, per rank query
  uint32_t rem[kQ];    // rank remaining inside that prefix
  double gamma[2];     // interpolation weights of (pmin, pmax)
};

// The prev / next ranks of a percentile nearly always share their prefix (and on short-range data both percentiles do): a
// query whose prefix equals its predecessor's is not histogrammed a second time - its slot is parked on a key no sample has -
// and the scan reads the predecessor's histogram for it (select_scan_kernel).  r03 trace: pass 2 was the slowest of the three
// (168 us against 142 / 88 us on three 6144 x 6144 planes), every matching sample paying two LDS atomics.
constexpr uint32_t kNoPrefix = 0xffffffffu;      // prefixes have 11 or 22 bits
__device__ __forceinline__ void dedupe_prefixes(uint32_t (&pre)[kQ]) {
  const uint32_t p0 = pre[0], p1 = pre[1], p2 = pre[2], p3 = pre[3];
  pre[1] = p1 == p0 ? kNoPrefix : p1;
  pre[2] = p2 == p1 ? kNoPrefix : p2;
  pre[3] = p3 == p2 ? kNoPrefix : p3;
}

// LDS holds the histograms of TWO queries in passes 2 and 3 (kLdsQ): query 0 and the next query with a prefix of its own
// (`second`, 1..3; 0 = none).  A third / fourth distinct prefix - the prev / next ranks of a percentile straddling a bin
// boundary - is rare and counts straight into the global histogram.  With all four in LDS pass 2 needed 32 KB per channel
// (96 KB for the three channels of the band-last kernel: one workgroup per CU) and was the slowest pass for lack of waves in
// flight, not for its atomics.
constexpr int kLdsQ = 2;
#ifndef HSR_SEL_COPIES
#define HSR_SEL_COPIES 4
#endif
constexpr int kPass1Copies = HSR_SEL_COPIES;   // pass-1 histogram copies of select_hist_kernel (planes)
// r04: the prefixes are wave-uniform, so which of them exist is a SCALAR question: a sample is compared with query 0's prefix, with
// the second distinct one only if there is one and with a third / fourth only if there are (`more`) - the four compare-and-branch
// groups per sample of r03 (three of them against parked prefixes that nothing matches) were most of what pass 2 cost over its bytes.
// r04: a prefix also fixes the SIGN class of the samples that can match it (the key's top bit), so the match is tested on the raw
// float bits - (bits >> s) == raw prefix, one shift and one compare - and only a matching sample's bin needs the key's lower bits,
// (bits >> s') ^ flip.  The key transform (four instructions per sample and channel) is gone from passes 2 and 3.
struct PrefixRaw {
  uint32_t r0, f0;   // query 0: raw prefix, xor mask of the bin (0 for the positive class, ~0 for the negative)
  uint32_t rs, fs;   // the second distinct prefix (valid if `second`)
};
template <int PASS>
__device__ __forceinline__ PrefixRaw prefix_raw(const uint32_t (&pre)[kQ], uint32_t pre_second) {
  constexpr uint32_t top = PASS == 2 ? 0x400u : 0x200000u, all = PASS == 2 ? 0x7ffu : 0x3fffffu;
  auto raw = [&](uint32_t p) { return (p & top) ? (p & (top - 1u)) : (~p & all); };
  auto flip = [&](uint32_t p) { return (p & top) ? 0u : 0xffffffffu; };
  return PrefixRaw{raw(pre[0]), flip(pre[0]), raw(pre_second), flip(pre_second)};
}

template <int PASS, int COPIES = 1>
__device__ __forceinline__ void hist_sample(uint32_t* h, uint32_t* nanc, const uint32_t (&pre)[kQ], int second, const PrefixRaw& pr,
                                            bool more, uint32_t* g, float v, bool use) {
  if (PASS == 1) {
    const uint32_t k = f32_key(v);
    hist_add_wave<COPIES>(h, use ? (k >> 21) : kNoBin);
    if (use && v != v) atomicAdd(nanc, 1u);
  } else {
    constexpr int NBINS = PASS == 2 ? kBins2 : kBins3;
    const uint32_t bits = __float_as_uint(v);
    const uint32_t hi = PASS == 2 ? (bits >> 21) : (bits >> 10);              // the raw image of the prefix
    const uint32_t lo = PASS == 2 ? (bits >> 10) : bits;                        // the bin's bits, before the class's flip
    constexpr uint32_t bmask = PASS == 2 ? 2047u : 1023u;
    if (use) {
      if (hi == pr.r0) atomicAdd(&h[(lo ^ pr.f0) & bmask], 1u);
      if (second) {                                   // scalar
        if (hi == pr.rs) atomicAdd(&h[NBINS + ((lo ^ pr.fs) & bmask)], 1u);
        if (more) {                                   // scalar, rare: the prev / next ranks of a percentile straddle a bin boundary
          const uint32_t k = f32_key(v);
          const uint32_t key = PASS == 2 ? (k >> 21) : (k >> 10);
          const uint32_t bin = PASS == 2 ? ((k >> 10) & 2047u) : (k & 1023u);
#pragma unroll
          for (int q = 2; q < kQ; ++q)
            if (q != second && key == pre[q]) atomicAdd(&g[q * NBINS + bin], 1u);
        }
      }
    }
  }
}

// the second distinct prefix itself and whether a third exists (after dedupe_prefixes; `second` from second_query)
__device__ __forceinline__ uint32_t second_prefix(const uint32_t (&pre)[kQ], int second) {
  return second == 1 ? pre[1] : (second == 2 ? pre[2] : (second == 3 ? pre[3] : kNoPrefix));
}
__device__ __forceinline__ bool more_prefixes(const uint32_t (&pre)[kQ], int second) {
  return (second == 1 && (pre[2] != kNoPrefix || pre[3] != kNoPrefix)) || (second == 2 && pre[3] != kNoPrefix);
}

// first query after 0 whose prefix is its own (after dedupe_prefixes); 0 if there is none
__device__ __forceinline__ int second_query(const uint32_t (&pre)[kQ]) {
  return pre[1] != kNoPrefix ? 1 : (pre[2] != kNoPrefix ? 2 : (pre[3] != kNoPrefix ? 3 : 0));
}

}  // namespace hsr
#endif
