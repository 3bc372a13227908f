// a4: exact masked percentiles on the device (np.percentile(vals[mask], [pmin, pmax]), method
// 'linear'), the limits of apply_shared_percentile_stretch (reference s2_emit/color.py:31-32).
//
// Exact order statistics by a 3-pass MSD radix select on the monotone uint32 image of the float32
// values (11 + 11 + 10 bits), all four ranks (prev/next of both percentiles) of all channels at
// once, integer histograms only (LDS-privatised, then integer global atomics -> deterministic).
// Each pass streams the planes once: 3 x 4 B per masked sample, ~4 % of the cube's bytes.
// The final interpolation mirrors NumPy's _lerp bit for bit:
//   d = float32(b - a);  t < 0.5 ? a + d*t : b - d*(1 - t)   (float64), NaN if any masked NaN.
#include <type_traits>

#include "hsr_common.h"
#include "hsr_select_dev.h"

namespace hsr {

struct SelArgs {
  const float* x;
  int64_t cs, ps;  // element (c, p) at c*cs + p*ps
  const uint8_t* mask;
  int64_t npix;
  int32_t nb;
  uint32_t* hist1;   // [nb][kHist1]
  uint32_t* hist2;   // [nb][kQ][kBins2]
  uint32_t* hist3;   // [nb][kQ][kBins3]
  SelState* state;   // [nb]
};

// r03: 1024-thread workgroups (were 256 / 512).  Every workgroup ends by adding its LDS histogram to the global one with
// integer atomics - in pass 2 a few thousand non-zero bins each, all workgroups on the same addresses; the same number of
// resident waves in a quarter / half of the workgroups: 404 -> 367 us (three 6144 x 6144 planes), 86 -> 60 us (1024 x 1024).
#ifndef HSR_SEL_THREADS
#define HSR_SEL_THREADS 1024
#endif
#ifndef HSR_ROWS_THREADS
#define HSR_ROWS_THREADS 1024
#endif
constexpr int kSelThreads = HSR_SEL_THREADS;      // workgroup of select_hist_kernel
constexpr int kRowsThreads = HSR_ROWS_THREADS;    // workgroup of select_hist_rows4_kernel
// resident waves stay the same whatever the workgroup size: the caps below count workgroups of 256 / 512 threads
#ifndef HSR_SEL_PLANE_WGS
#define HSR_SEL_PLANE_WGS (1024 * 256 / HSR_SEL_THREADS)
#endif
#ifndef HSR_SEL_ROWS4_WGS
#define HSR_SEL_ROWS4_WGS (1024 * 512 / HSR_ROWS_THREADS)
#endif

// VEC: band-major planes whose rows start 16-byte aligned (and a 4-byte aligned mask): 4 samples + 4 mask bytes
// per load, two loads in flight per thread.  The first version walked the plane sample by sample behind a
// dependent mask-byte load and ran at 1.4 TB/s.
// MODE 0: any strides, one sample per load.  MODE 1: VEC above.  (Band-last rows of 4 floats: select_hist_rows4_kernel.)
// One radix pass of channel c over the pixels of workgroup `bid` of `nblk` (kSelThreads threads): LDS-privatised histogram in
// h[NB] (+ *nancp), flushed to the global one with integer atomics.  Shared by the per-pass kernels and the one-launch form.
template <int PASS, int MODE>
__device__ __forceinline__ void hist_pass_planes(const SelArgs& a, int c, int bid, int nblk, uint32_t* h, uint32_t* nancp) {
  constexpr bool VEC = MODE == 1;
  constexpr int NBINS = PASS == 1 ? kBins1 : (PASS == 2 ? kBins2 : kBins3);
  constexpr int CP = PASS == 1 ? kPass1Copies : 1;
  constexpr int NB = PASS == 1 ? kBins1 * CP : NBINS * kLdsQ;
  uint32_t& nanc = *nancp;
  for (int i = threadIdx.x; i < NB; i += kSelThreads) h[i] = 0u;
  if (threadIdx.x == 0) nanc = 0u;
  uint32_t pre[kQ] = {0u, 0u, 0u, 0u};
  if (PASS > 1) {
#pragma unroll
    for (int q = 0; q < kQ; ++q) pre[q] = a.state[c].prefix[q];
    dedupe_prefixes(pre);
  }
  const int second = PASS > 1 ? second_query(pre) : 0;
  const uint32_t pre_second = second_prefix(pre, second);
  const bool more = more_prefixes(pre, second);
  const PrefixRaw praw = prefix_raw<(PASS > 1 ? PASS : 2)>(pre, pre_second);
  uint32_t* g = PASS == 1 ? a.hist1 + (size_t)c * kHist1
                          : (PASS == 2 ? a.hist2 + (size_t)c * kQ * kBins2 : a.hist3 + (size_t)c * kQ * kBins3);
  __syncthreads();
  const float* x = a.x + (size_t)c * a.cs;
  const int64_t stride = (int64_t)nblk * kSelThreads;
  // Pass 1 (r04): a thread keeps the bin of its last sample and a count in registers and touches LDS only when the bin changes - images
  // are smooth, a thread's eight samples of an iteration (two runs of four neighbours) mostly share their top 11 key bits, and the
  // ballot / shuffle peel of hist_add_wave (~25 instructions per sample, r02) made this the slowest pass: 148 us on three 6144 x 6144
  // planes against 88 us for pass 3, which is the time of the bytes.  No cross-lane instruction is left; on noise it is one atomic per
  // sample, as before.  (Measured with it and dropped: the LAST workgroup of a pass running the pass's scan - hist + scan one launch,
  // four launches instead of seven.  With a __threadfence() per workgroup 1024 x 1024 took 340 us instead of 52; with only a wait for
  // the workgroup's own atomics and agent-scope loads in the scan 65 us - and 338 against 307 us at 6144 x 6144: the scan's uncached
  // reads and the tail of the launch cost more than a launch boundary.)
  // (runs are compared on the raw top 11 bits; the key's bin, raw ^ 0x400 for the positive class and raw ^ 0x7ff for the negative, is
  // formed when a run is flushed)
  uint32_t run_bin = 0xffffffffu, run_cnt = 0u, run_nan = 0u;
  const uint32_t run_copy = CP > 1 ? (threadIdx.x & (CP - 1)) * kBins1 : 0u;
  auto bin_of = [](uint32_t raw) { return raw ^ ((raw & 0x400u) ? 0x7ffu : 0x400u); };
  auto add = [&](float v, bool use) {
    if (PASS == 1) {
      if (use) {
        const uint32_t b = __float_as_uint(v) >> 21;
        if (b != run_bin) {
          if (run_cnt) atomicAdd(&h[run_copy + bin_of(run_bin)], run_cnt);
          run_bin = b;
          run_cnt = 0u;
        }
        ++run_cnt;
        run_nan += v != v ? 1u : 0u;
      }
    } else {
      hist_sample<PASS, CP>(h, &nanc, pre, second, praw, more, g, v, use);
    }
  };
  if (VEC) {
    const int64_t n4 = a.npix >> 2;
    const float4* x4 = reinterpret_cast<const float4*>(x);
    const uint32_t* m4 = reinterpret_cast<const uint32_t*>(a.mask);
    // whole-wave trip count: hist_add_wave uses ballots, so every lane of a wave runs every iteration
    const int64_t first = (int64_t)bid * kSelThreads + (threadIdx.x & ~63);
    for (int64_t base = first; base < n4; base += 2 * stride) {
      const int64_t i0 = base + (threadIdx.x & 63), i1 = i0 + stride;
      const bool on0 = i0 < n4, on1 = i1 < n4;
      float4 v0 = make_float4(0.f, 0.f, 0.f, 0.f), v1 = v0;
      uint32_t mk0 = 0u, mk1 = 0u;
      if (on0) {
        v0 = ld_stream(x4 + i0);
        mk0 = m4 ? m4[i0] : 0x01010101u;
      }
      if (on1) {
        v1 = ld_stream(x4 + i1);
        mk1 = m4 ? m4[i1] : 0x01010101u;
      }
      add(v0.x, (mk0 & 0x000000ffu) != 0u);
      add(v0.y, (mk0 & 0x0000ff00u) != 0u);
      add(v0.z, (mk0 & 0x00ff0000u) != 0u);
      add(v0.w, (mk0 & 0xff000000u) != 0u);
      add(v1.x, (mk1 & 0x000000ffu) != 0u);
      add(v1.y, (mk1 & 0x0000ff00u) != 0u);
      add(v1.z, (mk1 & 0x00ff0000u) != 0u);
      add(v1.w, (mk1 & 0xff000000u) != 0u);
    }
    if (bid == 0 && threadIdx.x < 64) {   // up to 3 tail samples, one wave (ballots need the whole wave)
      const int64_t p = n4 * 4 + threadIdx.x;
      const bool on = p < a.npix;
      const float v = on ? x[p] : 0.0f;
      add(v, on && (!a.mask || a.mask[p] != 0));
    }
  } else {
    const int64_t first = (int64_t)bid * kSelThreads + (threadIdx.x & ~63);
    for (int64_t base = first; base < a.npix; base += stride) {
      const int64_t p = base + (threadIdx.x & 63);
      const bool on = p < a.npix && (!a.mask || a.mask[p] != 0);
      const float v = on ? ld_stream(x + p * a.ps) : 0.0f;
      add(v, on);
    }
  }
  if (PASS == 1) {
    if (run_cnt) atomicAdd(&h[run_copy + bin_of(run_bin)], run_cnt);
    if (run_nan) atomicAdd(&nanc, run_nan);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < NBINS; i += kSelThreads) {
    uint32_t cnt = h[i];
    if (PASS == 1) {
#pragma unroll
      for (int cp = 1; cp < CP; ++cp) cnt += h[cp * kBins1 + i];
    }
    if (cnt) atomicAdd(&g[i], cnt);
    if (PASS > 1 && second && h[NBINS + i]) atomicAdd(&g[second * NBINS + i], h[NBINS + i]);
  }
  if (PASS == 1 && threadIdx.x == 0 && nanc) atomicAdd(&a.hist1[(size_t)c * kHist1 + kBins1], nanc);
}

template <int PASS, int MODE>
__global__ __launch_bounds__(kSelThreads) void select_hist_kernel(const SelArgs a) {
  constexpr int NBINS = PASS == 1 ? kBins1 : (PASS == 2 ? kBins2 : kBins3);
  constexpr int NB = PASS == 1 ? kBins1 * kPass1Copies : NBINS * kLdsQ;
  __shared__ uint32_t h[NB];
  __shared__ uint32_t nanc;
  hist_pass_planes<PASS, MODE>(a, blockIdx.y, blockIdx.x, gridDim.x, h, &nanc);
}

// Band-last rows of exactly 4 floats (the RGB + pad images of the driver): ONE pass over the image per radix pass
// for all channels - a 16-byte load per pixel, the histograms of all nb <= 4 channels side by side in LDS (24 / 96 /
// 48 KB for three channels).  Walking the image once per channel, sample by sample (MODE 0), ran at ~1.1 TB/s on the
// 6144 x 6144 x 4 image and was half of match_pair's time; loading whole rows once per channel was worse still
// (the channel passes do not share L2 lines in time: 3x the traffic).
template <int PASS>
__device__ __forceinline__ void hist_pass_rows4(const SelArgs& a, int bid, int nblk, uint32_t* hall /*[nb][NB]*/, uint32_t (&nanc)[4],
                                                uint32_t (&pre_s)[4][kQ]) {
  constexpr int NBINS = PASS == 1 ? kBins1 : (PASS == 2 ? kBins2 : kBins3);
  constexpr int NB = PASS == 1 ? kBins1 : NBINS * kLdsQ;
  const int nb = a.nb;
  for (int i = threadIdx.x; i < nb * NB; i += kRowsThreads) hall[i] = 0u;
  if (threadIdx.x < 4) nanc[threadIdx.x] = 0u;
  if (PASS > 1 && threadIdx.x < nb * kQ) pre_s[threadIdx.x / kQ][threadIdx.x % kQ] = a.state[threadIdx.x / kQ].prefix[threadIdx.x % kQ];
  __syncthreads();
  uint32_t pre[4][kQ];
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int q = 0; q < kQ; ++q) pre[c][q] = (PASS > 1 && c < nb) ? (uint32_t)__builtin_amdgcn_readfirstlane((int)pre_s[c][q]) : 0u;   // scalar
  if (PASS > 1) {
#pragma unroll
    for (int c = 0; c < 4; ++c) dedupe_prefixes(pre[c]);
  }
  int second[4];
  uint32_t pre_second[4];
  PrefixRaw praw[4];
  bool more[4];
  uint32_t* gq[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    second[c] = PASS > 1 ? second_query(pre[c]) : 0;
    pre_second[c] = second_prefix(pre[c], second[c]);
    more[c] = more_prefixes(pre[c], second[c]);
    praw[c] = prefix_raw<(PASS > 1 ? PASS : 2)>(pre[c], pre_second[c]);
    gq[c] = PASS == 1 ? a.hist1 + (size_t)c * kHist1
                      : (PASS == 2 ? a.hist2 + (size_t)c * kQ * kBins2 : a.hist3 + (size_t)c * kQ * kBins3);
  }
  const float4* rows = reinterpret_cast<const float4*>(a.x);
#ifndef HSR_ROWS_U
#define HSR_ROWS_U 4
#endif
  constexpr int U = HSR_ROWS_U;
  const int64_t stride = (int64_t)nblk * kRowsThreads;
  const int64_t first = (int64_t)bid * kRowsThreads + (threadIdx.x & ~63);
  auto sweep = [&](auto has_mask) {
  for (int64_t base = first; base < a.npix; base += U * stride) {   // wave-uniform trip count (ballots inside)
    float4 v[U];
    bool on[U];
    uint32_t mk[U];
    // all U rows and mask bytes are requested before the first is looked at (r04: `on = on && mask[p] != 0` made the mask byte a
    // conditional load inside an exec-masked block with its own s_waitcnt vmcnt(0) - four dependent round trips per iteration instead
    // of one, and the rows kernels ran at 4.3 TB/s where the planes' passes run at 6.4)
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t p = base + u * stride + (threadIdx.x & 63);
      on[u] = p < a.npix;
      const int64_t pc = on[u] ? p : 0;
      v[u] = ld_stream(rows + pc);
      mk[u] = decltype(has_mask)::value ? (uint32_t)a.mask[pc] : 1u;   // (the branch on a.mask stands outside the loop)
    }
#pragma unroll
    for (int u = 0; u < U; ++u) on[u] = on[u] && mk[u] != 0u;
    if (PASS == 3) {
      // pass 3 (r04): a sample matches a 22-bit prefix once in ~16 000, so the wave first asks whether ANY of its 4 x nb samples of this
      // iteration matches anything (two compares per sample, one ballot) and usually moves on; the compare-and-branch groups of the
      // counting path ran for every sample and made the pass 155 us for 640 MB
      bool hit = false;
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const float e[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
#pragma unroll
        for (int c = 0; c < 4; ++c)
          if (c < nb) {
            const uint32_t hi = __float_as_uint(e[c]) >> 10;
            bool m = hi == praw[c].r0 || (second[c] && hi == praw[c].rs);
            if (more[c]) {
              const uint32_t key = f32_key(e[c]) >> 10;
              m = m || key == pre[c][2] || key == pre[c][3];
            }
            hit = hit || (m && on[u]);
          }
      }
      if (__ballot(hit) == 0) continue;                 // wave-uniform
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const float e[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
#pragma unroll
      for (int c = 0; c < 4; ++c)
        if (c < nb) hist_sample<PASS>(hall + c * NB, &nanc[c], pre[c], second[c], praw[c], more[c], gq[c], e[c], on[u]);   // c < nb is block-uniform
    }
  }
  };
  if (a.mask) sweep(std::true_type{});
  else sweep(std::false_type{});
  __syncthreads();
  for (int c = 0; c < nb; ++c) {
    uint32_t* g = PASS == 1 ? a.hist1 + (size_t)c * kHist1
                            : (PASS == 2 ? a.hist2 + (size_t)c * kQ * kBins2 : a.hist3 + (size_t)c * kQ * kBins3);
    int sec = 0;                                 // (second[c] with a run-time c)
#pragma unroll
    for (int cc = 0; cc < 4; ++cc)
      if (cc == c) sec = second[cc];
    for (int i = threadIdx.x; i < NBINS; i += kRowsThreads) {
      if (hall[c * NB + i]) atomicAdd(&g[i], hall[c * NB + i]);
      if (PASS > 1 && sec && hall[c * NB + NBINS + i]) atomicAdd(&g[sec * NBINS + i], hall[c * NB + NBINS + i]);
    }
    if (PASS == 1 && threadIdx.x == 0 && nanc[c]) atomicAdd(&a.hist1[(size_t)c * kHist1 + kBins1], nanc[c]);
  }
}

template <int PASS>
__global__ __launch_bounds__(kRowsThreads) void select_hist_rows4_kernel(const SelArgs a) {
  extern __shared__ uint32_t hall[];          // [nb][NB]
  __shared__ uint32_t nanc[4];
  __shared__ uint32_t pre_s[4][kQ];
  hist_pass_rows4<PASS>(a, blockIdx.x, gridDim.x, hall, nanc, pre_s);
}

template <int PASS>
static void launch_rows4(const SelArgs& a, hipStream_t s) {
  constexpr int NB = PASS == 1 ? kBins1 : (PASS == 2 ? kBins2 * kLdsQ : kBins3 * kLdsQ);
  const size_t lds = (size_t)a.nb * NB * sizeof(uint32_t);
  static thread_local size_t configured = 0;
  if (lds > configured) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(select_hist_rows4_kernel<PASS>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)lds);
    (void)hipGetLastError();
    configured = lds;
  }
  // (r04: above a megapixel a workgroup takes two iterations - half the workgroups, half the flush atomics onto the same global words:
  // 1448 x 1448 rows 99 -> 83 us, planes 71 -> 58; four or eight iterations, and two below a megapixel, were slower)
  const int64_t per_wg = (int64_t)kRowsThreads * 4 * (a.npix > (1 << 20) ? 2 : 1);
  int64_t gx = (a.npix + per_wg - 1) / per_wg;
  if (gx > HSR_SEL_ROWS4_WGS) gx = HSR_SEL_ROWS4_WGS;
  hipLaunchKernelGGL(select_hist_rows4_kernel<PASS>, dim3((unsigned)gx), dim3(kRowsThreads), lds, s, a);
}

// The four rank queries side by side (r04): wave q of the workgroup's first four locates query q in ITS histogram - a lane sums
// nbins / 64 consecutive bins, one shuffle scan over the 64 sums, the lane that holds the rank walks its bins.  One barrier for all four;
// block_locate (256 threads and two barriers per query, the queries one after another) was 8 barriers per scan and 24 in the
// one-workgroup select.  Every thread of the workgroup must call it; the workgroup has at least 256 threads.
__device__ __forceinline__ void locate4(const uint32_t* h0, const uint32_t* h1, const uint32_t* h2, const uint32_t* h3, int nbins,
                                        const uint32_t* ranks, uint32_t* out_bin, uint32_t* out_rem) {
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (w < kQ) {                                                // wave-uniform
    const uint32_t* h = w == 0 ? h0 : (w == 1 ? h1 : (w == 2 ? h2 : h3));
    const uint32_t r = ranks[w];
    const int per = nbins / 64;
    const uint32_t* mine = h + lane * per;
    uint32_t local = 0;
    for (int i = 0; i < per; ++i) local += mine[i];
    uint32_t v = local;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const uint32_t n = __shfl_up(v, off, 64);
      if (lane >= off) v += n;
    }
    uint32_t before = v - local;
    if (r >= before && r < v) {                                // exactly one lane (r is below the histogram's total)
      for (int i = 0; i < per; ++i) {
        const uint32_t cnt = mine[i];
        if (r < before + cnt) {
          out_bin[w] = (uint32_t)(lane * per + i);
          out_rem[w] = r - before;
          break;
        }
        before += cnt;
      }
    }
  }
  __syncthreads();
}

struct ScanLds {
  uint32_t scratch[256];
  uint32_t bins[kQ], rems[kQ], ranks[kQ];
  uint32_t total;
};

// Scan of pass PASS for channel c by one workgroup (its first 256 threads; all threads must call).
template <int PASS>
__device__ __forceinline__ void scan_pass(const SelArgs& a, int c, double qlo, double qhi, double* lohi, ScanLds& L) {
  const bool act = threadIdx.x < 256;
  SelState* st = a.state + c;
  if (PASS == 1) {
    // total masked count, then the four ranks NumPy would index
    if (act) {
      uint32_t local = 0;
      for (int i = threadIdx.x; i < kBins1; i += 256) local += a.hist1[(size_t)c * kHist1 + i];
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) local += __shfl_xor(local, off, 64);
      if ((threadIdx.x & 63) == 0) L.scratch[threadIdx.x >> 6] = local;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      const uint32_t n = L.scratch[0] + L.scratch[1] + L.scratch[2] + L.scratch[3];
      L.total = n;
      st->n = n;
      st->nan_count = a.hist1[(size_t)c * kHist1 + kBins1];
      const double q[2] = {qlo, qhi};
      for (int j = 0; j < 2; ++j) {
        uint32_t prev = 0, next = 0;
        double g = 0.0;
        if (n > 0) {
          const double vi = (double)(n - 1) * q[j];  // (n - 1) * quantile
          if (vi >= (double)(n - 1)) {
            prev = next = n - 1;
            g = vi - floor(vi);
          } else if (vi < 0.0) {
            prev = next = 0;
            g = vi - floor(vi);
          } else {
            const double f = floor(vi);
            prev = (uint32_t)f;
            next = prev + 1;
            g = vi - f;
          }
        }
        L.ranks[2 * j] = prev;
        L.ranks[2 * j + 1] = next;
        st->gamma[j] = g;
      }
    }
    __syncthreads();
    if (L.total == 0) return;
    {
      const uint32_t* h1 = a.hist1 + (size_t)c * kHist1;
      locate4(h1, h1, h1, h1, kBins1, L.ranks, L.bins, L.rems);
      if (threadIdx.x < kQ) {
        st->prefix[threadIdx.x] = L.bins[threadIdx.x];
        st->rem[threadIdx.x] = L.rems[threadIdx.x];
      }
    }
  } else {
    if (st->n == 0) {
      if (PASS == 3 && threadIdx.x == 0) lohi[2 * c] = lohi[2 * c + 1] = __longlong_as_double(0x7ff8000000000000LL);
      return;
    }
    constexpr int NBINS = PASS == 2 ? kBins2 : kBins3;
    constexpr int SHIFT = PASS == 2 ? 11 : 10;
    const uint32_t* hist = PASS == 2 ? a.hist2 + (size_t)c * kQ * kBins2 : a.hist3 + (size_t)c * kQ * kBins3;
    {
      const uint32_t* hq[kQ];
#pragma unroll
      for (int q = 0; q < kQ; ++q) {
        int src = q;                                 // the first query of the run of equal prefixes: the one that was histogrammed
        while (src > 0 && st->prefix[src - 1] == st->prefix[q]) --src;
        hq[q] = hist + (size_t)src * NBINS;
      }
      if (threadIdx.x < kQ) L.ranks[threadIdx.x] = st->rem[threadIdx.x];
      __syncthreads();
      locate4(hq[0], hq[1], hq[2], hq[3], NBINS, L.ranks, L.bins, L.rems);
    }
    if (threadIdx.x == 0) {
      for (int q = 0; q < kQ; ++q) {
        st->prefix[q] = (st->prefix[q] << SHIFT) | L.bins[q];
        st->rem[q] = L.rems[q];
      }
      if (PASS == 3) {
        for (int j = 0; j < 2; ++j) {
          const float va = key_f32(st->prefix[2 * j]), vb = key_f32(st->prefix[2 * j + 1]);
          const double t = st->gamma[j];
          const float d = vb - va;  // NumPy subtracts in the array dtype (float32)
          double r = t >= 0.5 ? (double)vb - (double)d * (1.0 - t) : (double)va + (double)d * t;
          if (st->nan_count) r = __longlong_as_double(0x7ff8000000000000LL);
          lohi[2 * c + j] = r;
        }
      }
    }
  }
}

template <int PASS>
__global__ __launch_bounds__(256) void select_scan_kernel(const SelArgs a, double qlo, double qhi, double* lohi) {
  __shared__ ScanLds L;
  scan_pass<PASS>(a, blockIdx.x, qlo, qhi, lohi, L);
}

// ---- tiny images: the whole select of a channel inside ONE workgroup (r04) --------------------------------------------------
// The reference's own tiles are 100 x 100 pixels (tiles_helpers/utils.py:223-305): seven dependent launches (memset, 3 x (histogram,
// scan)) cost ~42 us of launch latency for 120 KB of data.  Up to 32 768 pixels a channel's masked samples fit the registers of one
// 1024-thread workgroup (32 keys per thread): three radix passes over the registers with LDS histograms and the same rank / locate /
// lerp code as the scans - exact order statistics are unique, so the limits are those of the multi-launch path bit for bit.  One
// launch, no workspace traffic.  (Measured and dropped on the way, r04.  (1) COMPACTION: pass 2 appending the keys of the
// samples that match a channel's pass-1 prefix to a buffer (wave-private LDS staging, one global atomic per 192 keys) so that pass 3
// reads those instead of the image: the top 11 key bits are sign + exponent + 2 mantissa bits, so on reflectance-like data in [0, 1) a
// prefix bin holds ~12 % of the samples, not N / 2048, and the per-sample ballot / rank / staging more than doubled pass 2: 1033 us
// instead of 358 on three 6144 x 6144 planes, 2511 instead of 520 on band-last rows.  (3) MERGED scans: passes 2 and 3 running the previous pass's scan in every workgroup
// (five launches instead of seven for mid-size images): 38.7 vs 39.6 us at 600 x 600 planes, 51.2 vs 52.2 at 1024 x 1024 - and 71 vs 46 /
// 91 vs 68 us on band-last rows, where a workgroup repeats the scans of all channels.  (2) a co-resident grid running the six phases of larger images
// with spinning grid barriers in one launch - the agent-scope fences of five barriers cost more than six launch boundaries:
// 53 us against 42 at 100 x 100, 90 against 57 at 1024 x 1024.)
constexpr int kTinyKeys = 32;
constexpr int64_t kTinyMaxPix = (int64_t)kTinyKeys * kSelThreads;

__global__ __launch_bounds__(kSelThreads) void select_tiny_kernel(const SelArgs a, double qlo, double qhi, double* lohi) {
  __shared__ uint32_t h[kQ * kBins2];           // pass 1: [2048]; pass 2: [kQ][2048]; pass 3: [kQ][1024]
  __shared__ ScanLds L;
  __shared__ uint32_t pre[kQ], rem[kQ], nan_s;
  __shared__ double gam[2];
  const int c = blockIdx.x, t = threadIdx.x;
  const bool act = t < 256;
  const float* x = a.x + (size_t)c * a.cs;
  uint32_t key[kTinyKeys];
  uint32_t use = 0u;
  // sixteen mask bytes and sixteen samples are requested before the first is looked at, twice (r04: `p < npix && mask[p] != 0` ahead
  // of the sample's load made every one of the 32 a mask load, a wait, a sample load - 32 dependent round trips were most of the 22 us)
  auto gather = [&](auto has_mask) {
#pragma unroll
    for (int h = 0; h < kTinyKeys; h += 16) {
      uint32_t mk[16];
      float val[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int64_t p = (int64_t)(h + i) * kSelThreads + t;
        const int64_t pc = p < a.npix ? p : 0;
        mk[i] = decltype(has_mask)::value ? (uint32_t)a.mask[pc] : 1u;
        val[i] = x[pc * a.ps];
      }
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int64_t p = (int64_t)(h + i) * kSelThreads + t;
        const bool on = p < a.npix && mk[i] != 0u;
        key[h + i] = f32_key(on ? val[i] : 0.0f);
        use |= on ? (1u << (h + i)) : 0u;
      }
    }
  };
  if (a.mask) gather(std::true_type{});
  else gather(std::false_type{});
  for (int i = t; i < kBins1; i += kSelThreads) h[i] = 0u;
  if (t == 0) nan_s = 0u;
  __syncthreads();
  uint32_t nans = 0;
#pragma unroll
  for (int i = 0; i < kTinyKeys; ++i)
    if (use & (1u << i)) {
      atomicAdd(&h[key[i] >> 21], 1u);
      const float v = key_f32(key[i]);
      nans += v != v ? 1u : 0u;
    }
  if (nans) atomicAdd(&nan_s, nans);
  __syncthreads();
  // ranks, as scan_pass<1>
  if (act) {
    uint32_t local = 0;
    for (int i = t; i < kBins1; i += 256) local += h[i];
    L.scratch[t] = local;
  }
  __syncthreads();
  if (t == 0) {
    uint32_t n = 0;
    for (int i = 0; i < 256; ++i) n += L.scratch[i];
    L.total = n;
    const double q[2] = {qlo, qhi};
    for (int j = 0; j < 2; ++j) {
      uint32_t prev = 0, next = 0;
      double g = 0.0;
      if (n > 0) {
        const double vi = (double)(n - 1) * q[j];
        if (vi >= (double)(n - 1)) {
          prev = next = n - 1;
          g = vi - floor(vi);
        } else if (vi < 0.0) {
          prev = next = 0;
          g = vi - floor(vi);
        } else {
          const double f = floor(vi);
          prev = (uint32_t)f;
          next = prev + 1;
          g = vi - f;
        }
      }
      L.ranks[2 * j] = prev;
      L.ranks[2 * j + 1] = next;
      gam[j] = g;
    }
  }
  __syncthreads();
  if (L.total == 0) {
    if (t == 0) lohi[2 * c] = lohi[2 * c + 1] = __longlong_as_double(0x7ff8000000000000LL);
    return;
  }
  locate4(h, h, h, h, kBins1, L.ranks, L.bins, L.rems);
  if (t < kQ) {
    pre[t] = L.bins[t];
    rem[t] = L.rems[t];
  }
  __syncthreads();
  // passes 2 and 3: one histogram per query (no sharing of equal prefixes: LDS has room)
#pragma unroll
  for (int pass = 2; pass <= 3; ++pass) {
    const int nbins = pass == 2 ? kBins2 : kBins3;
    const int sh_key = pass == 2 ? 21 : 10, sh_bin = pass == 2 ? 10 : 0;
    const uint32_t bmask = pass == 2 ? 2047u : 1023u;
    for (int i = t; i < kQ * nbins; i += kSelThreads) h[i] = 0u;
    __syncthreads();
    const uint32_t p0 = pre[0], p1 = pre[1], p2 = pre[2], p3 = pre[3];
#pragma unroll
    for (int i = 0; i < kTinyKeys; ++i)
      if (use & (1u << i)) {
        const uint32_t k = key[i], kk = k >> sh_key, bin = (k >> sh_bin) & bmask;
        if (kk == p0) atomicAdd(&h[bin], 1u);
        if (kk == p1) atomicAdd(&h[nbins + bin], 1u);
        if (kk == p2) atomicAdd(&h[2 * nbins + bin], 1u);
        if (kk == p3) atomicAdd(&h[3 * nbins + bin], 1u);
      }
    __syncthreads();
    locate4(h, h + nbins, h + 2 * nbins, h + 3 * nbins, nbins, rem, L.bins, L.rems);
    if (t < kQ) {
      pre[t] = (pre[t] << (pass == 2 ? 11 : 10)) | L.bins[t];
      rem[t] = L.rems[t];
    }
    __syncthreads();
  }
  if (t == 0) {
    for (int j = 0; j < 2; ++j) {
      const float va = key_f32(pre[2 * j]), vb = key_f32(pre[2 * j + 1]);
      const double tt = gam[j];
      const float d = vb - va;  // NumPy subtracts in the array dtype (float32)
      double r = tt >= 0.5 ? (double)vb - (double)d * (1.0 - tt) : (double)va + (double)d * tt;
      if (nan_s) r = __longlong_as_double(0x7ff8000000000000LL);
      lohi[2 * c + j] = r;
    }
  }
}

static size_t hist1_bytes(int nb) { return (size_t)nb * kHist1 * 4; }
static size_t hist2_bytes(int nb) { return (size_t)nb * kQ * kBins2 * 4; }
static size_t hist3_bytes(int nb) { return (size_t)nb * kQ * kBins3 * 4; }
static size_t state_bytes(int nb) { return (size_t)nb * sizeof(SelState); }

}  // namespace hsr

using namespace hsr;

extern "C" size_t hsr_percentile_work_bytes(int32_t nb) {
  if (nb < 1) nb = 1;
  // + 128 spare bytes behind the histograms: [0, 64) the compaction counters of passes 2 / 3
  return ((state_bytes(nb) + 7) & ~(size_t)7) + hist1_bytes(nb) + hist2_bytes(nb) + hist3_bytes(nb) + 128;
}

static int select_setup(SelArgs& a, const float* x_dev, int64_t x_bs, int64_t x_ps, const uint8_t* mask_dev,
                        int64_t npix, int32_t nb, void* work_dev, const char* who) {
  HSR_REQUIRE(work_dev, HSR_ERR_INVALID, "%s: NULL workspace", who);
  HSR_REQUIRE(nb >= 1 && nb <= HSR_MAX_BANDS, HSR_ERR_UNSUPPORTED, "%s: nb=%d", who, nb);
  HSR_REQUIRE(((uintptr_t)work_dev & 7) == 0, HSR_ERR_INVALID, "%s: workspace not 8-byte aligned", who);
  a = SelArgs{};
  a.x = x_dev;
  a.cs = x_bs;
  a.ps = x_ps;
  a.mask = mask_dev;
  a.npix = npix;
  a.nb = nb;
  unsigned char* w = (unsigned char*)work_dev;
  a.state = (SelState*)w;
  w += (state_bytes(nb) + 7) & ~(size_t)7;
  a.hist1 = (uint32_t*)w;
  w += hist1_bytes(nb);
  a.hist2 = (uint32_t*)w;
  w += hist2_bytes(nb);
  a.hist3 = (uint32_t*)w;
  return HSR_OK;
}

static unsigned int* select_spare(const SelArgs& a) { return a.hist3 + hist3_bytes(a.nb) / 4; }       // the 128 spare bytes

static dim3 select_grid(int64_t npix, int nb, int pass = 1) {
  // see launch_rows4; pass 2 (every workgroup flushes up to 2 x 2048 bins onto the same global words) takes half the workgroups from a
  // megapixel on: 1024 x 1024 45.5 -> 43.6 us, 1448 x 1448 53.8 -> 51.3 (a quarter, or half below a megapixel: no better)
  const int64_t per_wg = (int64_t)kSelThreads * 8 * (npix > (1 << 20) ? 2 : 1) * (pass == 2 && npix >= (1 << 20) ? 2 : 1);
  int64_t gx = (npix + per_wg - 1) / per_wg;
  if (gx > HSR_SEL_PLANE_WGS) gx = HSR_SEL_PLANE_WGS;
  return dim3((unsigned)gx, nb);
}

// ---- per-pass interface: lets ranks all-reduce(sum) the integer histogram of each pass in between, which
// makes the order statistics exact over the union of the ranks' samples (global stretch limits).
extern "C" int hsr_percentile_begin(void* work_dev, int32_t nb, hsr_stream_t stream) {
  HSR_REQUIRE(work_dev && nb >= 1 && nb <= HSR_MAX_BANDS, HSR_ERR_INVALID, "hsr_percentile_begin: bad argument");
  return check_hip(hipMemsetAsync(work_dev, 0, hsr_percentile_work_bytes(nb), (hipStream_t)stream), "hipMemsetAsync");
}

extern "C" int hsr_percentile_hist_region(int32_t pass, int32_t nb, int64_t* offset_bytes, int64_t* count_u32) {
  HSR_REQUIRE(pass >= 1 && pass <= 3 && nb >= 1 && nb <= HSR_MAX_BANDS && offset_bytes && count_u32, HSR_ERR_INVALID,
              "hsr_percentile_hist_region: bad argument");
  const size_t o1 = (state_bytes(nb) + 7) & ~(size_t)7, o2 = o1 + hist1_bytes(nb), o3 = o2 + hist2_bytes(nb);
  *offset_bytes = (int64_t)(pass == 1 ? o1 : (pass == 2 ? o2 : o3));
  *count_u32 = (int64_t)((pass == 1 ? hist1_bytes(nb) : (pass == 2 ? hist2_bytes(nb) : hist3_bytes(nb))) / 4);
  return HSR_OK;
}

extern "C" int hsr_percentile_hist(int32_t pass, const float* x_dev, int64_t x_bs, int64_t x_ps,
                                   const uint8_t* mask_dev, int64_t npix, int32_t nb, void* work_dev,
                                   hsr_stream_t stream) {
  HSR_REQUIRE(x_dev && pass >= 1 && pass <= 3, HSR_ERR_INVALID, "hsr_percentile_hist: bad argument");
  HSR_REQUIRE(npix >= 0 && npix < ((int64_t)1 << 31), HSR_ERR_UNSUPPORTED, "hsr_percentile_hist: npix=%lld", (long long)npix);
  HSR_REQUIRE((x_ps == 1 && x_bs >= npix) || (x_bs == 1 && x_ps >= nb), HSR_ERR_INVALID,
              "hsr_percentile_hist: strides are neither band-major nor pixel-major");
  SelArgs a;
  int rc = select_setup(a, x_dev, x_bs, x_ps, mask_dev, npix, nb, work_dev, "hsr_percentile_hist");
  if (rc != HSR_OK) return rc;
  if (npix == 0) return HSR_OK;
  hipStream_t s = (hipStream_t)stream;
  const dim3 grid = select_grid(npix, nb, pass), block(kSelThreads);
  // band-major planes with 16-byte aligned rows (and a 4-byte aligned mask) take the 4-samples-per-load path
  const bool vec = x_ps == 1 && (x_bs & 3) == 0 && (((uintptr_t)x_dev) & 15) == 0 && (((uintptr_t)mask_dev) & 3) == 0;
  const bool rows4 = x_bs == 1 && x_ps == 4 && nb <= 4 && (((uintptr_t)x_dev) & 15) == 0;
  if (vec) {
    if (pass == 1) hipLaunchKernelGGL((select_hist_kernel<1, 1>), grid, block, 0, s, a);
    else if (pass == 2) hipLaunchKernelGGL((select_hist_kernel<2, 1>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((select_hist_kernel<3, 1>), grid, block, 0, s, a);
  } else if (rows4) {
    if (pass == 1) launch_rows4<1>(a, s);
    else if (pass == 2) launch_rows4<2>(a, s);
    else launch_rows4<3>(a, s);
  } else {
    if (pass == 1) hipLaunchKernelGGL((select_hist_kernel<1, 0>), grid, block, 0, s, a);
    else if (pass == 2) hipLaunchKernelGGL((select_hist_kernel<2, 0>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((select_hist_kernel<3, 0>), grid, block, 0, s, a);
  }
  HSR_LAUNCH_CHECK("select_hist_kernel");
  return HSR_OK;
}

extern "C" int hsr_percentile_scan(int32_t pass, int32_t nb, double pmin, double pmax, void* work_dev,
                                   double* lohi_dev, hsr_stream_t stream) {
  HSR_REQUIRE(pass >= 1 && pass <= 3 && lohi_dev, HSR_ERR_INVALID, "hsr_percentile_scan: bad argument");
  HSR_REQUIRE(pmin >= 0.0 && pmin <= 100.0 && pmax >= 0.0 && pmax <= 100.0, HSR_ERR_INVALID,
              "hsr_percentile_scan: percentiles must be in the range [0, 100]");
  SelArgs a;
  int rc = select_setup(a, nullptr, 0, 0, nullptr, 0, nb, work_dev, "hsr_percentile_scan");
  if (rc != HSR_OK) return rc;
  hipStream_t s = (hipStream_t)stream;
  const double qlo = pmin / 100.0, qhi = pmax / 100.0;
  if (pass == 1) hipLaunchKernelGGL(select_scan_kernel<1>, dim3(nb), dim3(256), 0, s, a, qlo, qhi, lohi_dev);
  else if (pass == 2) hipLaunchKernelGGL(select_scan_kernel<2>, dim3(nb), dim3(256), 0, s, a, qlo, qhi, lohi_dev);
  else hipLaunchKernelGGL(select_scan_kernel<3>, dim3(nb), dim3(256), 0, s, a, qlo, qhi, lohi_dev);
  HSR_LAUNCH_CHECK("select_scan_kernel");
  return HSR_OK;
}

extern "C" int hsr_percentile_limits(const float* x_dev, int64_t x_bs, int64_t x_ps, const uint8_t* mask_dev,
                                     int64_t npix, int32_t nb, double pmin, double pmax, void* work_dev,
                                     double* lohi_dev, hsr_stream_t stream) {
  HSR_REQUIRE(x_dev && work_dev && lohi_dev, HSR_ERR_INVALID, "hsr_percentile_limits: NULL pointer");
  HSR_REQUIRE(npix >= 1, HSR_ERR_UNSUPPORTED, "hsr_percentile_limits: npix=%lld", (long long)npix);
  HSR_REQUIRE(pmin >= 0.0 && pmin <= 100.0 && pmax >= 0.0 && pmax <= 100.0, HSR_ERR_INVALID,
              "hsr_percentile_limits: percentiles must be in the range [0, 100]");
  HSR_REQUIRE(npix < ((int64_t)1 << 31), HSR_ERR_UNSUPPORTED, "hsr_percentile_limits: npix=%lld", (long long)npix);
  HSR_REQUIRE((x_ps == 1 && x_bs >= npix) || (x_bs == 1 && x_ps >= nb), HSR_ERR_INVALID,
              "hsr_percentile_limits: strides are neither band-major nor pixel-major");
  if (npix <= kTinyMaxPix) {                       // a channel fits one workgroup's registers: one launch, no workspace traffic
    SelArgs a;
    int rc0 = select_setup(a, x_dev, x_bs, x_ps, mask_dev, npix, nb, work_dev, "hsr_percentile_limits");
    if (rc0 != HSR_OK) return rc0;
    hipLaunchKernelGGL(select_tiny_kernel, dim3(nb), dim3(kSelThreads), 0, (hipStream_t)stream, a, pmin / 100.0, pmax / 100.0, lohi_dev);
    HSR_LAUNCH_CHECK("select_tiny_kernel");
    return HSR_OK;
  }
  int rc = hsr_percentile_begin(work_dev, nb, stream);
  if (rc != HSR_OK) return rc;
  for (int pass = 1; pass <= 3 && rc == HSR_OK; ++pass) {
    rc = hsr_percentile_hist(pass, x_dev, x_bs, x_ps, mask_dev, npix, nb, work_dev, stream);
    if (rc == HSR_OK) rc = hsr_percentile_scan(pass, nb, pmin, pmax, work_dev, lohi_dev, stream);
  }
  return rc;
}
