// K1 (+ fused K2): SRF band integration of an (npix, B) float32 cube on gfx950.
//
// Replaces the hot loop of pseudo_s2_srf_integral (reference s2_emit/synth.py:32-43).  The reference
// makes 13 full-cube float64 passes; here the cube is streamed from HBM exactly once.
//
// Data flow per workgroup (512 threads = 8 waves, 2 workgroups resident per CU):
//   1. a tile of 64 consecutive pixels (64*B*4 bytes, one linear 16-byte-aligned slab because the
//      cube is pixel-major) goes HBM -> LDS with global_load_lds_dwordx4 (no VGPR round trip,
//      1 KiB per wave instruction, fully coalesced), marked non-temporal: the cube is read once,
//      and without the hint the stream thrashes L2 against the output lines (-12 % on the kernel).
//      ~73 KB in flight per workgroup.
//   2. one linear ds_read_b128 sweep flags pixels holding a non-finite sample.
//   3. lane = pixel, wave = band group: each band is a short dot product over its SRF support read
//      from LDS with a row stride of B words (B odd -> bank-conflict free).  The weight taps of all
//      bands (a few hundred floats, 16-byte aligned segments) are staged into LDS once per
//      workgroup and read as broadcast ds_read_b128 - no scalar-load latency inside the tile loop
//      (the first version fetched them with s_load per tile and spent ~5 us per tile waiting).
//      Flagged pixels take the dense product so that 0*Inf -> NaN poisons exactly the bands the
//      reference poisons (synth.py:41 multiplies all B samples of every band).
//   4. output.  Pixel-major (band-last) output is staged in LDS as the tile's contiguous
//      [pixel][band] slab and flushed with 16-byte stores in the next iteration, after that tile's
//      DMA has been issued: one contiguous ~3 KB write per tile.  (Band-major planes are stored directly: nb scattered
//      256-B segments per tile; measured 10 % slower on the whole kernel because of the write
//      pattern, although the planes are only 4 % of the bytes.)  With DEG > 0 the same lane
//      also accumulates the Vandermonde power sums of (x = plane value, y = real S2 value) in
//      float64 registers; they are reduced over the wave by a fixed butterfly and written to a
//      per-workgroup slot (no float atomics -> bitwise reproducible).
// HBM-bound by construction: 4*B bytes in, 4*nb (+4*nb+1) bytes out/in per pixel, ~0.35 kflop.
#include "hsr_common.h"

namespace hsr {

struct SrfBands {
  int32_t k0[HSR_MAX_BANDS];
  int32_t klen[HSR_MAX_BANDS];
  int32_t woff[HSR_MAX_BANDS];  // offset (floats, multiple of 4) of the band's taps in the LDS weight area
  int8_t band_of[8][2];         // band handled by (group, slot), -1 = none: balanced over the 8 groups (srf_common)
};

constexpr int kWeightCap = 1024;  // floats of LDS reserved for compact weight taps (4 KiB)

struct SrfArgs {
  const float* cube;
  int64_t npix;
  int64_t ntiles;
  int32_t B;
  int32_t ldsB;  // LDS row stride in words (odd)
  int32_t wtaps; // floats used in the LDS weight area (0: weights do not fit, read them from global)
  const float* wn;
  SrfBands bands;
  int32_t nb;
  float* out;        // element (b, p) at out[b * out_bs + p * out_ps]
  int64_t out_bs, out_ps;
  const float* real;  // element (b, p) at real[b * real_bs + p * real_ps]
  int64_t real_bs, real_ps;
  const uint8_t* mask;
  float min_x, min_y;
  double* partials;
  int32_t slots;
  // uint16 tiles (srf_u16_kernel): cube points at uint16 samples, x = float(u) * scale, u == nodata -> NaN
  int32_t u16;
  float scale;
  uint32_t nodata;   // > 0xffff: no nodata value
#ifdef HSR_PHASE_STAMPS
  unsigned long long* stamps;
#endif
};

#ifdef HSR_PHASE_STAMPS
// Diagnostic build only (tools/k1_lab): per-phase s_memtime stamps, written to a buffer nothing else
// reads.  Never compiled into libhsr_mi355x.so.
unsigned long long* g_stamp_buffer = nullptr;
__device__ __forceinline__ unsigned long long phase_stamp() {
  unsigned long long t;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  return t;
}
#define HSR_STAMP(var) const unsigned long long var = phase_stamp()
#else
#define HSR_STAMP(var)
#endif

constexpr int kGroups = 8;                          // band groups per workgroup (band = group + 8 * slot)
constexpr int kBandSlots = HSR_MAX_BANDS / kGroups;  // 2 bands per thread
constexpr int kTapChunk = 16;                       // taps per unrolled dot-product chunk
constexpr int kScanBatch = 5;                       // ds_read_b128 in flight per thread in the sweep

// Tile geometry: P pixels per LDS tile, 8*P threads per workgroup.
//   P = 64: 512 threads (8 waves), lane = pixel, wave = band group;      2 workgroups per CU
//   P = 32: 256 threads (4 waves), lane&31 = pixel, each wave half = one band group; 4 per CU
// (Measured and dropped: P = 8, one-wave workgroups with wave-local barriers, ~14 per CU: 0.2535 ms; P = 128,
//  one 1024-thread workgroup per CU: 0.234 ms; P = 64: 0.2133 ms on the same box.)
// Both keep 16 waves (4 per SIMD, <= 128 VGPRs) and ~146 KB of LDS tiles per CU; the smaller tile
// gives the CU's memory pipe four queued customers instead of two (see DESIGN.md, K1 tuning).
static int g_tile_pixels = 64;
// CUs left without a persistent K1 workgroup so that small kernels on another stream (slot reduction,
// RCCL exchange, solve) can run while K1 of the next tile owns the rest of the chip.
static int g_reserved_cus = 0;
// uint16 tiles: 1 = double-buffered kernel where it fits (default), 0 = single-buffer kernel (A/B switch)
static int g_u16_ring = 1;

typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// Workgroup barrier that orders LDS accesses only: s_waitcnt lgkmcnt(0) (leaves vmcnt untouched, so
// the plane stores just issued are not waited for) + s_barrier.
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// Loads the compiler's waitcnt pass does not see (see load_targets in the kernel): the caller must wait
// (s_waitcnt vmcnt(0)) before the first use and pin the registers behind that wait.
__device__ __forceinline__ float load_f32_async(const float* p) {
  float v;
  asm volatile("global_load_dword %0, %1, off" : "=v"(v) : "v"(p) : "memory");
  return v;
}
__device__ __forceinline__ uint32_t load_u8_async(const uint8_t* p) {
  uint32_t v;
  asm volatile("global_load_ubyte %0, %1, off" : "=v"(v) : "v"(p) : "memory");
  return v;
}

// x*0 is NaN exactly when x is NaN or +-Inf; four of them chained cost 4 VALU ops.
__device__ __forceinline__ bool any_nonfinite4(const float4& v) {
  float z = v.x * 0.0f;
  z = fmaf(v.y, 0.0f, z);
  z = fmaf(v.z, 0.0f, z);
  z = fmaf(v.w, 0.0f, z);
  return z != z;
}

// Flush the staged [pixel][band] slab of the previous tile: contiguous in HBM, 16 bytes per lane.
template <int T>
__device__ __forceinline__ void flush_stage(const float* ostage, float* out, int64_t pix0, int npx, int ops, int t) {
  const int n4 = (npx * ops) >> 2;  // ops is a multiple of 4
  float4* dst = reinterpret_cast<float4*>(out + pix0 * ops);
  const float4* src = reinterpret_cast<const float4*>(ostage);
  for (int i = t; i < n4; i += T) st_stream(dst + i, src[i]);
}

template <int DEG, bool FAST, bool WLDS, int P, bool OUTV>
__global__ __launch_bounds__(8 * P, 4) void srf_kernel(const SrfArgs a) {
  constexpr int T = 8 * P;
  constexpr int NW = T / 64;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* tile = reinterpret_cast<float*>(smem);
  const int B = a.B;
  const int ldsB = a.ldsB;
  uint32_t* flags = reinterpret_cast<uint32_t*>(smem + (size_t)P * ldsB * 4);
  const float* wl = reinterpret_cast<const float*>(flags + 64);  // 16-byte aligned
  float* ostage = const_cast<float*>(wl) + (WLDS ? a.wtaps : 0);  // [P][out_ps] output slab (OUTV only)
  const int ops = (int)a.out_ps;
  int64_t prev_pix0 = 0;
  int prev_npx = 0;

  const int t = threadIdx.x;
  const int lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int pl = t % P;     // pixel of this thread inside the tile
  const int grp = t / P;    // band group 0..7 (wave-uniform for P = 64, per half-wave for P = 32)
  const int nchunk = P * B / 4;  // 16-byte chunks of a full tile (P is a multiple of 4)

  // the (at most two) bands of this thread, fixed for the whole launch
  int bk0[kBandSlots], bkl[kBandSlots], bwo[kBandSlots], bidx[kBandSlots];
  bool bval[kBandSlots];
#pragma unroll
  for (int j = 0; j < kBandSlots; ++j) {
    const int b = a.bands.band_of[grp][j];   // balanced assignment, not grp + 8*j
    bval[j] = b >= 0;
    bidx[j] = bval[j] ? b : 0;
    const int bb = bidx[j];
    bk0[j] = a.bands.k0[bb];
    bkl[j] = bval[j] ? a.bands.klen[bb] : 0;
    bwo[j] = a.bands.woff[bb];
  }

  if (WLDS) {  // compact weight taps -> LDS, once per workgroup (visible after the first barrier below)
    float* wlw = const_cast<float*>(wl);
    for (int b = 0; b < a.nb; ++b) {
      const int kl = a.bands.klen[b];  // whole 16-tap chunks inside [0, B) (srf_common)
      for (int i = t; i < kl; i += T) wlw[a.bands.woff[b] + i] = a.wn[(size_t)b * B + a.bands.k0[b] + i];
    }
  }

#ifdef HSR_PHASE_STAMPS
  unsigned long long stamp_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
  constexpr int M = DEG > 0 ? moment_count(DEG) : 1;
  double acc_m[kBandSlots][M];
  if (DEG > 0) {
#pragma unroll
    for (int j = 0; j < kBandSlots; ++j)
#pragma unroll
      for (int m = 0; m < M; ++m) acc_m[j][m] = 0.0;
  }

  for (int64_t tileidx = blockIdx.x; tileidx < a.ntiles; tileidx += gridDim.x) {
    const int64_t pix0 = tileidx * P;
    const int64_t left = a.npix - pix0;
    const int npx = left < P ? (int)left : P;
    const float* src = a.cube + pix0 * B;
    const bool pvalid = pl < npx;

    // operands of the fused fit (2 target values + 1 mask byte per thread).  hipcc drains vmcnt to 0
    // whenever ordinary VGPR loads and LDS-DMA mix (before the DMA issue if the loads come first, before
    // the loads if they come second), which exposed one extra HBM round trip per tile (+20 us on the
    // kernel).  So these loads are issued from inline asm, invisible to the waitcnt pass, right after the
    // DMA; they retire under the same wait as the tile (explicit vmcnt(0) after the barrier) and the
    // registers are pinned there so that no use can be scheduled ahead of it.
    float yv[kBandSlots];
    uint32_t mraw = 1u;
    auto load_targets = [&]() {
      if (DEG > 0) {
        const int64_t pc = pvalid ? pix0 + pl : a.npix - 1;   // clamped: always a valid address, no branch
#pragma unroll
        for (int j = 0; j < kBandSlots; ++j) {
          const int bb = bidx[j];
          yv[j] = load_f32_async(a.real + bb * a.real_bs + pc * a.real_ps);
        }
        if (a.mask != nullptr) mraw = load_u8_async(a.mask + pc);
      }
    };
    auto wait_targets = [&]() {
      if (DEG > 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        static_assert(kBandSlots == 2, "pin list below");
        asm volatile("" : "+v"(yv[0]), "+v"(yv[1]), "+v"(mraw));
      }
    };

    HSR_STAMP(st0);
    if (t < P) flags[t] = 0u;

    const bool fast_tile = FAST && npx == P;
    // Order matters (measured, and checked in the .s): the DMA is issued FIRST - anything ahead of it is
    // dead time (flushing the previous slab first cost +12 us on the kernel); then the small target loads
    // (inline asm, see above); then the flush of the previous slab, whose ds_reads see the pending LDS-DMA
    // and make hipcc wait vmcnt(0) - i.e. it runs once the tile (and the targets) have landed, just ahead
    // of the barrier that waits for the same thing.
    if (fast_tile) {
      const char* srcb = reinterpret_cast<const char*>(src);
      for (int c0 = wave * 64; c0 < nchunk; c0 += T) {  // c0 is wave-uniform
        const int c = c0 + lane;
        if (c < nchunk)
          __builtin_amdgcn_global_load_lds((gptr_t)(srcb + (size_t)c * 16), (lptr_t)(smem + (size_t)c0 * 16),
                                           16, 0, kGldsStream);
      }
      load_targets();
      if (OUTV) flush_stage<8 * P>(ostage, a.out, prev_pix0, prev_npx, ops, t);
      HSR_STAMP(st1);
      __syncthreads();
      wait_targets();
      HSR_STAMP(st2);
      // non-finite sweep: kScanBatch independent ds_read_b128 in flight per thread (a serial
      // read->wait->test loop cost 3.7k cycles per tile; batched it is LDS-bandwidth bound).
      // Indices past the tile are clamped to its last chunk (a harmless re-read, no predication).
      const float4* t4 = reinterpret_cast<const float4*>(smem);
      for (int c0 = t; c0 < nchunk; c0 += T * kScanBatch) {
        float4 v[kScanBatch];
#pragma unroll
        for (int u = 0; u < kScanBatch; ++u) {
          const int c = c0 + u * T;
          v[u] = t4[c < nchunk ? c : nchunk - 1];
        }
        bool bad = false;
#pragma unroll
        for (int u = 0; u < kScanBatch; ++u) bad |= any_nonfinite4(v[u]);
        if (bad) {  // rare
#pragma unroll
          for (int u = 0; u < kScanBatch; ++u) {
            const int c = c0 + u * T;
            const int e = (c < nchunk ? c : nchunk - 1) * 4;
            if (!finite_f32(v[u].x)) flags[(e + 0) / B] = 1u;
            if (!finite_f32(v[u].y)) flags[(e + 1) / B] = 1u;
            if (!finite_f32(v[u].z)) flags[(e + 2) / B] = 1u;
            if (!finite_f32(v[u].w)) flags[(e + 3) / B] = 1u;
          }
        }
      }
#ifdef HSR_PHASE_STAMPS
      HSR_STAMP(st3);
      if (a.stamps) { stamp_acc[0] += st1 - st0; stamp_acc[1] += st2 - st1; stamp_acc[2] += st3 - st2; }
#endif
    } else {
      // generic loader: any 4-byte alignment, any B, ragged last tile.  One pixel row per wave step.
      load_targets();
      if (OUTV) flush_stage<8 * P>(ostage, a.out, prev_pix0, prev_npx, ops, t);
      __syncthreads();  // flags zeroed before anybody sets one
      wait_targets();
      for (int pp = wave; pp < npx; pp += NW) {
        bool bad = false;
        for (int k = lane; k < B; k += 64) {
          const float v = ld_stream(src + (size_t)pp * B + k);
          bad |= !finite_f32(v);
          tile[pp * ldsB + k] = v;
        }
        if (bad) flags[pp] = 1u;
      }
    }
    __syncthreads();

    HSR_STAMP(st4);
    const bool slow = flags[pl] != 0u;
    const float* v = tile + pl * ldsB;
    float accv[kBandSlots];
#pragma unroll
    for (int j = 0; j < kBandSlots; ++j) {
      float acc = 0.0f;
      const float* vs = v + bk0[j];
      if (WLDS) {
        // [k0, k0+kl) was widened by srf_common to whole 16-tap chunks that stay inside this pixel's
        // row; the added taps carry weight 0 and (for an unflagged pixel) multiply finite samples,
        // so the sum is bit-identical to the exact-support sum.  One chunk = 4 broadcast
        // ds_read_b128 (weights) + 16 ds_read_b32 (samples, stride ldsB words: conflict-free)
        // issued together, then a 16-deep fma chain: no serial remainder loop.
        const float4* w4 = reinterpret_cast<const float4*>(wl + bwo[j]);
        for (int i0 = 0; i0 < bkl[j]; i0 += kTapChunk) {
          float4 ww[kTapChunk / 4];
          float xv[kTapChunk];
#pragma unroll
          for (int u = 0; u < kTapChunk / 4; ++u) ww[u] = w4[(i0 >> 2) + u];
#pragma unroll
          for (int u = 0; u < kTapChunk; ++u) xv[u] = vs[i0 + u];
#pragma unroll
          for (int u = 0; u < kTapChunk / 4; ++u) {
            acc = fmaf(ww[u].x, xv[4 * u + 0], acc);
            acc = fmaf(ww[u].y, xv[4 * u + 1], acc);
            acc = fmaf(ww[u].z, xv[4 * u + 2], acc);
            acc = fmaf(ww[u].w, xv[4 * u + 3], acc);
          }
        }
      } else {
        const float* ws = a.wn + (size_t)bidx[j] * B + bk0[j];
        for (int i = 0; i < bkl[j]; ++i) acc = fmaf(ws[i], vs[i], acc);
      }
      accv[j] = acc;
    }
    if (slow) {  // rare: the dense product reproduces the reference's Inf/NaN classification per band
#pragma unroll
      for (int j = 0; j < kBandSlots; ++j) {
        if (bval[j]) {
          const float* w = a.wn + (size_t)bidx[j] * B;
          float acc = 0.0f;
          for (int k = 0; k < B; ++k) acc = fmaf(w[k], v[k], acc);
          accv[j] = acc;
        }
      }
    }
#pragma unroll
    for (int j = 0; j < kBandSlots; ++j) {
      if (bval[j]) {
        const float acc = accv[j];
        if (OUTV) ostage[pl * ops + bidx[j]] = acc;
        else if (pvalid) st_stream(a.out + bidx[j] * a.out_bs + (pix0 + pl) * a.out_ps, acc);
        if (DEG > 0) {
          const float y = yv[j];
          const bool ok = pvalid && mraw != 0u && finite_f32(acc) && finite_f32(y) && acc > a.min_x && y > a.min_y;
          if (ok) {
            const double xd = (double)acc, yd = (double)y;
            acc_m[j][0] += 1.0;
            acc_m[j][2 * DEG + 1] += yd;
            double pw = 1.0;
#pragma unroll
            for (int k = 1; k <= 2 * DEG; ++k) {
              pw *= xd;
              acc_m[j][k] += pw;
              if (k <= DEG) acc_m[j][2 * DEG + 1 + k] += pw * yd;
            }
          }
        }
      }
    }
    prev_pix0 = pix0;
    prev_npx = npx;
    HSR_STAMP(st5);
    // The tile and the flags are rewritten by the next iteration: an LDS-only hazard.  A plain
    // __syncthreads() here would also wait (vmcnt(0)) for the plane stores just issued.
    lds_barrier();
#ifdef HSR_PHASE_STAMPS
    {
      HSR_STAMP(st6);
      if (a.stamps) { stamp_acc[3] += st5 - st4; stamp_acc[4] += st6 - st5; stamp_acc[5] += st6 - st0; stamp_acc[6] += 1; }
    }
#endif
  }
  if (OUTV) flush_stage<8 * P>(ostage, a.out, prev_pix0, prev_npx, ops, t);  // last tile (after the barrier)
#ifdef HSR_PHASE_STAMPS
  if (a.stamps && lane == 0)
    for (int k = 0; k < 8; ++k) a.stamps[((size_t)blockIdx.x * NW + wave) * 8 + k] = stamp_acc[k];
#endif

  if (DEG > 0) {
    // fixed butterfly over the P lanes that share a band (whole wave for P = 64, half wave for P = 32)
#pragma unroll
    for (int j = 0; j < kBandSlots; ++j) {
#pragma unroll
      for (int m = 0; m < M; ++m) {
        double sv = acc_m[j][m];
#pragma unroll
        for (int off = (P < 64 ? P : 64) / 2; off >= 1; off >>= 1) sv += __shfl_xor(sv, off, 64);
        if (bval[j] && (lane % P) == 0)
          a.partials[((size_t)bidx[j] * M + m) * a.slots + blockIdx.x] = sv;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// K1 on uint16 tiles: the training tiles on disk are uint16 reflectance x 10000 with
// nodata 65535 (reference writer tiles_helpers/utils.py:362-374).  Same data flow as srf_kernel, but
// the LDS-DMA moves the 2-byte samples (half the HBM bytes) and the decode x = float(u) * scale
// happens on the way out of LDS, one separately rounded multiply per tap, so that the planes and the
// moments are bit-identical to hsr_tile_decode_u16 followed by the float32 kernel.  A pixel holding a
// nodata sample decodes to NaN there, and 0 * NaN poisons every band: flagged pixels are NaN in all
// bands.  64-pixel tiles (P*B*2 = 128*B bytes: always whole 16-byte chunks), 8 waves, lane = pixel.
template <int DEG, bool FAST, bool OUTV>
__global__ __launch_bounds__(512, 4) void srf_u16_kernel(const SrfArgs a) {
  constexpr int P = 64, T = 512;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint16_t* tile = reinterpret_cast<uint16_t*>(smem);
  const int B = a.B;
  uint32_t* flags = reinterpret_cast<uint32_t*>(smem + (size_t)P * B * 2);
  const float* wl = reinterpret_cast<const float*>(flags + 64);
  const bool wlds = a.wtaps > 0;
  float* ostage = const_cast<float*>(wl) + a.wtaps;
  const int ops = (int)a.out_ps;
  int64_t prev_pix0 = 0;
  int prev_npx = 0;

  const int t = threadIdx.x;
  const int lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  // lane -> pixel: lanes 0..31 take the even pixels of the tile, lanes 32..63 the odd ones.  A pixel row is
  // 2*B bytes = B/2 dwords (142.5 for B = 285), so with lane = pixel neighbouring rows alternate between two
  // bank phases and 10 of 16 lane pairs collide; inside each half-wave the rows now start B dwords apart
  // (odd -> conflict-free, as in the float32 kernel).
  const int pl = 2 * (lane & 31) + (lane >> 5), grp = wave;
  const float scale = a.scale;
  const uint32_t nd2 = (a.nodata & 0xffffu) * 0x00010001u;
  const bool has_nodata = a.nodata <= 0xffffu;

  int bk0[kBandSlots], bkl[kBandSlots], bwo[kBandSlots], bidx[kBandSlots];
  bool bval[kBandSlots];
#pragma unroll
  for (int j = 0; j < kBandSlots; ++j) {
    const int b = a.bands.band_of[grp][j];
    bval[j] = b >= 0;
    bidx[j] = bval[j] ? b : 0;
    bk0[j] = a.bands.k0[bidx[j]];
    bkl[j] = bval[j] ? a.bands.klen[bidx[j]] : 0;
    bwo[j] = a.bands.woff[bidx[j]];
  }
  if (wlds) {
    float* wlw = const_cast<float*>(wl);
    for (int b = 0; b < a.nb; ++b) {
      const int kl = a.bands.klen[b];
      for (int i = t; i < kl; i += T) wlw[a.bands.woff[b] + i] = a.wn[(size_t)b * B + a.bands.k0[b] + i];
    }
  }

  constexpr int M = DEG > 0 ? moment_count(DEG) : 1;
  double acc_m[kBandSlots][M];
  if (DEG > 0) {
#pragma unroll
    for (int j = 0; j < kBandSlots; ++j)
#pragma unroll
      for (int m = 0; m < M; ++m) acc_m[j][m] = 0.0;
  }
  const uint16_t* cube = reinterpret_cast<const uint16_t*>(a.cube);

  for (int64_t tileidx = blockIdx.x; tileidx < a.ntiles; tileidx += gridDim.x) {
    const int64_t pix0 = tileidx * P;
    const int64_t left = a.npix - pix0;
    const int npx = left < P ? (int)left : P;
    const uint16_t* src = cube + pix0 * B;
    const bool pvalid = pl < npx;
    const int nchunk = (npx * B + 7) >> 3;  // 16-byte chunks (8 samples) holding the tile

    // targets of the fused fit: inline-asm loads right after the DMA (see srf_kernel for why)
    float yv[kBandSlots];
    uint32_t mraw = 1u;
    auto load_targets = [&]() {
      if (DEG > 0) {
        const int64_t pc = pvalid ? pix0 + pl : a.npix - 1;
#pragma unroll
        for (int j = 0; j < kBandSlots; ++j) yv[j] = load_f32_async(a.real + bidx[j] * a.real_bs + pc * a.real_ps);
        if (a.mask != nullptr) mraw = load_u8_async(a.mask + pc);
      }
    };
    auto wait_targets = [&]() {
      if (DEG > 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        static_assert(kBandSlots == 2, "pin list below");
        asm volatile("" : "+v"(yv[0]), "+v"(yv[1]), "+v"(mraw));
      }
    };

    if (t < P) flags[t] = 0u;
    if (FAST && npx == P) {
      const char* srcb = reinterpret_cast<const char*>(src);
      for (int c0 = wave * 64; c0 < nchunk; c0 += T) {
        const int c = c0 + lane;
        if (c < nchunk)
          __builtin_amdgcn_global_load_lds((gptr_t)(srcb + (size_t)c * 16), (lptr_t)(smem + (size_t)c0 * 16),
                                           16, 0, kGldsStream);
      }
      load_targets();
      if (OUTV) flush_stage<T>(ostage, a.out, prev_pix0, prev_npx, ops, t);
      __syncthreads();
      wait_targets();
    } else {
      // generic loader: any 2-byte alignment, ragged last tile; the tail of the last chunk is zeroed
      load_targets();
      if (OUTV) flush_stage<T>(ostage, a.out, prev_pix0, prev_npx, ops, t);
      wait_targets();
      const int n = npx * B;
      for (int i = t; i < nchunk * 8; i += T) tile[i] = i < n ? ld_stream(src + i) : (uint16_t)0;
      __syncthreads();
    }
    if (has_nodata) {
      // nodata sweep: a 16-bit lane of (v ^ nodata:nodata) is zero exactly where the sample is nodata
      const uint4* t4 = reinterpret_cast<const uint4*>(smem);
      for (int c0 = t; c0 < nchunk; c0 += T * kScanBatch) {
        uint4 v[kScanBatch];
#pragma unroll
        for (int u = 0; u < kScanBatch; ++u) {
          const int c = c0 + u * T;
          v[u] = t4[c < nchunk ? c : nchunk - 1];
        }
        uint32_t hit = 0u;
#pragma unroll
        for (int u = 0; u < kScanBatch; ++u) {
          const uint32_t w[4] = {v[u].x ^ nd2, v[u].y ^ nd2, v[u].z ^ nd2, v[u].w ^ nd2};
#pragma unroll
          for (int q = 0; q < 4; ++q) hit |= (w[q] - 0x00010001u) & ~w[q] & 0x80008000u;
        }
        if (hit) {  // rare
#pragma unroll
          for (int u = 0; u < kScanBatch; ++u) {
            const int c = c0 + u * T;
            const int e = (c < nchunk ? c : nchunk - 1) * 8;
            const uint32_t w[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              if ((w[q] & 0xffffu) == (a.nodata & 0xffffu) && e + 2 * q < npx * B) flags[(e + 2 * q) / B] = 1u;
              if ((w[q] >> 16) == (a.nodata & 0xffffu) && e + 2 * q + 1 < npx * B) flags[(e + 2 * q + 1) / B] = 1u;
            }
          }
        }
      }
      __syncthreads();
    }

    const bool bad = flags[pl] != 0u;
    const uint16_t* v = tile + pl * B;
#pragma unroll
    for (int j = 0; j < kBandSlots; ++j) {
      float acc = 0.0f;
      const uint16_t* vs = v + bk0[j];
      if (wlds) {
        // 16 taps = 9 aligned dwords (two samples each; one extra because odd sample offsets start in the
        // middle of a dword) realigned per lane with v_alignbyte: 9 ds_read_b32 instead of 16 ds_read_u16.
        // The ninth dword may lie past the row (next pixel / the flag words): its upper half is never used.
        const float4* w4 = reinterpret_cast<const float4*>(wl + bwo[j]);
        const int e0 = pl * B + bk0[j];
        const uint32_t sh = (uint32_t)(e0 & 1) * 2u;
        const uint32_t* d32 = reinterpret_cast<const uint32_t*>(smem) + (e0 >> 1);
        for (int i0 = 0; i0 < bkl[j]; i0 += kTapChunk) {
          float4 ww[kTapChunk / 4];
          uint32_t d[kTapChunk / 2 + 1];
#pragma unroll
          for (int u = 0; u < kTapChunk / 4; ++u) ww[u] = w4[(i0 >> 2) + u];
#pragma unroll
          for (int u = 0; u <= kTapChunk / 2; ++u) d[u] = d32[(i0 >> 1) + u];
#pragma unroll
          for (int u = 0; u < kTapChunk / 4; ++u) {
            const uint32_t ea = __builtin_amdgcn_alignbyte(d[2 * u + 1], d[2 * u], sh);
            const uint32_t eb = __builtin_amdgcn_alignbyte(d[2 * u + 2], d[2 * u + 1], sh);
            acc = fmaf(ww[u].x, (float)(ea & 0xffffu) * scale, acc);
            acc = fmaf(ww[u].y, (float)(ea >> 16) * scale, acc);
            acc = fmaf(ww[u].z, (float)(eb & 0xffffu) * scale, acc);
            acc = fmaf(ww[u].w, (float)(eb >> 16) * scale, acc);
          }
        }
      } else {
        const float* ws = a.wn + (size_t)bidx[j] * B + bk0[j];
        for (int i = 0; i < bkl[j]; ++i) acc = fmaf(ws[i], (float)vs[i] * scale, acc);
      }
      if (bad) acc = __uint_as_float(0x7fc00000u);
      if (bval[j]) {
        if (OUTV) ostage[pl * ops + bidx[j]] = acc;
        else if (pvalid) st_stream(a.out + bidx[j] * a.out_bs + (pix0 + pl) * a.out_ps, acc);
        if (DEG > 0) {
          const float y = yv[j];
          const bool ok = pvalid && mraw != 0u && finite_f32(acc) && finite_f32(y) && acc > a.min_x && y > a.min_y;
          if (ok) {
            const double xd = (double)acc, yd = (double)y;
            acc_m[j][0] += 1.0;
            acc_m[j][2 * DEG + 1] += yd;
            double pw = 1.0;
#pragma unroll
            for (int k = 1; k <= 2 * DEG; ++k) {
              pw *= xd;
              acc_m[j][k] += pw;
              if (k <= DEG) acc_m[j][2 * DEG + 1 + k] += pw * yd;
            }
          }
        }
      }
    }
    prev_pix0 = pix0;
    prev_npx = npx;
    lds_barrier();
  }
  if (OUTV) flush_stage<T>(ostage, a.out, prev_pix0, prev_npx, ops, t);

  if (DEG > 0) {
#pragma unroll
    for (int j = 0; j < kBandSlots; ++j) {
#pragma unroll
      for (int m = 0; m < M; ++m) {
        // bring the accumulator of pixel q to lane q first: the butterfly then adds in exactly the order of
        // the float32 kernel (lane = pixel), so the partial sums are bit-identical to decode + float32 K1
        double sv = __shfl(acc_m[j][m], (lane >> 1) + 32 * (lane & 1), 64);
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) sv += __shfl_xor(sv, off, 64);
        if (bval[j] && lane == 0) a.partials[((size_t)bidx[j] * M + m) * a.slots + blockIdx.x] = sv;
      }
    }
  }
}

// LDS-DMA issued from inline asm: the compiler's waitcnt pass then does not know a DMA is pending and does
// not force vmcnt(0) in front of every ds_read; the wait is the explicit vmcnt(0) at the top of each iteration.
__device__ __forceinline__ void glds16_nt_asm(const void* gaddr, uint32_t lds_base) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off nt" ::"v"(gaddr), "s"(lds_base) : "memory");
}

// Double-buffered form of srf_u16_kernel for 16-byte aligned cubes (the normal case).  A uint16 tile is
// half the bytes of a float32 one, so with one tile per workgroup only ~73 KB per CU were in flight and the
// kernel was latency-bound (0.169 ms).  Here every workgroup owns two tile buffers (2 x 36.5 KB, the LDS
// footprint of the float32 kernel): the DMA of tile k+1 and its fit targets are issued right after the
// barrier that publishes tile k and land while tile k is swept and reduced.  Two barriers per tile; the
// closing barrier of the single-buffer kernel is not needed because nothing of tile k is overwritten before
// the next top barrier.  Same arithmetic, same summation trees -> same bits as srf_u16_kernel.
template <int DEG, bool OUTV>
__global__ __launch_bounds__(512, 4) void srf_u16_ring_kernel(const SrfArgs a) {
  constexpr int P = 64, T = 512;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int B = a.B;
  const int tile_bytes = P * B * 2;   // 128*B: whole 16-byte chunks
  uint32_t* flags = reinterpret_cast<uint32_t*>(smem + 2 * (size_t)tile_bytes);   // [2][64]
  const float* wl = reinterpret_cast<const float*>(flags + 128);
  const bool wlds = a.wtaps > 0;
  float* ostage = const_cast<float*>(wl) + a.wtaps;
  const int ops = (int)a.out_ps;

  const int t = threadIdx.x;
  const int lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int pl = 2 * (lane & 31) + (lane >> 5), grp = wave;   // see srf_u16_kernel
  const float scale = a.scale;
  const uint32_t nd2 = (a.nodata & 0xffffu) * 0x00010001u;
  const bool has_nodata = a.nodata <= 0xffffu;
  const int nchunk_full = tile_bytes >> 4;

  int bk0[kBandSlots], bkl[kBandSlots], bwo[kBandSlots], bidx[kBandSlots];
  bool bval[kBandSlots];
#pragma unroll
  for (int j = 0; j < kBandSlots; ++j) {
    const int b = a.bands.band_of[grp][j];
    bval[j] = b >= 0;
    bidx[j] = bval[j] ? b : 0;
    bk0[j] = a.bands.k0[bidx[j]];
    bkl[j] = bval[j] ? a.bands.klen[bidx[j]] : 0;
    bwo[j] = a.bands.woff[bidx[j]];
  }
  if (wlds) {
    float* wlw = const_cast<float*>(wl);
    for (int b = 0; b < a.nb; ++b) {
      const int kl = a.bands.klen[b];
      for (int i = t; i < kl; i += T) wlw[a.bands.woff[b] + i] = a.wn[(size_t)b * B + a.bands.k0[b] + i];
    }
  }
  constexpr int M = DEG > 0 ? moment_count(DEG) : 1;
  double acc_m[kBandSlots][M];
  if (DEG > 0) {
#pragma unroll
    for (int j = 0; j < kBandSlots; ++j)
#pragma unroll
      for (int m = 0; m < M; ++m) acc_m[j][m] = 0.0;
  }
  const uint16_t* cube = reinterpret_cast<const uint16_t*>(a.cube);
  const uint32_t lds0 = (uint32_t)(uintptr_t)(lptr_t)(smem);

  // prefetch of one tile: LDS-DMA of the samples (full tiles only; the ragged last tile is filled by hand when
  // it is consumed) + the fit targets of this thread's pixel, all invisible to the compiler's waitcnt pass
  float yn[kBandSlots] = {0.0f, 0.0f};
  uint32_t mn = 1u;
  auto prefetch = [&](int64_t tileidx, int buf) {
    const int64_t pix0 = tileidx * P;
    const int64_t left = a.npix - pix0;
    if (left >= P) {
      const char* srcb = reinterpret_cast<const char*>(cube + pix0 * B);
      for (int c0 = wave * 64; c0 < nchunk_full; c0 += T) {
        const int c = c0 + lane;
        if (c < nchunk_full)
          glds16_nt_asm(srcb + (size_t)c * 16, __builtin_amdgcn_readfirstlane(lds0 + (uint32_t)buf * tile_bytes + (uint32_t)c0 * 16));
      }
    }
    if (DEG > 0) {
      const int64_t pc = pl < left ? pix0 + pl : a.npix - 1;
#pragma unroll
      for (int j = 0; j < kBandSlots; ++j) yn[j] = load_f32_async(a.real + bidx[j] * a.real_bs + pc * a.real_ps);
      if (a.mask != nullptr) mn = load_u8_async(a.mask + pc);
    }
  };

  int64_t tileidx = blockIdx.x;
  if (tileidx < a.ntiles) prefetch(tileidx, 0);
  int cur = 0;
  int64_t prev_pix0 = 0;
  int prev_npx = 0;
  for (; tileidx < a.ntiles; tileidx += gridDim.x, cur ^= 1) {
    const int64_t pix0 = tileidx * P;
    const int64_t left = a.npix - pix0;
    const int npx = left < P ? (int)left : P;
    const bool pvalid = pl < npx;
    const int nchunk = (npx * B + 7) >> 3;
    uint16_t* tile = reinterpret_cast<uint16_t*>(smem + (size_t)cur * tile_bytes);
    uint32_t* fl = flags + 64 * cur;

    if (t < P) fl[t] = 0u;
    // everything this wave issued one iteration ago has landed: tile k, its targets, the flush of tile k-2
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float yv[kBandSlots];
    uint32_t mraw;
    static_assert(kBandSlots == 2, "pin list below");
    asm volatile("" : "+v"(yn[0]), "+v"(yn[1]), "+v"(mn));
    yv[0] = yn[0];
    yv[1] = yn[1];
    mraw = mn;
    if (npx < P) {  // ragged last tile: plain copy, tail of the last chunk zeroed
      const uint16_t* src = cube + pix0 * B;
      const int n = npx * B;
      for (int i = t; i < nchunk * 8; i += T) tile[i] = i < n ? src[i] : (uint16_t)0;
    }
    __syncthreads();   // tile k (every wave's share of the DMA) and the staged planes of tile k-1 are visible

    const int64_t nxt = tileidx + gridDim.x;
    if (nxt < a.ntiles) prefetch(nxt, cur ^ 1);   // buffer cur^1 was last read before the barrier above
    if (OUTV) flush_stage<T>(ostage, a.out, prev_pix0, prev_npx, ops, t);

    if (has_nodata) {
      const uint4* t4 = reinterpret_cast<const uint4*>(tile);
      for (int c0 = t; c0 < nchunk; c0 += T * kScanBatch) {
        uint4 v[kScanBatch];
#pragma unroll
        for (int u = 0; u < kScanBatch; ++u) {
          const int c = c0 + u * T;
          v[u] = t4[c < nchunk ? c : nchunk - 1];
        }
        uint32_t hit = 0u;
#pragma unroll
        for (int u = 0; u < kScanBatch; ++u) {
          const uint32_t w[4] = {v[u].x ^ nd2, v[u].y ^ nd2, v[u].z ^ nd2, v[u].w ^ nd2};
#pragma unroll
          for (int q = 0; q < 4; ++q) hit |= (w[q] - 0x00010001u) & ~w[q] & 0x80008000u;
        }
        if (hit) {  // rare
#pragma unroll
          for (int u = 0; u < kScanBatch; ++u) {
            const int c = c0 + u * T;
            const int e = (c < nchunk ? c : nchunk - 1) * 8;
            const uint32_t w[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              if ((w[q] & 0xffffu) == (a.nodata & 0xffffu) && e + 2 * q < npx * B) fl[(e + 2 * q) / B] = 1u;
              if ((w[q] >> 16) == (a.nodata & 0xffffu) && e + 2 * q + 1 < npx * B) fl[(e + 2 * q + 1) / B] = 1u;
            }
          }
        }
      }
    }
    // flags of tile k complete; also orders the flush reads of the staged slab before the writes below
    lds_barrier();

    const bool bad = fl[pl] != 0u;
#pragma unroll
    for (int j = 0; j < kBandSlots; ++j) {
      float acc = 0.0f;
      if (wlds) {
        const float4* w4 = reinterpret_cast<const float4*>(wl + bwo[j]);
        const int e0 = pl * B + bk0[j];
        const uint32_t sh = (uint32_t)(e0 & 1) * 2u;
        const uint32_t* d32 = reinterpret_cast<const uint32_t*>(tile) + (e0 >> 1);
        for (int i0 = 0; i0 < bkl[j]; i0 += kTapChunk) {
          float4 ww[kTapChunk / 4];
          uint32_t d[kTapChunk / 2 + 1];
#pragma unroll
          for (int u = 0; u < kTapChunk / 4; ++u) ww[u] = w4[(i0 >> 2) + u];
#pragma unroll
          for (int u = 0; u <= kTapChunk / 2; ++u) d[u] = d32[(i0 >> 1) + u];
#pragma unroll
          for (int u = 0; u < kTapChunk / 4; ++u) {
            const uint32_t ea = __builtin_amdgcn_alignbyte(d[2 * u + 1], d[2 * u], sh);
            const uint32_t eb = __builtin_amdgcn_alignbyte(d[2 * u + 2], d[2 * u + 1], sh);
            acc = fmaf(ww[u].x, (float)(ea & 0xffffu) * scale, acc);
            acc = fmaf(ww[u].y, (float)(ea >> 16) * scale, acc);
            acc = fmaf(ww[u].z, (float)(eb & 0xffffu) * scale, acc);
            acc = fmaf(ww[u].w, (float)(eb >> 16) * scale, acc);
          }
        }
      } else {
        const uint16_t* vs = tile + pl * B + bk0[j];
        const float* ws = a.wn + (size_t)bidx[j] * B + bk0[j];
        for (int i = 0; i < bkl[j]; ++i) acc = fmaf(ws[i], (float)vs[i] * scale, acc);
      }
      if (bad) acc = __uint_as_float(0x7fc00000u);
      if (bval[j]) {
        if (OUTV) ostage[pl * ops + bidx[j]] = acc;
        else if (pvalid) st_stream(a.out + bidx[j] * a.out_bs + (pix0 + pl) * a.out_ps, acc);
        if (DEG > 0) {
          const float y = yv[j];
          const bool ok = pvalid && mraw != 0u && finite_f32(acc) && finite_f32(y) && acc > a.min_x && y > a.min_y;
          if (ok) {
            const double xd = (double)acc, yd = (double)y;
            acc_m[j][0] += 1.0;
            acc_m[j][2 * DEG + 1] += yd;
            double pw = 1.0;
#pragma unroll
            for (int k = 1; k <= 2 * DEG; ++k) {
              pw *= xd;
              acc_m[j][k] += pw;
              if (k <= DEG) acc_m[j][2 * DEG + 1 + k] += pw * yd;
            }
          }
        }
      }
    }
    prev_pix0 = pix0;
    prev_npx = npx;
  }
  if (OUTV) {
    __syncthreads();
    flush_stage<T>(ostage, a.out, prev_pix0, prev_npx, ops, t);
  }

  if (DEG > 0) {
#pragma unroll
    for (int j = 0; j < kBandSlots; ++j) {
#pragma unroll
      for (int m = 0; m < M; ++m) {
        double sv = __shfl(acc_m[j][m], (lane >> 1) + 32 * (lane & 1), 64);
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) sv += __shfl_xor(sv, off, 64);
        if (bval[j] && lane == 0) a.partials[((size_t)bidx[j] * M + m) * a.slots + blockIdx.x] = sv;
      }
    }
  }
}

template <int DEG, bool OUTV>
static int launch_srf_u16_ring(const SrfArgs& a, hipStream_t stream) {
  const size_t lds = (size_t)2 * 64 * a.B * 2 + 128 * sizeof(uint32_t) + (size_t)a.wtaps * 4 + (OUTV ? (size_t)64 * a.out_ps * 4 : 0);
  auto kern = srf_u16_ring_kernel<DEG, OUTV>;
  static thread_local size_t configured = 0;
  if (lds > configured) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipGetLastError();
    configured = lds;
  }
  hipLaunchKernelGGL(kern, dim3(a.slots), dim3(512), lds, stream, a);
  HSR_LAUNCH_CHECK("srf_u16_ring_kernel");
  return HSR_OK;
}

template <int DEG, bool FAST, bool OUTV>
static int launch_srf_u16(const SrfArgs& a, hipStream_t stream) {
  const size_t lds = (size_t)64 * a.B * 2 + 64 * sizeof(uint32_t) + (size_t)a.wtaps * 4 + (OUTV ? (size_t)64 * a.out_ps * 4 : 0);
  auto kern = srf_u16_kernel<DEG, FAST, OUTV>;
  static thread_local size_t configured = 0;
  if (lds > configured) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipGetLastError();
    configured = lds;
  }
  hipLaunchKernelGGL(kern, dim3(a.slots), dim3(512), lds, stream, a);
  HSR_LAUNCH_CHECK("srf_u16_kernel");
  return HSR_OK;
}

template <int DEG>
static int dispatch_u16_deg(const SrfArgs& a, bool fast, hipStream_t s) {
  const bool outv = a.out_bs == 1 && (a.out_ps & 3) == 0 && a.out_ps <= HSR_MAX_BANDS && (((uintptr_t)a.out) & 15) == 0;
  // two tile buffers must fit twice per CU next to the weights and the output slab: B <= ~300 spectral samples
  const size_t ring_lds = (size_t)2 * 64 * a.B * 2 + 512 + (size_t)a.wtaps * 4 + (outv ? (size_t)64 * a.out_ps * 4 : 0);
  if (fast && g_u16_ring && ring_lds <= 80 * 1024)
    return outv ? launch_srf_u16_ring<DEG, true>(a, s) : launch_srf_u16_ring<DEG, false>(a, s);
  if (outv) return fast ? launch_srf_u16<DEG, true, true>(a, s) : launch_srf_u16<DEG, false, true>(a, s);
  return fast ? launch_srf_u16<DEG, true, false>(a, s) : launch_srf_u16<DEG, false, false>(a, s);
}

static int dispatch_u16(const SrfArgs& a, int deg, bool fast, hipStream_t s) {
  switch (deg) {
    case 0: return dispatch_u16_deg<0>(a, fast, s);
    case 1: return dispatch_u16_deg<1>(a, fast, s);
    case 2: return dispatch_u16_deg<2>(a, fast, s);
    case 3: return dispatch_u16_deg<3>(a, fast, s);
    case 4: return dispatch_u16_deg<4>(a, fast, s);
  }
  set_error("hsr_srf_integrate_moments_u16: deg=%d outside [1,%d]", deg, HSR_MAX_DEG);
  return HSR_ERR_UNSUPPORTED;
}

template <int DEG, bool FAST, bool WLDS, int P, bool OUTV>
static int launch_srf(const SrfArgs& a, hipStream_t stream) {
  const size_t lds = (size_t)P * a.ldsB * 4 + 64 * sizeof(uint32_t) + (WLDS ? (size_t)a.wtaps * 4 : 0) +
                     (OUTV ? (size_t)P * a.out_ps * 4 : 0);
  auto kern = srf_kernel<DEG, FAST, WLDS, P, OUTV>;
  static thread_local size_t configured = 0;
  if (lds > configured) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipGetLastError();
    configured = lds;
  }
#ifdef HSR_PHASE_STAMPS
  const_cast<SrfArgs&>(a).stamps = g_stamp_buffer;
#endif
  hipLaunchKernelGGL(kern, dim3(a.slots), dim3(8 * P), lds, stream, a);
  HSR_LAUNCH_CHECK("srf_kernel");
  return HSR_OK;
}

template <int DEG, int P>
static int dispatch_fast(const SrfArgs& a, bool fast, hipStream_t s) {
  // pixel-major output with a 16-byte friendly row: stage the slab in LDS and flush it vectorised
  const bool outv = a.out_bs == 1 && (a.out_ps & 3) == 0 && a.out_ps <= HSR_MAX_BANDS &&
                    (((uintptr_t)a.out) & 15) == 0;
  if (a.wtaps == 0) return launch_srf<DEG, false, false, 64, false>(a, s);  // rare fallback: one generic kernel
  if (outv) return fast ? launch_srf<DEG, true, true, P, true>(a, s) : launch_srf<DEG, false, true, P, true>(a, s);
  return fast ? launch_srf<DEG, true, true, P, false>(a, s) : launch_srf<DEG, false, true, P, false>(a, s);
}

template <int P>
static int dispatch_deg(const SrfArgs& a, int deg, bool fast, hipStream_t s) {
  switch (deg) {
    case 0: return dispatch_fast<0, P>(a, fast, s);
    case 1: return dispatch_fast<1, P>(a, fast, s);
    case 2: return dispatch_fast<2, P>(a, fast, s);
    case 3: return dispatch_fast<3, P>(a, fast, s);
    case 4: return dispatch_fast<4, P>(a, fast, s);
  }
  set_error("hsr_srf_integrate_moments: deg=%d outside [1,%d]", deg, HSR_MAX_DEG);
  return HSR_ERR_UNSUPPORTED;
}

int srf_partial_slots(int64_t npix) {
  const int P = g_tile_pixels;
  int64_t tiles = (npix + P - 1) / P;
  if (tiles < 1) tiles = 1;
  const int64_t cap = (int64_t)(256 - g_reserved_cus) * (P == 64 ? 2 : 4);  // CUs x resident workgroups
  return (int)(tiles < cap ? tiles : cap);
}

static int srf_common(SrfArgs& a, const int32_t* k0, const int32_t* klen, int32_t deg, hipStream_t stream) {
  HSR_REQUIRE(a.cube && a.wn && a.out && k0 && klen, HSR_ERR_INVALID, "hsr_srf_integrate: NULL pointer");
  HSR_REQUIRE(a.npix >= 0, HSR_ERR_INVALID, "hsr_srf_integrate: npix < 0");
  HSR_REQUIRE(a.B >= 1 && a.B <= HSR_MAX_SPECTRAL, HSR_ERR_UNSUPPORTED,
              "hsr_srf_integrate: B=%d outside [1,%d]", a.B, HSR_MAX_SPECTRAL);
  HSR_REQUIRE(a.nb >= 1 && a.nb <= HSR_MAX_BANDS, HSR_ERR_UNSUPPORTED,
              "hsr_srf_integrate: nb=%d outside [1,%d]", a.nb, HSR_MAX_BANDS);
  HSR_REQUIRE((a.out_ps == 1 && a.out_bs >= a.npix) || (a.out_bs == 1 && a.out_ps >= a.nb), HSR_ERR_INVALID,
              "hsr_srf_integrate: output strides (%lld, %lld) are neither band-major nor pixel-major",
              (long long)a.out_bs, (long long)a.out_ps);
  HSR_REQUIRE(((uintptr_t)a.cube & (a.u16 ? 1 : 3)) == 0, HSR_ERR_INVALID, "hsr_srf_integrate: cube not %d-byte aligned",
              a.u16 ? 2 : 4);
  for (int b = 0; b < a.nb; ++b) {
    HSR_REQUIRE(k0[b] >= 0 && klen[b] >= 0 && k0[b] + klen[b] <= a.B, HSR_ERR_INVALID,
                "hsr_srf_integrate: support of band %d = [%d,%d) outside [0,%d)", b, k0[b], k0[b] + klen[b], a.B);
    a.bands.k0[b] = k0[b];
    a.bands.klen[b] = klen[b];
    a.bands.woff[b] = 0;
  }
  for (int b = a.nb; b < HSR_MAX_BANDS; ++b) a.bands.k0[b] = a.bands.klen[b] = a.bands.woff[b] = 0;
  // LDS weight segments: widen every support to whole 16-tap chunks that stay inside [0, B) (the
  // added taps have weight 0 in the dense rows of wn); if that is impossible (B < 16 * chunks) or the
  // taps do not fit the reserved LDS, the kernel reads the weights from global memory instead.
  {
    int32_t sk0[HSR_MAX_BANDS], skl[HSR_MAX_BANDS], off[HSR_MAX_BANDS], total = 0;
    bool fits = true;
    for (int b = 0; b < a.nb && fits; ++b) {
      const int nc = (klen[b] + kTapChunk - 1) / kTapChunk;
      int s0 = k0[b];
      if (s0 + kTapChunk * nc > a.B) s0 = a.B - kTapChunk * nc;
      if (s0 < 0) fits = false;
      sk0[b] = s0;
      skl[b] = kTapChunk * nc;
      off[b] = total;
      total += kTapChunk * nc;
    }
    if (fits && total <= kWeightCap) {
      for (int b = 0; b < a.nb; ++b) {
        a.bands.k0[b] = sk0[b];
        a.bands.klen[b] = skl[b];
        a.bands.woff[b] = off[b];
      }
      a.wtaps = total > 0 ? total : kTapChunk;
    } else {
      a.wtaps = 0;
    }
  }
  // Band -> (group, slot): the slowest of the 8 groups sets the length of the dot-product phase, so deal the
  // bands out longest first to the least loaded group (2 slots each).  Any assignment gives identical bits.
  {
    int order[HSR_MAX_BANDS], load[8] = {0, 0, 0, 0, 0, 0, 0, 0}, used[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int g = 0; g < 8; ++g) a.bands.band_of[g][0] = a.bands.band_of[g][1] = -1;
    for (int b = 0; b < a.nb; ++b) order[b] = b;
    for (int i = 1; i < a.nb; ++i)   // insertion sort by descending tap count (stable)
      for (int k = i; k > 0 && a.bands.klen[order[k]] > a.bands.klen[order[k - 1]]; --k) {
        const int t = order[k]; order[k] = order[k - 1]; order[k - 1] = t;
      }
    for (int i = 0; i < a.nb; ++i) {
      int best = -1;
      for (int g = 0; g < 8; ++g)
        if (used[g] < 2 && (best < 0 || load[g] < load[best])) best = g;
      a.bands.band_of[best][used[best]++] = (int8_t)order[i];
      load[best] += a.bands.klen[order[i]] + 1;
    }
  }
  if (a.npix == 0) return HSR_OK;
  if (a.u16) {  // 64-pixel tiles, two resident workgroups per CU
    a.ntiles = (a.npix + 63) / 64;
    const int64_t cap = (int64_t)(256 - g_reserved_cus) * 2;
    a.slots = (int)(a.ntiles < cap ? a.ntiles : cap);
    return dispatch_u16(a, deg, (((uintptr_t)a.cube & 15) == 0), stream);
  }
  int P = g_tile_pixels;
  if (a.wtaps == 0) P = 64;  // the generic fallback kernel exists for 64-pixel tiles only
  a.ntiles = (a.npix + P - 1) / P;
  a.slots = srf_partial_slots(a.npix);
  if (a.wtaps == 0) { int64_t tl = a.ntiles; a.slots = (int)(tl < 512 ? tl : 512); }
  a.ldsB = (a.B & 1) ? a.B : a.B + 1;
  const bool fast = (a.B & 1) && (((uintptr_t)a.cube & 15) == 0);
  return P == 64 ? dispatch_deg<64>(a, deg, fast, stream) : dispatch_deg<32>(a, deg, fast, stream);
}

}  // namespace hsr

extern "C" int hsr_set_srf_tile(int32_t pixels) {
  HSR_REQUIRE(pixels == 64 || pixels == 32, HSR_ERR_INVALID, "hsr_set_srf_tile: pixels must be 64 or 32, got %d", pixels);
  hsr::g_tile_pixels = pixels;
  return HSR_OK;
}

extern "C" int hsr_get_srf_tile(void) { return hsr::g_tile_pixels; }

extern "C" int hsr_set_srf_u16_ring(int32_t on) {
  hsr::g_u16_ring = on != 0;
  return HSR_OK;
}

extern "C" int hsr_set_srf_reserved_cus(int32_t cus) {
  HSR_REQUIRE(cus >= 0 && cus <= 128, HSR_ERR_INVALID, "hsr_set_srf_reserved_cus: %d outside [0,128]", cus);
  hsr::g_reserved_cus = cus;
  return HSR_OK;
}

extern "C" int hsr_srf_integrate(const float* cube_dev, int64_t npix, int32_t B, const float* wn_dev,
                                 const int32_t* k0, const int32_t* klen, int32_t nb, float* out_dev,
                                 int64_t out_bs, int64_t out_ps, hsr_stream_t stream) {
  hsr::SrfArgs a{};
  a.cube = cube_dev;
  a.npix = npix;
  a.B = B;
  a.wn = wn_dev;
  a.nb = nb;
  a.out = out_dev;
  a.out_bs = out_bs;
  a.out_ps = out_ps;
  return hsr::srf_common(a, k0, klen, 0, (hipStream_t)stream);
}

extern "C" int hsr_srf_integrate_moments(const float* cube_dev, int64_t npix, int32_t B, const float* wn_dev,
                                         const int32_t* k0, const int32_t* klen, int32_t nb, float* out_dev,
                                         int64_t out_bs, int64_t out_ps, const float* real_dev, int64_t real_bs,
                                         int64_t real_ps, const uint8_t* mask_dev, float min_x, float min_y, int32_t deg,
                                         double* partials_dev, int32_t* slots_out, hsr_stream_t stream) {
  HSR_REQUIRE(deg >= 1 && deg <= HSR_MAX_DEG, HSR_ERR_UNSUPPORTED, "hsr_srf_integrate_moments: deg=%d outside [1,%d]",
              deg, HSR_MAX_DEG);
  HSR_REQUIRE(real_dev && partials_dev, HSR_ERR_INVALID, "hsr_srf_integrate_moments: NULL pointer");
  HSR_REQUIRE((real_ps == 1 && real_bs >= npix) || (real_bs == 1 && real_ps >= nb), HSR_ERR_INVALID,
              "hsr_srf_integrate_moments: real strides (%lld, %lld) are neither band-major nor pixel-major",
              (long long)real_bs, (long long)real_ps);
  HSR_REQUIRE(npix > 0, HSR_ERR_INVALID, "hsr_srf_integrate_moments: npix must be > 0");
  hsr::SrfArgs a{};
  a.cube = cube_dev;
  a.npix = npix;
  a.B = B;
  a.wn = wn_dev;
  a.nb = nb;
  a.out = out_dev;
  a.out_bs = out_bs;
  a.out_ps = out_ps;
  a.real = real_dev;
  a.real_bs = real_bs;
  a.real_ps = real_ps;
  a.mask = mask_dev;
  a.min_x = min_x;
  a.min_y = min_y;
  a.partials = partials_dev;
  int rc = hsr::srf_common(a, k0, klen, deg, (hipStream_t)stream);
  if (rc == HSR_OK && slots_out) *slots_out = a.slots;
  return rc;
}

extern "C" int hsr_srf_integrate_u16(const uint16_t* cube_dev, int64_t npix, int32_t B, float scale, int32_t nodata,
                                     const float* wn_dev, const int32_t* k0, const int32_t* klen, int32_t nb,
                                     float* out_dev, int64_t out_bs, int64_t out_ps, hsr_stream_t stream) {
  HSR_REQUIRE(nodata <= 0xffff, HSR_ERR_INVALID, "hsr_srf_integrate_u16: nodata=%d is not a uint16 value (negative = none)", nodata);
  hsr::SrfArgs a{};
  a.cube = reinterpret_cast<const float*>(cube_dev);
  a.u16 = 1;
  a.scale = scale;
  a.nodata = nodata < 0 ? 0x10000u : (uint32_t)nodata;
  a.npix = npix;
  a.B = B;
  a.wn = wn_dev;
  a.nb = nb;
  a.out = out_dev;
  a.out_bs = out_bs;
  a.out_ps = out_ps;
  return hsr::srf_common(a, k0, klen, 0, (hipStream_t)stream);
}

extern "C" int hsr_srf_integrate_moments_u16(const uint16_t* cube_dev, int64_t npix, int32_t B, float scale,
                                             int32_t nodata, const float* wn_dev, const int32_t* k0,
                                             const int32_t* klen, int32_t nb, float* out_dev, int64_t out_bs,
                                             int64_t out_ps, const float* real_dev, int64_t real_bs, int64_t real_ps,
                                             const uint8_t* mask_dev, float min_x, float min_y, int32_t deg,
                                             double* partials_dev, int32_t* slots_out, hsr_stream_t stream) {
  HSR_REQUIRE(deg >= 1 && deg <= HSR_MAX_DEG, HSR_ERR_UNSUPPORTED, "hsr_srf_integrate_moments_u16: deg=%d outside [1,%d]",
              deg, HSR_MAX_DEG);
  HSR_REQUIRE(real_dev && partials_dev, HSR_ERR_INVALID, "hsr_srf_integrate_moments_u16: NULL pointer");
  HSR_REQUIRE((real_ps == 1 && real_bs >= npix) || (real_bs == 1 && real_ps >= nb), HSR_ERR_INVALID,
              "hsr_srf_integrate_moments_u16: real strides (%lld, %lld) are neither band-major nor pixel-major",
              (long long)real_bs, (long long)real_ps);
  HSR_REQUIRE(npix > 0, HSR_ERR_INVALID, "hsr_srf_integrate_moments_u16: npix must be > 0");
  HSR_REQUIRE(nodata <= 0xffff, HSR_ERR_INVALID, "hsr_srf_integrate_moments_u16: nodata=%d is not a uint16 value (negative = none)", nodata);
  hsr::SrfArgs a{};
  a.cube = reinterpret_cast<const float*>(cube_dev);
  a.u16 = 1;
  a.scale = scale;
  a.nodata = nodata < 0 ? 0x10000u : (uint32_t)nodata;
  a.npix = npix;
  a.B = B;
  a.wn = wn_dev;
  a.nb = nb;
  a.out = out_dev;
  a.out_bs = out_bs;
  a.out_ps = out_ps;
  a.real = real_dev;
  a.real_bs = real_bs;
  a.real_ps = real_ps;
  a.mask = mask_dev;
  a.min_x = min_x;
  a.min_y = min_y;
  a.partials = partials_dev;
  int rc = hsr::srf_common(a, k0, klen, deg, (hipStream_t)stream);
  if (rc == HSR_OK && slots_out) *slots_out = a.slots;
  return rc;
}
