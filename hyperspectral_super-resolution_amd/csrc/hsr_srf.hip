// K1 (+ fused K2): SRF band integration of an (npix, B) float32 cube on gfx950.
//
// Replaces the hot loop of pseudo_s2_srf_integral (reference s2_emit/synth.py:32-43).  The reference
// makes 13 full-cube float64 passes; here the cube is streamed from HBM exactly once.
//
// Data flow per workgroup (256 threads = 4 waves, 2 workgroups resident per CU):
//   1. a tile of 64 consecutive pixels (64*B*4 bytes, one linear 16-byte-aligned slab because the
//      cube is pixel-major) goes HBM -> LDS with global_load_lds_dwordx4 (no VGPR round trip,
//      1 KiB per wave instruction, fully coalesced).  ~73 KB in flight per workgroup.
//   2. one linear ds_read_b128 sweep flags pixels holding a non-finite sample.
//   3. lane = pixel, wave = band group: each band is a short dot product over its SRF support read
//      from LDS with a row stride of B words (B odd -> bank-conflict free); the weights are
//      wave-uniform and arrive through the scalar cache (s_load) as SGPR operands of v_fmac.
//      Flagged pixels take the dense product so that 0*Inf -> NaN poisons exactly the bands the
//      reference poisons (synth.py:41 multiplies all B samples of every band).
//   4. planes[b][pixel] is stored coalesced (256 B per wave store).  With DEG > 0 the same lane
//      also accumulates the Vandermonde power sums of (x = plane value, y = real S2 value) in
//      float64 registers; they are reduced over the wave by a fixed butterfly and written to a
//      per-workgroup slot (no float atomics -> bitwise reproducible).
// HBM-bound by construction: 4*B bytes in, 4*nb (+4*nb+1) bytes out/in per pixel, ~0.35 kflop.
#include "hsr_common.h"

namespace hsr {

struct SrfBands {
  int32_t k0[HSR_MAX_BANDS];
  int32_t klen[HSR_MAX_BANDS];
};

struct SrfArgs {
  const float* cube;
  int64_t npix;
  int64_t ntiles;
  int32_t B;
  int32_t ldsB;  // LDS row stride in words (odd)
  const float* wn;
  SrfBands bands;
  int32_t nb;
  float* planes;
  int64_t plane_stride;
  const float* real;
  int64_t real_stride;
  const uint8_t* mask;
  float min_x, min_y;
  double* partials;
  int32_t slots;
};

constexpr int kBlock = 256;
constexpr int kBandSlots = HSR_MAX_BANDS / 4;  // bands per wave

typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// x*0 is NaN exactly when x is NaN or +-Inf; four of them chained cost 4 VALU ops.
__device__ __forceinline__ bool any_nonfinite4(const float4& v) {
  float z = v.x * 0.0f;
  z = fmaf(v.y, 0.0f, z);
  z = fmaf(v.z, 0.0f, z);
  z = fmaf(v.w, 0.0f, z);
  return z != z;
}

template <int DEG, bool FAST>
__global__ __launch_bounds__(kBlock, 2) void srf_kernel(const SrfArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* tile = reinterpret_cast<float*>(smem);
  const int B = a.B;
  const int ldsB = a.ldsB;
  uint32_t* flags = reinterpret_cast<uint32_t*>(smem + (size_t)HSR_TILE_PIXELS * ldsB * 4);

  const int t = threadIdx.x;
  const int lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int nchunk = 16 * B;  // 16-byte chunks of a full tile

  constexpr int M = DEG > 0 ? moment_count(DEG) : 1;
  double acc_m[kBandSlots][M];
  if (DEG > 0) {
#pragma unroll
    for (int j = 0; j < kBandSlots; ++j)
#pragma unroll
      for (int m = 0; m < M; ++m) acc_m[j][m] = 0.0;
  }

  for (int64_t tileidx = blockIdx.x; tileidx < a.ntiles; tileidx += gridDim.x) {
    const int64_t pix0 = tileidx * HSR_TILE_PIXELS;
    const int64_t left = a.npix - pix0;
    const int npx = left < HSR_TILE_PIXELS ? (int)left : HSR_TILE_PIXELS;
    const float* src = a.cube + pix0 * B;
    const bool pvalid = lane < npx;

    // operands of the fused fit: issue these loads before waiting for the tile
    float yv[kBandSlots];
    bool mv = true;
    if (DEG > 0) {
#pragma unroll
      for (int j = 0; j < kBandSlots; ++j) {
        const int b = wave + 4 * j;
        yv[j] = (b < a.nb && pvalid) ? a.real[b * a.real_stride + pix0 + lane] : 0.0f;
      }
      if (a.mask != nullptr) mv = pvalid && a.mask[pix0 + lane] != 0;
    }

    if (t < HSR_TILE_PIXELS) flags[t] = 0u;

    const bool fast_tile = FAST && npx == HSR_TILE_PIXELS;
    if (fast_tile) {
      const char* srcb = reinterpret_cast<const char*>(src);
      for (int c0 = wave * 64; c0 < nchunk; c0 += kBlock) {  // c0 is wave-uniform
        const int c = c0 + lane;
        if (c < nchunk)
          __builtin_amdgcn_global_load_lds((gptr_t)(srcb + (size_t)c * 16), (lptr_t)(smem + (size_t)c0 * 16),
                                           16, 0, 0);
      }
      __syncthreads();
      const float4* t4 = reinterpret_cast<const float4*>(smem);
      for (int c = t; c < nchunk; c += kBlock) {
        const float4 v = t4[c];
        if (any_nonfinite4(v)) {  // rare
          const int e = c * 4;
          if (!finite_f32(v.x)) flags[(e + 0) / B] = 1u;
          if (!finite_f32(v.y)) flags[(e + 1) / B] = 1u;
          if (!finite_f32(v.z)) flags[(e + 2) / B] = 1u;
          if (!finite_f32(v.w)) flags[(e + 3) / B] = 1u;
        }
      }
    } else {
      // generic loader: any 4-byte alignment, any B, ragged last tile.  One pixel row per wave step.
      __syncthreads();  // flags zeroed before anybody sets one
      for (int pp = wave; pp < npx; pp += 4) {
        bool bad = false;
        for (int k = lane; k < B; k += 64) {
          const float v = src[(size_t)pp * B + k];
          bad |= !finite_f32(v);
          tile[pp * ldsB + k] = v;
        }
        if (bad) flags[pp] = 1u;
      }
    }
    __syncthreads();

    const bool slow = flags[lane] != 0u;
    const float* v = tile + lane * ldsB;
#pragma unroll
    for (int j = 0; j < kBandSlots; ++j) {
      const int b = wave + 4 * j;  // wave-uniform
      if (b < a.nb) {
        const float* w = a.wn + (size_t)b * B;
        float acc = 0.0f;
        if (!slow) {
          const int k0 = a.bands.k0[b];
          const int kl = a.bands.klen[b];
          const float* ws = w + k0;
          const float* vs = v + k0;
          for (int i = 0; i < kl; ++i) acc = fmaf(ws[i], vs[i], acc);
        } else {
          for (int k = 0; k < B; ++k) acc = fmaf(w[k], v[k], acc);
        }
        if (pvalid) a.planes[b * a.plane_stride + pix0 + lane] = acc;
        if (DEG > 0) {
          const float y = yv[j];
          const bool ok = pvalid && mv && finite_f32(acc) && finite_f32(y) && acc > a.min_x && y > a.min_y;
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
    __syncthreads();  // tile and flags are rewritten by the next iteration
  }

  if (DEG > 0) {
#pragma unroll
    for (int j = 0; j < kBandSlots; ++j) {
      const int b = wave + 4 * j;
      if (b < a.nb) {
#pragma unroll
        for (int m = 0; m < M; ++m) {
          const double s = wave_sum(acc_m[j][m]);
          if (lane == 0) a.partials[((size_t)b * M + m) * a.slots + blockIdx.x] = s;
        }
      }
    }
  }
}

template <int DEG, bool FAST>
static int launch_srf(const SrfArgs& a, hipStream_t stream) {
  const size_t lds = (size_t)HSR_TILE_PIXELS * a.ldsB * 4 + HSR_TILE_PIXELS * sizeof(uint32_t);
  auto kern = srf_kernel<DEG, FAST>;
  static thread_local size_t configured = 0;
  if (lds > configured) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipGetLastError();
    configured = lds;
  }
  hipLaunchKernelGGL(kern, dim3(a.slots), dim3(kBlock), lds, stream, a);
  HSR_LAUNCH_CHECK("srf_kernel");
  return HSR_OK;
}

template <int DEG>
static int dispatch_fast(const SrfArgs& a, bool fast, hipStream_t s) {
  return fast ? launch_srf<DEG, true>(a, s) : launch_srf<DEG, false>(a, s);
}

static int srf_common(SrfArgs& a, const int32_t* k0, const int32_t* klen, int32_t deg, hipStream_t stream) {
  HSR_REQUIRE(a.cube && a.wn && a.planes && k0 && klen, HSR_ERR_INVALID, "hsr_srf_integrate: NULL pointer");
  HSR_REQUIRE(a.npix >= 0, HSR_ERR_INVALID, "hsr_srf_integrate: npix < 0");
  HSR_REQUIRE(a.B >= 1 && a.B <= HSR_MAX_SPECTRAL, HSR_ERR_UNSUPPORTED,
              "hsr_srf_integrate: B=%d outside [1,%d]", a.B, HSR_MAX_SPECTRAL);
  HSR_REQUIRE(a.nb >= 1 && a.nb <= HSR_MAX_BANDS, HSR_ERR_UNSUPPORTED,
              "hsr_srf_integrate: nb=%d outside [1,%d]", a.nb, HSR_MAX_BANDS);
  HSR_REQUIRE(a.plane_stride >= a.npix, HSR_ERR_INVALID, "hsr_srf_integrate: plane_stride < npix");
  HSR_REQUIRE(((uintptr_t)a.cube & 3) == 0, HSR_ERR_INVALID, "hsr_srf_integrate: cube not 4-byte aligned");
  for (int b = 0; b < a.nb; ++b) {
    HSR_REQUIRE(k0[b] >= 0 && klen[b] >= 0 && k0[b] + klen[b] <= a.B, HSR_ERR_INVALID,
                "hsr_srf_integrate: support of band %d = [%d,%d) outside [0,%d)", b, k0[b], k0[b] + klen[b], a.B);
    a.bands.k0[b] = k0[b];
    a.bands.klen[b] = klen[b];
  }
  for (int b = a.nb; b < HSR_MAX_BANDS; ++b) a.bands.k0[b] = a.bands.klen[b] = 0;
  if (a.npix == 0) return HSR_OK;
  a.ntiles = (a.npix + HSR_TILE_PIXELS - 1) / HSR_TILE_PIXELS;
  a.slots = partial_slots(a.npix);
  a.ldsB = (a.B & 1) ? a.B : a.B + 1;
  const bool fast = (a.B & 1) && (((uintptr_t)a.cube & 15) == 0);
  switch (deg) {
    case 0: return dispatch_fast<0>(a, fast, stream);
    case 1: return dispatch_fast<1>(a, fast, stream);
    case 2: return dispatch_fast<2>(a, fast, stream);
    case 3: return dispatch_fast<3>(a, fast, stream);
    case 4: return dispatch_fast<4>(a, fast, stream);
  }
  set_error("hsr_srf_integrate_moments: deg=%d outside [1,%d]", deg, HSR_MAX_DEG);
  return HSR_ERR_UNSUPPORTED;
}

}  // namespace hsr

extern "C" int hsr_srf_integrate(const float* cube_dev, int64_t npix, int32_t B, const float* wn_dev,
                                 const int32_t* k0, const int32_t* klen, int32_t nb, float* planes_dev,
                                 int64_t plane_stride, hsr_stream_t stream) {
  hsr::SrfArgs a{};
  a.cube = cube_dev;
  a.npix = npix;
  a.B = B;
  a.wn = wn_dev;
  a.nb = nb;
  a.planes = planes_dev;
  a.plane_stride = plane_stride;
  return hsr::srf_common(a, k0, klen, 0, (hipStream_t)stream);
}

extern "C" int hsr_srf_integrate_moments(const float* cube_dev, int64_t npix, int32_t B, const float* wn_dev,
                                         const int32_t* k0, const int32_t* klen, int32_t nb, float* planes_dev,
                                         int64_t plane_stride, const float* real_dev, int64_t real_stride,
                                         const uint8_t* mask_dev, float min_x, float min_y, int32_t deg,
                                         double* partials_dev, int32_t* slots_out, hsr_stream_t stream) {
  HSR_REQUIRE(deg >= 1 && deg <= HSR_MAX_DEG, HSR_ERR_UNSUPPORTED, "hsr_srf_integrate_moments: deg=%d outside [1,%d]",
              deg, HSR_MAX_DEG);
  HSR_REQUIRE(real_dev && partials_dev, HSR_ERR_INVALID, "hsr_srf_integrate_moments: NULL pointer");
  HSR_REQUIRE(real_stride >= npix, HSR_ERR_INVALID, "hsr_srf_integrate_moments: real_stride < npix");
  HSR_REQUIRE(npix > 0, HSR_ERR_INVALID, "hsr_srf_integrate_moments: npix must be > 0");
  hsr::SrfArgs a{};
  a.cube = cube_dev;
  a.npix = npix;
  a.B = B;
  a.wn = wn_dev;
  a.nb = nb;
  a.planes = planes_dev;
  a.plane_stride = plane_stride;
  a.real = real_dev;
  a.real_stride = real_stride;
  a.mask = mask_dev;
  a.min_x = min_x;
  a.min_y = min_y;
  a.partials = partials_dev;
  int rc = hsr::srf_common(a, k0, klen, deg, (hipStream_t)stream);
  if (rc == HSR_OK && slots_out) *slots_out = a.slots;
  return rc;
}
