// f1: grid-aligned resamplers between the phases of the pipeline (reference: downsample_s2_to_grid /
// reproject_stack_to_grid, notebook-only, Pairs_EMIT_S2_demo-2.ipynb cell 73 raw lines 4538-4599, which
// call rasterio/GDAL `reproject`).  The repository warps EMIT onto the S2 UTM grid at exactly 6 x 10 m
// (EMIT_data/emit_proj.py:791-797) and cuts tiles as exact 6x windows (tiles_helpers/utils.py:256-277), so
// for the pairs this path sees the two warps reduce to an f x f block mean and a pixel-centre-aligned
// separable bilinear upsampling.  GDAL's edge / nodata conventions are PARITY UNPINNED (rasterio absent).
// Both kernels are trivially HBM-bound streams; images use (band_stride, pixel_stride) addressing.
#include "hsr_common.h"
#include "hsr_select_dev.h"

namespace hsr {

template <typename T>
__global__ __launch_bounds__(256) void block_mean_kernel(const T* __restrict__ in, int64_t in_bs, int64_t in_ps,
                                                         int Hc, int Wc, int f, float scale, int nb,
                                                         float* __restrict__ out, int64_t out_bs, int64_t out_ps) {
  const int64_t total = (int64_t)Hc * Wc;
  const int b = blockIdx.y;
  const int64_t Wf = (int64_t)Wc * f;
  for (int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x; p < total; p += (int64_t)gridDim.x * 256) {
    const int y = (int)(p / Wc), x = (int)(p - (int64_t)y * Wc);
    const T* src = in + (size_t)b * in_bs + ((int64_t)y * f * Wf + (int64_t)x * f) * in_ps;
    double s = 0.0;
    for (int dy = 0; dy < f; ++dy)
      for (int dx = 0; dx < f; ++dx) s += (double)src[((int64_t)dy * Wf + dx) * in_ps];
    const float m = (float)(s / (double)(f * f));     // GDAL 'average' accumulates in double, stores float32
    out[(size_t)b * out_bs + p * out_ps] = m * scale;  // `out *= float(src_scale)` on the float32 array
  }
}

// Staged form for the two layouts the pipeline uses - band-major planes (in_ps == 1) and tightly packed band-last
// rows (in_bs == 1, in_ps == nb, e.g. the uint8 RGB of the S2 visual product): a workgroup copies the f fine rows
// under 64 coarse pixels into LDS with 16-byte loads (rows and segments start 16-byte aligned, checked on the host)
// and every thread then adds its f x f window from there in the same (dy, dx) order as block_mean_kernel - same
// bits.  The direct kernel fetched every sample with its own 1- or 4-byte global load (1.5 / 3.4 TB/s).
constexpr int kBmCols = 64;   // coarse pixels per workgroup

template <typename T>
__global__ __launch_bounds__(256) void block_mean_tile_kernel(const T* __restrict__ in, int64_t plane_stride,
                                                              int interleave, int Hc, int Wc, int f, float scale,
                                                              float* __restrict__ out, int64_t out_bs, int64_t out_ps) {
  extern __shared__ __attribute__((aligned(16))) unsigned char bm_smem[];
  T* tile = reinterpret_cast<T*>(bm_smem);
  const int cx0 = blockIdx.x * kBmCols, y = blockIdx.y, plane = blockIdx.z;
  const int ncx = Wc - cx0 < kBmCols ? Wc - cx0 : kBmCols;
  const int64_t row_elems = (int64_t)Wc * f * interleave;          // elements per fine row
  const int seg_full = kBmCols * f * interleave;                   // LDS row pitch (elements)
  const int seg = ncx * f * interleave;                            // elements of this segment
  const T* src = in + (size_t)plane * plane_stride + (int64_t)y * f * row_elems + (int64_t)cx0 * f * interleave;
  const int chunks = (int)(((size_t)seg * sizeof(T) + 15) / 16);   // the segment ends on a 16-byte boundary of the row
  for (int i = threadIdx.x; i < chunks * f; i += 256) {
    const int dy = i / chunks, c = i - dy * chunks;
    const uint4 v = *reinterpret_cast<const uint4*>(reinterpret_cast<const unsigned char*>(src + (int64_t)dy * row_elems) + (size_t)c * 16);
    *reinterpret_cast<uint4*>(bm_smem + ((size_t)dy * seg_full * sizeof(T) + (size_t)c * 16)) = v;
  }
  __syncthreads();
  for (int o = threadIdx.x; o < ncx * interleave; o += 256) {
    const int cx = o / interleave, b = o - cx * interleave;
    double s = 0.0;
    for (int dy = 0; dy < f; ++dy)
      for (int dx = 0; dx < f; ++dx) s += (double)tile[(size_t)dy * seg_full + (cx * f + dx) * interleave + b];
    const float m = (float)(s / (double)(f * f));
    const int band = interleave > 1 ? b : plane;
    out[(size_t)band * out_bs + ((int64_t)y * Wc + cx0 + cx) * out_ps] = m * scale;
  }
}

template <typename T>
static bool launch_block_mean_tile(const T* in, int64_t in_bs, int64_t in_ps, int nb, int Hc, int Wc, int f, float scale,
                                   float* out, int64_t out_bs, int64_t out_ps, hipStream_t s) {
  int interleave, planes;
  int64_t plane_stride;
  if (in_ps == 1) {                       // band-major planes
    interleave = 1;
    planes = nb;
    plane_stride = in_bs;
  } else if (in_bs == 1 && in_ps == nb) { // tightly packed band-last rows
    interleave = nb;
    planes = 1;
    plane_stride = 0;
  } else {
    return false;
  }
  const size_t row_bytes = (size_t)Wc * f * interleave * sizeof(T);
  const size_t lds = (size_t)f * kBmCols * f * interleave * sizeof(T);
  if ((((uintptr_t)in) & 15) || (row_bytes & 15) || ((plane_stride * sizeof(T)) & 15) || lds > 48 * 1024 ||
      ((size_t)kBmCols * f * interleave * sizeof(T) & 15) || Hc > 65535 || planes > 65535)
    return false;
  const dim3 grid((unsigned)((Wc + kBmCols - 1) / kBmCols), (unsigned)Hc, (unsigned)planes);
  hipLaunchKernelGGL(block_mean_tile_kernel<T>, grid, dim3(256), lds, s, in, plane_stride, interleave, Hc, Wc, f, scale, out,
                     out_bs, out_ps);
  return true;
}

// Separable taps of the pixel-centre aligned upsampling: pos = (i + 0.5) / f - 0.5, i0 = floor(pos), t = pos - i0,
// both neighbours clamped to the image (in this order: i1 = i0 + 1 is formed before i0 is clamped).
__device__ __forceinline__ void up_tap(int i, int f, int n, int* i0, int* i1, double* t) {
  const double pos = ((double)i + 0.5) / f - 0.5;
  const double fl = floor(pos);
  *t = pos - fl;
  const int a = (int)fl, b = a + 1;
  *i0 = a < 0 ? 0 : (a > n - 1 ? n - 1 : a);
  *i1 = b < 0 ? 0 : (b > n - 1 ? n - 1 : b);
}

// A workgroup writes a 256-column x kUpRows-row block of the fine image; a thread owns one output column: its
// column taps (a float64 division, floor, clamps) are computed once, the row taps once per row, and both are shared
// by all bands - the first version redid both divisions for every pixel and band and was compute-bound at
// 1.0 TB/s (planes) / 0.57 TB/s (band-last rows, one 4-byte store per 16 bytes).  Same expressions, same bits.
constexpr int kUpRows = 8;
constexpr int kUpMaxVec = 4;   // band-last rows of 4 floats are stored with one 16-byte store

// r04: the rows of a column that share their pair of coarse rows (y0, y1) - six in a row at the driver's 6 x - share `top` and `bot`,
// the two horizontal interpolations; and when the pair moves on by one row the old `bot` IS the new `top`.  A thread keeps them (per
// band, float64) while it walks down its column: four 16-byte taps, twelve conversions and eighteen float64 operations per pixel become
// one tap pair, six conversions and nine operations per SIX pixels.  The same expressions on the same operands: the same bits.
struct UpRowCache {
  int y0 = -1, y1 = -1;
  double top[kUpMaxVec], bot[kUpMaxVec];
};
__device__ __forceinline__ void up_rows4(UpRowCache& c, const float* __restrict__ in, int64_t in_ps, int Wc, int x0, int x1, int y0, int y1,
                                         double ux, double tx, int nb) {
  y0 = __builtin_amdgcn_readfirstlane(y0);        // the same for the whole workgroup: scalar branches
  y1 = __builtin_amdgcn_readfirstlane(y1);
  if (y0 == c.y0 && y1 == c.y1) return;
  auto row = [&](int y, double (&dst)[kUpMaxVec]) {
    const float4 q0 = *reinterpret_cast<const float4*>(in + ((int64_t)y * Wc + x0) * in_ps);
    const float4 q1 = *reinterpret_cast<const float4*>(in + ((int64_t)y * Wc + x1) * in_ps);
    const float a0[4] = {q0.x, q0.y, q0.z, q0.w}, a1[4] = {q1.x, q1.y, q1.z, q1.w};
#pragma unroll
    for (int b = 0; b < kUpMaxVec; ++b)
      if (b < nb) {
        const double v0 = a0[b], v1 = a1[b];
        dst[b] = v0 * ux + v1 * tx;
      }
  };
  if (y0 == c.y1) {
#pragma unroll
    for (int b = 0; b < kUpMaxVec; ++b) c.top[b] = c.bot[b];
  } else {
    row(y0, c.top);
  }
  if (y1 == y0) {
#pragma unroll
    for (int b = 0; b < kUpMaxVec; ++b) c.bot[b] = c.top[b];
  } else {
    row(y1, c.bot);
  }
  c.y0 = y0;
  c.y1 = y1;
}

// r03 (rocprofv3 of the reference driver: 260 us for 1024^2 x 3 -> 6144^2 x 4, the largest kernel of the least-squares variant):
// the ROW taps - a float64 division, a floor and two clamps, ~40 instructions - were recomputed by every thread for every
// output pixel although they are the same for the whole row: the first kUpRows threads compute them once per workgroup
// into LDS; and a band-last input row of 4 floats (IN4) is read with one 16-byte load per tap instead of one 4-byte load
// per tap and band.  Same expressions on the same operands, same bits.
template <bool VEC4, bool IN4>
__global__ __launch_bounds__(256) void bilinear_up_kernel(const float* __restrict__ in, int64_t in_bs, int64_t in_ps,
                                                          int Hc, int Wc, int f, int nb, float* __restrict__ out,
                                                          int64_t out_bs, int64_t out_ps) {
  __shared__ int ry0[kUpRows], ry1[kUpRows];
  __shared__ double rty[kUpRows];
  const int Hf = Hc * f, Wf = Wc * f;
  const int ybeg = blockIdx.y * kUpRows;
  const int yend = ybeg + kUpRows < Hf ? ybeg + kUpRows : Hf;
  if (threadIdx.x < kUpRows && ybeg + (int)threadIdx.x < yend) {
    int a, b;
    double t;
    up_tap(ybeg + threadIdx.x, f, Hc, &a, &b, &t);
    ry0[threadIdx.x] = a;
    ry1[threadIdx.x] = b;
    rty[threadIdx.x] = t;
  }
  __syncthreads();
  const int x = blockIdx.x * 256 + threadIdx.x;
  if (x >= Wf) return;
  int x0, x1;
  double tx;
  up_tap(x, f, Wc, &x0, &x1, &tx);
  const double ux = 1.0 - tx;
  if (!VEC4) {
    // planes: band by band, so that ONE pair of horizontal interpolations (see UpRowCache) serves the rows that share their coarse rows
    for (int b = 0; b < nb; ++b) {
      const float* src = in + (size_t)b * in_bs;
      int cy0 = -1, cy1 = -1;
      double top = 0.0, bot = 0.0;
      for (int y = ybeg; y < yend; ++y) {
        const int y0 = __builtin_amdgcn_readfirstlane(ry0[y - ybeg]), y1 = __builtin_amdgcn_readfirstlane(ry1[y - ybeg]);
        const double ty = rty[y - ybeg];
        const double uy = 1.0 - ty;
        if (y0 != cy0 || y1 != cy1) {
          if (y0 == cy1) {
            top = bot;
          } else {
            const double v00 = src[((int64_t)y0 * Wc + x0) * in_ps], v01 = src[((int64_t)y0 * Wc + x1) * in_ps];
            top = v00 * ux + v01 * tx;
          }
          if (y1 == y0) {
            bot = top;
          } else {
            const double v10 = src[((int64_t)y1 * Wc + x0) * in_ps], v11 = src[((int64_t)y1 * Wc + x1) * in_ps];
            bot = v10 * ux + v11 * tx;
          }
          cy0 = y0;
          cy1 = y1;
        }
        const int64_t p = (int64_t)y * Wf + x;
        st_stream(out + (size_t)b * out_bs + p * out_ps, (float)(top * uy + bot * ty));
      }
    }
    return;
  }
  UpRowCache rc;
  for (int y = ybeg; y < yend; ++y) {
    const int y0 = ry0[y - ybeg], y1 = ry1[y - ybeg];
    const double ty = rty[y - ybeg];
    const double uy = 1.0 - ty;
    const int64_t i00 = ((int64_t)y0 * Wc + x0) * in_ps, i01 = ((int64_t)y0 * Wc + x1) * in_ps;
    const int64_t i10 = ((int64_t)y1 * Wc + x0) * in_ps, i11 = ((int64_t)y1 * Wc + x1) * in_ps;
    const int64_t p = (int64_t)y * Wf + x;
    if (VEC4) {
      float r[kUpMaxVec] = {0.0f, 0.0f, 0.0f, 0.0f};
      if (IN4) {                                   // in_bs == 1, in_ps == 4, 16-byte aligned: the four taps as whole rows
        up_rows4(rc, in, in_ps, Wc, x0, x1, y0, y1, ux, tx, nb);
#pragma unroll
        for (int b = 0; b < kUpMaxVec; ++b)
          if (b < nb) r[b] = (float)(rc.top[b] * uy + rc.bot[b] * ty);
      } else {
#pragma unroll
        for (int b = 0; b < kUpMaxVec; ++b) {
          if (b < nb) {
            const float* src = in + (size_t)b * in_bs;
            const double v00 = src[i00], v01 = src[i01], v10 = src[i10], v11 = src[i11];
            const double top = v00 * ux + v01 * tx, bot = v10 * ux + v11 * tx;
            r[b] = (float)(top * uy + bot * ty);
          }
        }
      }
      st_stream(reinterpret_cast<float4*>(out + p * 4), make_float4(r[0], r[1], r[2], r[3]));
    } else {
      for (int b = 0; b < nb; ++b) {
        const float* src = in + (size_t)b * in_bs;
        const double v00 = src[i00], v01 = src[i01], v10 = src[i10], v11 = src[i11];
        const double top = v00 * ux + v01 * tx, bot = v10 * ux + v11 * tx;
        st_stream(out + (size_t)b * out_bs + p * out_ps, (float)(top * uy + bot * ty));
      }
    }
  }
}

// The upsampler as a PRODUCER for what the reference's driver does next with its output (poly_regression.py:159 ff.: finite
// mask, percentile stretch): while a workgroup still holds a pixel's values in registers it also writes the pixel's byte of
// the "all bands finite" mask and counts the values into the pass-1 histogram of the exact percentile select
// (csrc/hsr_select.hip; LDS histogram per channel, ballot-aggregated increments, integer atomics to the select's workspace at
// the end).  That removes two full reads of the fine image - 604 MB each at 6144 x 6144 x 4 - from the chain upsample ->
// valid_mask -> percentile_limits: r03 trace of match_pair, 145 us + 144 us of 1.7 ms.  Same values, same mask rule as
// valid_mask_kernel (no positivity test), same bins as select_hist_rows4_kernel<1>: the results are bit-identical.
// (Also built and measured: the REST of that chain without the fine image at all - the select's passes 2 and 3 and the
// stretch + polynomial each recomputing the fine values from the coarse image instead of reading them back - bit-identical
// and slower, 1.81 ms against 1.53 ms for match_pair: four float64 interpolations per sample cost more than one 604 MB read.)
constexpr int kUpHistRows = 32;      // rows per workgroup: the LDS histogram is zeroed and flushed once per 32 x 256 pixels

template <bool IN4>
__global__ __launch_bounds__(256) void bilinear_up_hist_kernel(const float* __restrict__ in, int64_t in_bs, int64_t in_ps,
                                                               int Hc, int Wc, int f, int nb, float* __restrict__ out,
                                                               uint8_t* __restrict__ mask_out, uint32_t* __restrict__ hist1) {
  extern __shared__ uint32_t uh[];            // [nb][kBins1]
  __shared__ int ry0[kUpHistRows], ry1[kUpHistRows];
  __shared__ double rty[kUpHistRows];
  const int Hf = Hc * f, Wf = Wc * f;
  const int ybeg = blockIdx.y * kUpHistRows;
  const int yend = ybeg + kUpHistRows < Hf ? ybeg + kUpHistRows : Hf;
  for (int i = threadIdx.x; i < nb * kBins1; i += 256) uh[i] = 0u;
  if (threadIdx.x < kUpHistRows && ybeg + (int)threadIdx.x < yend) {
    int a, b;
    double t;
    up_tap(ybeg + threadIdx.x, f, Hc, &a, &b, &t);
    ry0[threadIdx.x] = a;
    ry1[threadIdx.x] = b;
    rty[threadIdx.x] = t;
  }
  __syncthreads();
  const int x = blockIdx.x * 256 + threadIdx.x;
  const bool on = x < Wf;
  int x0, x1;
  double tx;
  up_tap(on ? x : Wf - 1, f, Wc, &x0, &x1, &tx);
  const double ux = 1.0 - tx;
  UpRowCache rc;
  uint32_t run_bin[kUpMaxVec] = {~0u, ~0u, ~0u, ~0u}, run_cnt[kUpMaxVec] = {0u, 0u, 0u, 0u};
  for (int y = ybeg; y < yend; ++y) {
    const int y0 = ry0[y - ybeg], y1 = ry1[y - ybeg];
    const double ty = rty[y - ybeg];
    const double uy = 1.0 - ty;
    const int64_t i00 = ((int64_t)y0 * Wc + x0) * in_ps, i01 = ((int64_t)y0 * Wc + x1) * in_ps;
    const int64_t i10 = ((int64_t)y1 * Wc + x0) * in_ps, i11 = ((int64_t)y1 * Wc + x1) * in_ps;
    float r[kUpMaxVec] = {0.0f, 0.0f, 0.0f, 0.0f};
    if (IN4) {
      up_rows4(rc, in, in_ps, Wc, x0, x1, y0, y1, ux, tx, nb);
#pragma unroll
      for (int b = 0; b < kUpMaxVec; ++b)
        if (b < nb) r[b] = (float)(rc.top[b] * uy + rc.bot[b] * ty);
    } else {
#pragma unroll
      for (int b = 0; b < kUpMaxVec; ++b) {
        if (b < nb) {
          const float* src = in + (size_t)b * in_bs;
          const double v00 = src[i00], v01 = src[i01], v10 = src[i10], v11 = src[i11];
          const double top = v00 * ux + v01 * tx, bot = v10 * ux + v11 * tx;
          r[b] = (float)(top * uy + bot * ty);
        }
      }
    }
    bool fin = true;
#pragma unroll
    for (int b = 0; b < kUpMaxVec; ++b)
      if (b < nb) fin = fin && finite_f32(r[b]);
    if (on) {
      const int64_t p = (int64_t)y * Wf + x;
      st_stream(reinterpret_cast<float4*>(out + p * 4), make_float4(r[0], r[1], r[2], r[3]));
      mask_out[p] = fin ? 1 : 0;
    }
    // r04: a thread walks DOWN its column, so consecutive samples are vertical neighbours of an interpolated image and nearly always
    // share their bin: the bin of the last sample and a count stay in registers per band, LDS is touched when the bin changes (as in
    // pass 1 of select_hist_kernel; the ballot / shuffle peel of hist_add_wave cost ~25 instructions per sample)
    if (on && fin) {
#pragma unroll
      for (int b = 0; b < kUpMaxVec; ++b)
        if (b < nb) {                                   // launch-uniform
          const uint32_t bin = __float_as_uint(r[b]) >> 21;          // raw top bits; the key's bin is formed at the flush
          if (bin != run_bin[b]) {
            if (run_cnt[b]) atomicAdd(&uh[b * kBins1 + (run_bin[b] ^ ((run_bin[b] & 0x400u) ? 0x7ffu : 0x400u))], run_cnt[b]);
            run_bin[b] = bin;
            run_cnt[b] = 0u;
          }
          ++run_cnt[b];
        }
    }
  }
#pragma unroll
  for (int b = 0; b < kUpMaxVec; ++b)
    if (b < nb && run_cnt[b]) atomicAdd(&uh[b * kBins1 + (run_bin[b] ^ ((run_bin[b] & 0x400u) ? 0x7ffu : 0x400u))], run_cnt[b]);
  __syncthreads();
  for (int i = threadIdx.x; i < nb * kBins1; i += 256)
    if (uh[i]) atomicAdd(&hist1[(size_t)(i / kBins1) * kHist1 + (i % kBins1)], uh[i]);
}

static int grid_for(int64_t n) {
  int64_t g = (n + 255) / 256;
  return (int)(g < 1 ? 1 : (g > 4096 ? 4096 : g));
}

}  // namespace hsr

using namespace hsr;

extern "C" int hsr_block_mean(const void* in_dev, int32_t in_dtype, int64_t in_bs, int64_t in_ps, int32_t nb,
                              int32_t Hc, int32_t Wc, int32_t factor, float scale, float* out_dev, int64_t out_bs,
                              int64_t out_ps, hsr_stream_t stream) {
  HSR_REQUIRE(in_dev && out_dev, HSR_ERR_INVALID, "hsr_block_mean: NULL pointer");
  HSR_REQUIRE(nb >= 1 && nb <= 65535 && Hc >= 1 && Wc >= 1 && factor >= 1 && factor <= 64, HSR_ERR_INVALID,
              "hsr_block_mean: bad shape");
  const dim3 grid(grid_for((int64_t)Hc * Wc), nb), block(256);
  hipStream_t s = (hipStream_t)stream;
  bool staged = false;
  switch (in_dtype) {
    case 0: staged = launch_block_mean_tile((const float*)in_dev, in_bs, in_ps, nb, Hc, Wc, factor, scale, out_dev, out_bs, out_ps, s); break;
    case 1: staged = launch_block_mean_tile((const uint8_t*)in_dev, in_bs, in_ps, nb, Hc, Wc, factor, scale, out_dev, out_bs, out_ps, s); break;
    case 2: staged = launch_block_mean_tile((const uint16_t*)in_dev, in_bs, in_ps, nb, Hc, Wc, factor, scale, out_dev, out_bs, out_ps, s); break;
    default: break;
  }
  if (staged) {
    HSR_LAUNCH_CHECK("block_mean_tile_kernel");
    return HSR_OK;
  }
  switch (in_dtype) {
    case 0: hipLaunchKernelGGL(block_mean_kernel<float>, grid, block, 0, s, (const float*)in_dev, in_bs, in_ps, Hc, Wc, factor, scale, nb, out_dev, out_bs, out_ps); break;
    case 1: hipLaunchKernelGGL(block_mean_kernel<uint8_t>, grid, block, 0, s, (const uint8_t*)in_dev, in_bs, in_ps, Hc, Wc, factor, scale, nb, out_dev, out_bs, out_ps); break;
    case 2: hipLaunchKernelGGL(block_mean_kernel<uint16_t>, grid, block, 0, s, (const uint16_t*)in_dev, in_bs, in_ps, Hc, Wc, factor, scale, nb, out_dev, out_bs, out_ps); break;
    default: set_error("hsr_block_mean: in_dtype=%d (0 float32, 1 uint8, 2 uint16)", in_dtype); return HSR_ERR_UNSUPPORTED;
  }
  HSR_LAUNCH_CHECK("block_mean_kernel");
  return HSR_OK;
}

extern "C" int hsr_bilinear_upsample(const float* in_dev, int64_t in_bs, int64_t in_ps, int32_t nb, int32_t Hc,
                                     int32_t Wc, int32_t factor, float* out_dev, int64_t out_bs, int64_t out_ps,
                                     hsr_stream_t stream) {
  HSR_REQUIRE(in_dev && out_dev, HSR_ERR_INVALID, "hsr_bilinear_upsample: NULL pointer");
  HSR_REQUIRE(nb >= 1 && nb <= 65535 && Hc >= 1 && Wc >= 1 && factor >= 1 && factor <= 64, HSR_ERR_INVALID,
              "hsr_bilinear_upsample: bad shape");
  HSR_REQUIRE((int64_t)Hc * factor <= 0x7fffffff / 2 && (int64_t)Wc * factor <= 0x7fffffff / 2, HSR_ERR_UNSUPPORTED,
              "hsr_bilinear_upsample: fine grid too large");
  const dim3 grid((unsigned)(((int64_t)Wc * factor + 255) / 256), (unsigned)(((int64_t)Hc * factor + kUpRows - 1) / kUpRows));
  HSR_REQUIRE(grid.y <= 65535u * 16u, HSR_ERR_UNSUPPORTED, "hsr_bilinear_upsample: fine grid too tall");
  const bool vec4 = out_bs == 1 && out_ps == 4 && nb <= kUpMaxVec && (((uintptr_t)out_dev) & 15) == 0;
  const bool in4 = in_bs == 1 && in_ps == 4 && (((uintptr_t)in_dev) & 15) == 0;
  if (vec4 && in4)
    hipLaunchKernelGGL((bilinear_up_kernel<true, true>), grid, dim3(256), 0, (hipStream_t)stream, in_dev, in_bs, in_ps, Hc, Wc,
                       factor, nb, out_dev, out_bs, out_ps);
  else if (vec4)
    hipLaunchKernelGGL((bilinear_up_kernel<true, false>), grid, dim3(256), 0, (hipStream_t)stream, in_dev, in_bs, in_ps, Hc, Wc,
                       factor, nb, out_dev, out_bs, out_ps);
  else
    hipLaunchKernelGGL((bilinear_up_kernel<false, false>), grid, dim3(256), 0, (hipStream_t)stream, in_dev, in_bs, in_ps, Hc, Wc,
                       factor, nb, out_dev, out_bs, out_ps);
  HSR_LAUNCH_CHECK("bilinear_up_kernel");
  return HSR_OK;
}

extern "C" int hsr_bilinear_upsample_mask_hist(const float* in_dev, int64_t in_bs, int64_t in_ps, int32_t nb, int32_t Hc,
                                               int32_t Wc, int32_t factor, float* out_dev, uint8_t* mask_out_dev,
                                               void* percentile_work_dev, hsr_stream_t stream) {
  HSR_REQUIRE(in_dev && out_dev && mask_out_dev && percentile_work_dev, HSR_ERR_INVALID, "hsr_bilinear_upsample_mask_hist: NULL pointer");
  HSR_REQUIRE(nb >= 1 && nb <= kUpMaxVec && Hc >= 1 && Wc >= 1 && factor >= 1 && factor <= 64, HSR_ERR_UNSUPPORTED,
              "hsr_bilinear_upsample_mask_hist: 1 <= nb <= 4 bands into band-last rows of 4 floats");
  HSR_REQUIRE((((uintptr_t)out_dev) & 15) == 0 && (int64_t)Hc * factor * Wc * factor < ((int64_t)1 << 31), HSR_ERR_UNSUPPORTED,
              "hsr_bilinear_upsample_mask_hist: output not 16-byte aligned or fine grid too large");
  int64_t off = 0, cnt = 0;
  int rc = hsr_percentile_hist_region(1, nb, &off, &cnt);
  if (rc != HSR_OK) return rc;
  uint32_t* hist1 = reinterpret_cast<uint32_t*>(static_cast<unsigned char*>(percentile_work_dev) + off);
  const dim3 grid((unsigned)(((int64_t)Wc * factor + 255) / 256), (unsigned)(((int64_t)Hc * factor + kUpHistRows - 1) / kUpHistRows));
  const size_t lds = (size_t)nb * kBins1 * sizeof(uint32_t);
  const bool in4 = in_bs == 1 && in_ps == 4 && (((uintptr_t)in_dev) & 15) == 0;
  if (in4)
    hipLaunchKernelGGL(bilinear_up_hist_kernel<true>, grid, dim3(256), lds, (hipStream_t)stream, in_dev, in_bs, in_ps, Hc, Wc, factor,
                       nb, out_dev, mask_out_dev, hist1);
  else
    hipLaunchKernelGGL(bilinear_up_hist_kernel<false>, grid, dim3(256), lds, (hipStream_t)stream, in_dev, in_bs, in_ps, Hc, Wc, factor,
                       nb, out_dev, mask_out_dev, hist1);
  HSR_LAUNCH_CHECK("bilinear_up_hist_kernel");
  return HSR_OK;
}
