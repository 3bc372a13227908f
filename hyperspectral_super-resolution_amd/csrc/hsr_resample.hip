// f1: grid-aligned resamplers between the phases of the pipeline (reference: downsample_s2_to_grid /
// reproject_stack_to_grid, notebook-only, Pairs_EMIT_S2_demo-2.ipynb cell 73 raw lines 4538-4599, which
// call rasterio/GDAL `reproject`).  The repository warps EMIT onto the S2 UTM grid at exactly 6 x 10 m
// (EMIT_data/emit_proj.py:791-797) and cuts tiles as exact 6x windows (tiles_helpers/utils.py:256-277), so
// for the pairs this path sees the two warps reduce to an f x f block mean and a pixel-centre-aligned
// separable bilinear upsampling.  GDAL's edge / nodata conventions are PARITY UNPINNED (rasterio absent).
// Both kernels are trivially HBM-bound streams; images use (band_stride, pixel_stride) addressing.
#include "hsr_common.h"

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

template <bool VEC4>
__global__ __launch_bounds__(256) void bilinear_up_kernel(const float* __restrict__ in, int64_t in_bs, int64_t in_ps,
                                                          int Hc, int Wc, int f, int nb, float* __restrict__ out,
                                                          int64_t out_bs, int64_t out_ps) {
  const int Hf = Hc * f, Wf = Wc * f;
  const int x = blockIdx.x * 256 + threadIdx.x;
  if (x >= Wf) return;
  int x0, x1;
  double tx;
  up_tap(x, f, Wc, &x0, &x1, &tx);
  const double ux = 1.0 - tx;
  const int ybeg = blockIdx.y * kUpRows;
  const int yend = ybeg + kUpRows < Hf ? ybeg + kUpRows : Hf;
  for (int y = ybeg; y < yend; ++y) {
    int y0, y1;
    double ty;
    up_tap(y, f, Hc, &y0, &y1, &ty);
    const double uy = 1.0 - ty;
    const int64_t i00 = ((int64_t)y0 * Wc + x0) * in_ps, i01 = ((int64_t)y0 * Wc + x1) * in_ps;
    const int64_t i10 = ((int64_t)y1 * Wc + x0) * in_ps, i11 = ((int64_t)y1 * Wc + x1) * in_ps;
    const int64_t p = (int64_t)y * Wf + x;
    if (VEC4) {
      float r[kUpMaxVec] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
      for (int b = 0; b < kUpMaxVec; ++b) {
        if (b < nb) {
          const float* src = in + (size_t)b * in_bs;
          const double v00 = src[i00], v01 = src[i01], v10 = src[i10], v11 = src[i11];
          const double top = v00 * ux + v01 * tx, bot = v10 * ux + v11 * tx;
          r[b] = (float)(top * uy + bot * ty);
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
  if (vec4)
    hipLaunchKernelGGL(bilinear_up_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, in_dev, in_bs, in_ps, Hc, Wc,
                       factor, nb, out_dev, out_bs, out_ps);
  else
    hipLaunchKernelGGL(bilinear_up_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, in_dev, in_bs, in_ps, Hc, Wc,
                       factor, nb, out_dev, out_bs, out_ps);
  HSR_LAUNCH_CHECK("bilinear_up_kernel");
  return HSR_OK;
}
