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

__global__ __launch_bounds__(256) void bilinear_up_kernel(const float* __restrict__ in, int64_t in_bs, int64_t in_ps,
                                                          int Hc, int Wc, int f, int nb, float* __restrict__ out,
                                                          int64_t out_bs, int64_t out_ps) {
  const int Hf = Hc * f, Wf = Wc * f;
  const int64_t total = (int64_t)Hf * Wf;
  const int b = blockIdx.y;
  const float* src = in + (size_t)b * in_bs;
  for (int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x; p < total; p += (int64_t)gridDim.x * 256) {
    const int y = (int)(p / Wf), x = (int)(p - (int64_t)y * Wf);
    const double py = ((double)y + 0.5) / f - 0.5, px = ((double)x + 0.5) / f - 0.5;
    const double fy = floor(py), fx = floor(px);
    const double ty = py - fy, tx = px - fx;
    int y0 = (int)fy, x0 = (int)fx, y1 = y0 + 1, x1 = x0 + 1;
    y0 = y0 < 0 ? 0 : (y0 > Hc - 1 ? Hc - 1 : y0);
    y1 = y1 < 0 ? 0 : (y1 > Hc - 1 ? Hc - 1 : y1);
    x0 = x0 < 0 ? 0 : (x0 > Wc - 1 ? Wc - 1 : x0);
    x1 = x1 < 0 ? 0 : (x1 > Wc - 1 ? Wc - 1 : x1);
    const double v00 = src[((int64_t)y0 * Wc + x0) * in_ps], v01 = src[((int64_t)y0 * Wc + x1) * in_ps];
    const double v10 = src[((int64_t)y1 * Wc + x0) * in_ps], v11 = src[((int64_t)y1 * Wc + x1) * in_ps];
    const double top = v00 * (1.0 - tx) + v01 * tx, bot = v10 * (1.0 - tx) + v11 * tx;
    out[(size_t)b * out_bs + p * out_ps] = (float)(top * (1.0 - ty) + bot * ty);
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
  hipLaunchKernelGGL(bilinear_up_kernel, dim3(grid_for((int64_t)Hc * factor * Wc * factor), nb), dim3(256), 0,
                     (hipStream_t)stream, in_dev, in_bs, in_ps, Hc, Wc, factor, nb, out_dev, out_bs, out_ps);
  HSR_LAUNCH_CHECK("bilinear_up_kernel");
  return HSR_OK;
}
