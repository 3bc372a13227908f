// uint16 tile format of the training pairs: reflectance x 10000 as uint16, 65535 = nodata.
//
// Encode follows the reference writer, tiles_helpers/utils.py:362-374, statement by statement:
//   emit  = tile.astype(float32)
//   valid = isfinite(emit) & (emit != src_nodata)                       (second term only if the source has one)
//   q     = clip(rint(emit * scale).astype(int32), 0, nodata_u16 - 1)   (float32 product, round half to even)
//   out   = valid ? uint16(q) : nodata_u16
// The float32 -> int32 cast of a finite value outside the int32 range is what NumPy does on x86-64
// (cvttss2si -> INT32_MIN, which the clip then turns into 0); v_cvt_i32_f32 would saturate instead, so
// that case is handled explicitly to stay bit-exact with the reference on the same inputs.
// Decode is the convention of the reference's consumers (Pairs_EMIT_S2_demo-2.ipynb cell 65:
// `out *= float(s2_scale)` on a float32 array): x = float32(u) * float32(scale); nodata -> NaN.
// Both are pure streaming kernels: 6 bytes per sample, HBM-bound.
#include "hsr_common.h"

namespace hsr {

__device__ __forceinline__ uint16_t encode_sample(float x, float scale, bool has_src_nodata, float src_nodata,
                                                  int32_t nodata_u16) {
  const bool valid = finite_f32(x) && !(has_src_nodata && x == src_nodata);
  if (!valid) return (uint16_t)nodata_u16;
  const float r = rintf(x * scale);   // -ffp-contract=off: the product is rounded to float32 first
  int32_t q;
  if (!(r >= -2147483648.0f && r < 2147483648.0f)) q = INT32_MIN;   // x86 "integer indefinite"
  else q = (int32_t)r;
  q = q < 0 ? 0 : (q > nodata_u16 - 1 ? nodata_u16 - 1 : q);
  return (uint16_t)q;
}

__global__ __launch_bounds__(256) void tile_encode_kernel(const float* __restrict__ x, int64_t n, float scale,
                                                          int has_src_nodata, float src_nodata, int32_t nodata_u16,
                                                          uint16_t* __restrict__ out) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int64_t n4 = n >> 2;
  const bool vec = ((((uintptr_t)x) & 15) == 0) && ((((uintptr_t)out) & 7) == 0);
  if (vec) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
      const float4 v = ld_stream(reinterpret_cast<const float4*>(x) + i);
      const uint32_t a = encode_sample(v.x, scale, has_src_nodata, src_nodata, nodata_u16);
      const uint32_t b = encode_sample(v.y, scale, has_src_nodata, src_nodata, nodata_u16);
      const uint32_t c = encode_sample(v.z, scale, has_src_nodata, src_nodata, nodata_u16);
      const uint32_t d = encode_sample(v.w, scale, has_src_nodata, src_nodata, nodata_u16);
      uint2 o = make_uint2(a | (b << 16), c | (d << 16));
      reinterpret_cast<uint2*>(out)[i] = o;
    }
    for (int64_t i = n4 * 4 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
      out[i] = encode_sample(x[i], scale, has_src_nodata, src_nodata, nodata_u16);
  } else {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
      out[i] = encode_sample(x[i], scale, has_src_nodata, src_nodata, nodata_u16);
  }
}

__device__ __forceinline__ float decode_sample(uint32_t u, float scale, uint32_t nodata) {
  return u == nodata ? __uint_as_float(0x7fc00000u) : (float)u * scale;
}

__global__ __launch_bounds__(256) void tile_decode_kernel(const uint16_t* __restrict__ u, int64_t n, float scale,
                                                          uint32_t nodata, float* __restrict__ out) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int64_t n4 = n >> 2;
  const bool vec = ((((uintptr_t)u) & 7) == 0) && ((((uintptr_t)out) & 15) == 0);
  if (vec) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
      const uint2 v = reinterpret_cast<const uint2*>(u)[i];
      st_stream(reinterpret_cast<float4*>(out) + i,
                make_float4(decode_sample(v.x & 0xffffu, scale, nodata), decode_sample(v.x >> 16, scale, nodata),
                            decode_sample(v.y & 0xffffu, scale, nodata), decode_sample(v.y >> 16, scale, nodata)));
    }
    for (int64_t i = n4 * 4 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
      out[i] = decode_sample(u[i], scale, nodata);
  } else {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
      out[i] = decode_sample(u[i], scale, nodata);
  }
}


// ---- ENVI interleaves -> pixel-major ---------------------------------------------------------------
// load_emit_envi_rfl (reference s2_emit/emit_io.py:7-16) returns the (H, W, B) array whatever the file's interleave;
// on disk EMIT cubes are usually BIL or BSQ (gdalwarp output).  Both are batched 2-D transposes
//     in [batch][R][C] -> out [batch][C][R]        BIL: batch = lines, R = bands, C = samples
//                                                  BSQ: batch = 1,     R = bands, C = lines * samples
// done through a 64 x 64 LDS tile (rows padded by one word: conflict-free both ways), coalesced 256-byte segments
// along C on the way in and along R on the way out.  HBM-bound: 2 x cube bytes.  TI -> TO converts on the fly
// (int16 / uint16 / float32 in, float32 or uint16 out; the integer -> float32 conversion is exact).
template <typename TI, typename TO>
__global__ __launch_bounds__(256) void transpose_rc_kernel(const TI* __restrict__ in, TO* __restrict__ out, int64_t R, int64_t C,
                                                           int64_t tiles_c) {
  __shared__ TO tile[64][65];
  const int64_t batch = blockIdx.y;
  const int64_t tc = blockIdx.x % tiles_c, tr = blockIdx.x / tiles_c;
  const TI* src = in + batch * R * C;
  TO* dst = out + batch * R * C;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;       // 4 rows of 64 per pass
  const int64_t c0 = tc * 64, r0 = tr * 64;
#pragma unroll 4
  for (int i = ty; i < 64; i += 4) {
    const int64_t r = r0 + i, c = c0 + tx;
    if (r < R && c < C) tile[i][tx] = (TO)ld_stream(src + r * C + c);
  }
  __syncthreads();
#pragma unroll 4
  for (int i = ty; i < 64; i += 4) {
    const int64_t c = c0 + i, r = r0 + tx;
    if (r < R && c < C) dst[c * R + r] = tile[tx][i];
  }
}

template <typename TI, typename TO>
static int launch_transpose(const void* in, void* out, int64_t batch, int64_t R, int64_t C, hipStream_t s) {
  const int64_t tiles_c = (C + 63) / 64, tiles_r = (R + 63) / 64;
  HSR_REQUIRE(tiles_c * tiles_r < ((int64_t)1 << 31) && batch <= 65535, HSR_ERR_UNSUPPORTED, "hsr_interleave_to_bip: shape too large");
  hipLaunchKernelGGL((transpose_rc_kernel<TI, TO>), dim3((unsigned)(tiles_c * tiles_r), (unsigned)batch), dim3(256), 0, s,
                     (const TI*)in, (TO*)out, R, C, tiles_c);
  HSR_LAUNCH_CHECK("transpose_rc_kernel");
  return HSR_OK;
}

static unsigned stream_grid(int64_t n) {
  int64_t g = (n / 4 + 255) / 256;
  if (g < 1) g = 1;
  if (g > 256 * 16) g = 256 * 16;
  return (unsigned)g;
}

}  // namespace hsr

extern "C" int hsr_tile_encode_u16(const float* x_dev, int64_t n, float scale, int32_t has_src_nodata,
                                   float src_nodata, int32_t nodata_u16, uint16_t* out_dev, hsr_stream_t stream) {
  HSR_REQUIRE(n >= 0, HSR_ERR_INVALID, "hsr_tile_encode_u16: n < 0");
  HSR_REQUIRE(nodata_u16 >= 1 && nodata_u16 <= 0xffff, HSR_ERR_INVALID, "hsr_tile_encode_u16: nodata_u16=%d outside [1,65535]", nodata_u16);
  if (n == 0) return HSR_OK;
  HSR_REQUIRE(x_dev && out_dev, HSR_ERR_INVALID, "hsr_tile_encode_u16: NULL pointer");
  hipLaunchKernelGGL(hsr::tile_encode_kernel, dim3(hsr::stream_grid(n)), dim3(256), 0, (hipStream_t)stream, x_dev, n,
                     scale, has_src_nodata, src_nodata, nodata_u16, out_dev);
  HSR_LAUNCH_CHECK("tile_encode_kernel");
  return HSR_OK;
}

extern "C" int hsr_tile_decode_u16(const uint16_t* u_dev, int64_t n, float scale, int32_t nodata, float* out_dev,
                                   hsr_stream_t stream) {
  HSR_REQUIRE(n >= 0, HSR_ERR_INVALID, "hsr_tile_decode_u16: n < 0");
  HSR_REQUIRE(nodata <= 0xffff, HSR_ERR_INVALID, "hsr_tile_decode_u16: nodata=%d is not a uint16 value (negative = none)", nodata);
  if (n == 0) return HSR_OK;
  HSR_REQUIRE(u_dev && out_dev, HSR_ERR_INVALID, "hsr_tile_decode_u16: NULL pointer");
  hipLaunchKernelGGL(hsr::tile_decode_kernel, dim3(hsr::stream_grid(n)), dim3(256), 0, (hipStream_t)stream, u_dev, n,
                     scale, nodata < 0 ? 0x10000u : (uint32_t)nodata, out_dev);
  HSR_LAUNCH_CHECK("tile_decode_kernel");
  return HSR_OK;
}

extern "C" int hsr_interleave_to_bip(const void* in_dev, int32_t in_dtype, int32_t interleave, int64_t lines, int64_t samples,
                                     int64_t bands, void* out_dev, int32_t out_dtype, hsr_stream_t stream) {
  HSR_REQUIRE(in_dev && out_dev, HSR_ERR_INVALID, "hsr_interleave_to_bip: NULL pointer");
  HSR_REQUIRE(lines >= 1 && samples >= 1 && bands >= 1, HSR_ERR_INVALID, "hsr_interleave_to_bip: empty shape");
  HSR_REQUIRE(interleave == 1 || interleave == 2, HSR_ERR_INVALID, "hsr_interleave_to_bip: interleave must be 1 (BIL) or 2 (BSQ)");
  // BIL: per line a (bands x samples) matrix; BSQ: one (bands x lines*samples) matrix.  BIL with more than 65535 lines:
  // fold as many lines as needed into separate launches.
  hipStream_t s = (hipStream_t)stream;
  const int64_t R = bands;
  const int64_t C = interleave == 1 ? samples : lines * samples;
  const int64_t batch = interleave == 1 ? lines : 1;
  const size_t isz = in_dtype == 0 ? 4 : 2, osz = out_dtype == 0 ? 4 : 2;
  for (int64_t b0 = 0; b0 < batch; b0 += 65535) {
    const int64_t nb = batch - b0 < 65535 ? batch - b0 : 65535;
    const char* ip = (const char*)in_dev + (size_t)b0 * R * C * isz;
    char* op = (char*)out_dev + (size_t)b0 * R * C * osz;
    int rc;
    if (in_dtype == 0 && out_dtype == 0) rc = hsr::launch_transpose<float, float>(ip, op, nb, R, C, s);
    else if (in_dtype == 2 && out_dtype == 2) rc = hsr::launch_transpose<uint16_t, uint16_t>(ip, op, nb, R, C, s);
    else if (in_dtype == 2 && out_dtype == 0) rc = hsr::launch_transpose<uint16_t, float>(ip, op, nb, R, C, s);
    else if (in_dtype == 3 && out_dtype == 0) rc = hsr::launch_transpose<int16_t, float>(ip, op, nb, R, C, s);
    else {
      hsr::set_error("hsr_interleave_to_bip: unsupported dtype pair (%d -> %d)", in_dtype, out_dtype);
      return HSR_ERR_UNSUPPORTED;
    }
    if (rc != HSR_OK) return rc;
  }
  return HSR_OK;
}
