// Shared helpers of libhsr_mi355x (gfx950 only; no portability layer on purpose).
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/hsr.h"

namespace hsr {

void set_error(const char* fmt, ...);

inline int check_hip(hipError_t e, const char* what) {
  if (e == hipSuccess) return HSR_OK;
  set_error("%s: %s", what, hipGetErrorString(e));
  return HSR_ERR_HIP;
}

#define HSR_REQUIRE(cond, code, ...)  \
  do {                                \
    if (!(cond)) {                    \
      hsr::set_error(__VA_ARGS__);    \
      return (code);                  \
    }                                 \
  } while (0)

#define HSR_LAUNCH_CHECK(name)                                   \
  do {                                                           \
    int rc_ = hsr::check_hip(hipGetLastError(), name " launch"); \
    if (rc_ != HSR_OK) return rc_;                               \
  } while (0)

constexpr int kWave = 64;

// Number of moments for a degree.
__host__ __device__ constexpr int moment_count(int deg) { return 3 * deg + 2; }

// Partial slots: fixed function of the pixel count only, so that the summation tree (and with it
// every bit of the fitted coefficients) does not depend on the device or the launch environment.
inline int partial_slots(int64_t npix) {
  int64_t tiles = (npix + HSR_TILE_PIXELS - 1) / HSR_TILE_PIXELS;
  if (tiles < 1) tiles = 1;
  return (int)(tiles < 512 ? tiles : 512);  // 256 CUs x 2 resident workgroups
}

// Wave-level butterfly sum of a double (fixed tree -> deterministic).
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// Streaming (non-temporal) accesses: everything on this path is touched exactly once, and letting
// the 1.2 GB cube stream allocate in the 4 MB L2s evicts the dirty output lines early - measured on
// K1: 0.2285 ms with plain loads, 0.2002 ms with `nt` on the LDS-DMA, 0.1957 ms with `nt` stores too.
typedef float native_f32x4 __attribute__((ext_vector_type(4)));
template <typename T>
__device__ __forceinline__ T ld_stream(const T* p) { return __builtin_nontemporal_load(p); }
template <typename T>
__device__ __forceinline__ void st_stream(T* p, T v) { __builtin_nontemporal_store(v, p); }
// HIP's float4 is a struct; the builtins want the native vector type (same size and alignment)
__device__ __forceinline__ float4 ld_stream(const float4* p) {
  const native_f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const native_f32x4*>(p));
  return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void st_stream(float4* p, float4 v) {
  native_f32x4 n = {v.x, v.y, v.z, v.w};
  __builtin_nontemporal_store(n, reinterpret_cast<native_f32x4*>(p));
}
constexpr int kGldsStream = 2;  // aux/cpol bits of global_load_lds: 2 = nt

__device__ __forceinline__ bool finite_f32(float v) {
  return (__float_as_uint(v) & 0x7f800000u) != 0x7f800000u;
}

// float32(clip((v - lo) / (hi - lo + 1e-12), 0, 1)) in float64, as s2_emit/color.py:33 evaluates it
// (np.percentile returns float64 limits, so the whole expression is float64 before the store).
__device__ __forceinline__ float stretch_f64(float v, double lo, double hi) {
  double r = ((double)v - lo) / (hi - lo + 1e-12);
  r = r < 0.0 ? 0.0 : (r > 1.0 ? 1.0 : r);  // NaN falls through both compares, like np.clip
  return (float)r;
}

}  // namespace hsr
