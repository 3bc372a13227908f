// Library-level entry points of libhsr_mi355x: error state, sizing helpers, and a pure-read
// bandwidth probe used by bench.py to put the HBM roofline of THIS box next to the vendor peak.
#include <string.h>

#include "hsr_common.h"

namespace hsr {

static thread_local char g_error[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_error, sizeof(g_error), fmt, ap);
  va_end(ap);
}

// Streams `n16` 16-byte words once and folds them into one float per workgroup (kept so the loads
// cannot be eliminated).  Same access shape and cache policy as the cube stream of K1: 16 B per lane,
// coalesced, non-temporal.
__global__ __launch_bounds__(256) void probe_read_kernel(const float4* __restrict__ src, int64_t n16,
                                                         float* __restrict__ sink) {
  float acc = 0.0f;
  const int64_t stride = (int64_t)gridDim.x * 256;
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  for (; i + 3 * stride < n16; i += 4 * stride) {
    const float4 a = hsr::ld_stream(src + i), b = hsr::ld_stream(src + i + stride), c = hsr::ld_stream(src + i + 2 * stride),
                 d = hsr::ld_stream(src + i + 3 * stride);
    acc += (a.x + a.y + a.z + a.w) + (b.x + b.y + b.z + b.w) + (c.x + c.y + c.z + c.w) + (d.x + d.y + d.z + d.w);
  }
  for (; i < n16; i += stride) {
    const float4 a = hsr::ld_stream(src + i);
    acc += a.x + a.y + a.z + a.w;
  }
  acc += __shfl_xor(acc, 32, 64);
  acc += __shfl_xor(acc, 16, 64);
  acc += __shfl_xor(acc, 8, 64);
  acc += __shfl_xor(acc, 4, 64);
  acc += __shfl_xor(acc, 2, 64);
  acc += __shfl_xor(acc, 1, 64);
  if ((threadIdx.x & 63) == 0) atomicAdd(&sink[blockIdx.x & 63], acc);
}

// The load shape of K1 with nothing else: 512 persistent workgroups of 512 threads (2 per CU), each streaming
// 72 KiB slabs (workgroup b takes slabs b, b + grid, ...) HBM -> LDS with non-temporal global_load_lds_dwordx4,
// one barrier per slab, nothing read back.  What this kernel reaches is the ceiling of K1's staging structure on
// this box (round 1 lab: 6.9 TB/s); a plain global_load stream (mode 1) reads 5.6 TB/s.
constexpr int kProbeSlab = 72 * 1024;
// SLAB bytes per barrier: 72 KiB = a float32 group of K1 (modes 0), 36 KiB = a uint16 group (modes 2, 3).
template <int SLAB>
__global__ __launch_bounds__(512, 4) void probe_lds_dma_kernel(const char* __restrict__ src, int64_t n16, float* __restrict__ sink) {
  extern __shared__ __attribute__((aligned(16))) unsigned char psmem[];
  typedef __attribute__((address_space(1))) const void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  constexpr int kChunks = SLAB / 16;
  const int64_t nslabs = (n16 + kChunks - 1) / kChunks;
  for (int64_t sl = blockIdx.x; sl < nslabs; sl += gridDim.x) {
    const int64_t base = sl * kChunks;
    const int64_t left = n16 - base;
    const int nch = left < kChunks ? (int)left : kChunks;
    for (int c0 = wave * 64; c0 < nch; c0 += 512) {
      const int c = c0 + lane;
      if (c < nch)
        __builtin_amdgcn_global_load_lds((gptr_t)(src + (base + c) * 16), (lptr_t)(psmem + (size_t)c0 * 16), 16, 0, kGldsStream);
    }
    __syncthreads();
  }
  if (threadIdx.x == 0 && blockIdx.x == 0) sink[0] = reinterpret_cast<const float*>(psmem)[0];   // keeps the LDS image observable
}

template <int SLAB>
static int launch_probe_lds(const void* buf, int64_t bytes, int lds_bytes, int grid_cap, float* sink, hipStream_t stream) {
  auto kern = probe_lds_dma_kernel<SLAB>;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
  (void)hipGetLastError();
  const int64_t nslabs = (bytes + SLAB - 1) / SLAB;
  hipLaunchKernelGGL(kern, dim3((unsigned)(nslabs < grid_cap ? nslabs : grid_cap)), dim3(512), lds_bytes, stream,
                     (const char*)buf, bytes / 16, sink);
  HSR_LAUNCH_CHECK("probe_lds_dma_kernel");
  return HSR_OK;
}

}  // namespace hsr

extern "C" int hsr_abi_version(void) { return HSR_ABI_VERSION; }

extern "C" const char* hsr_last_error(void) { return hsr::g_error; }

extern "C" int hsr_moment_count(int32_t deg) { return deg >= 1 && deg <= HSR_MAX_DEG ? hsr::moment_count(deg) : -1; }

extern "C" size_t hsr_partials_bytes(int32_t nb, int32_t deg) {
  if (nb < 1 || nb > HSR_MAX_BANDS || deg < 1 || deg > HSR_MAX_DEG) return 0;
  return (size_t)nb * hsr::moment_count(deg) * HSR_MAX_PARTIALS * sizeof(double);
}

extern "C" int hsr_probe_read(const void* buf_dev, int64_t bytes, int32_t mode, float* sink_dev, hsr_stream_t stream) {
  HSR_REQUIRE(buf_dev && sink_dev && bytes >= 16, HSR_ERR_INVALID, "hsr_probe_read: bad argument");
  HSR_REQUIRE(((uintptr_t)buf_dev & 15) == 0, HSR_ERR_INVALID, "hsr_probe_read: buffer not 16-byte aligned");
  HSR_REQUIRE(mode >= 0 && mode <= 3, HSR_ERR_INVALID, "hsr_probe_read: mode must be 0..3");
  if (mode == 1) {
    hipLaunchKernelGGL(hsr::probe_read_kernel, dim3(2048), dim3(256), 0, (hipStream_t)stream,
                       (const float4*)buf_dev, bytes / 16, sink_dev);
    HSR_LAUNCH_CHECK("probe_read_kernel");
    return HSR_OK;
  }
  // LDS footprint decides how many workgroups share a CU (160 KiB): 72 KiB -> 2, 36 KiB -> 4
  if (mode == 0) return hsr::launch_probe_lds<hsr::kProbeSlab>(buf_dev, bytes, hsr::kProbeSlab, 512, sink_dev, (hipStream_t)stream);
  if (mode == 2) return hsr::launch_probe_lds<hsr::kProbeSlab / 2>(buf_dev, bytes, hsr::kProbeSlab, 512, sink_dev, (hipStream_t)stream);
  return hsr::launch_probe_lds<hsr::kProbeSlab / 2>(buf_dev, bytes, hsr::kProbeSlab / 2, 1024, sink_dev, (hipStream_t)stream);
}
