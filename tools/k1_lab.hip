// K1 lab: A/B of kernel structures for the SRF band integration in ONE process (interleaved rounds,
// hipEvent timing).  Not part of the library; build (after `make -C hyperspectral_super-resolution_amd/csrc`):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/k1_lab.hip -Lhyperspectral_super-resolution_amd/lib -lhsr_mi355x \
//         -Wl,-rpath,'$ORIGIN/../hyperspectral_super-resolution_amd/lib' -o tools/k1_lab
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <vector>
#include <string>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)

typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int B = 285;
constexpr int NBMAX = 16;

struct Bands { int k0[NBMAX]; int klen[NBMAX]; int woff[NBMAX]; int wtaps; };
#include "../include/hsr.h"

// ---------------------------------------------------------------- V_lds<P, MODE>: LDS-staged, lane = pixel
// MODE 0 = full, 1 = load only (ceiling of the staging structure), 2 = load + scan
template <int P, int THREADS, int MINW, int MODE, bool RAWBAR = false, int ASSIGN = 0, bool NT = false, int AUX = 0, bool SMALLW = false>
__global__ __launch_bounds__(THREADS, MINW) void k_lds(const float* __restrict__ cube, int64_t npix, const float* __restrict__ wn,
                                                 Bands bands, int nb, float* __restrict__ planes, int64_t stride) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* tile = (float*)smem;
  uint32_t* flags = (uint32_t*)(smem + (size_t)P * B * 4);
  float* wl = (float*)(flags + 64);
  const int t = threadIdx.x, lane = t & 63;
  for (int b = 0; b < nb; ++b)
    for (int i = t; i < bands.klen[b]; i += THREADS) wl[bands.woff[b] + i] = wn[(size_t)b * B + bands.k0[b] + i];
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  constexpr int NW = THREADS / 64;
  constexpr int nchunk = P * B / 4;      // 16-byte chunks per tile (P*B*4/16)
  const int64_t ntiles = npix / P;       // lab: npix multiple of P
  float sink = 0.f;
  // ASSIGN 0: tile = block + k*grid (strided)   ASSIGN 1: each block owns a contiguous run of tiles
  const int64_t per = (ntiles + gridDim.x - 1) / gridDim.x;
  const int64_t tbeg = ASSIGN ? (int64_t)blockIdx.x * per : blockIdx.x;
  const int64_t tend = ASSIGN ? (tbeg + per < ntiles ? tbeg + per : ntiles) : ntiles;
  const int64_t tstep = ASSIGN ? 1 : gridDim.x;
  for (int64_t tileidx = tbeg; tileidx < tend; tileidx += tstep) {
    const int64_t pix0 = tileidx * P;
    const char* srcb = (const char*)(cube + pix0 * B);
    if (t < P) flags[t] = 0u;
    for (int c0 = wave * 64; c0 < nchunk; c0 += THREADS) {
      const int c = c0 + lane;
      if (c < nchunk)
        __builtin_amdgcn_global_load_lds((gptr_t)(srcb + (size_t)c * 16), (lptr_t)(smem + (size_t)c0 * 16), 16, 0, AUX);
    }
    __syncthreads();
    if (MODE != 1) {
      const float4* t4 = (const float4*)smem;
      for (int c = t; c < nchunk; c += THREADS) {
        const float4 v = t4[c];
        float z = v.x * 0.0f; z = fmaf(v.y, 0.0f, z); z = fmaf(v.z, 0.0f, z); z = fmaf(v.w, 0.0f, z);
        if (z != z) {
          const int e = c * 4;
          flags[(e + 0) / B] = 1u; flags[(e + 3) / B] = 1u;   // lab: coarse flagging is enough
        }
      }
      __syncthreads();
    }
    if (MODE == 0 || MODE == 3) {
      // lane -> pixel (lane % P), band group = wave * (64/P) + lane / P
      constexpr int SUB = 64 / P;                 // 1 for P=64, 2 for P=32
      const int p = lane % P;
      const int grp = wave * SUB + lane / P;
      constexpr int NG = NW * SUB;
      const bool slow = flags[p] != 0u;
      const float* v = tile + p * B;
#pragma unroll
      for (int j = 0; j < (NBMAX + NG - 1) / NG; ++j) {
        const int b = grp + NG * j;
        if (b < nb) {
          const float* w = wn + (size_t)b * B;
          float acc = 0.f;
          if (!slow) {
            const int k0 = bands.k0[b], n4 = bands.klen[b] >> 2;
            const float4* w4 = (const float4*)(wl + bands.woff[b]);
            const float* vs = v + k0;
            for (int i = 0; i < n4; ++i) {
              const float4 ww = w4[i];
              acc = fmaf(ww.x, vs[4 * i], acc); acc = fmaf(ww.y, vs[4 * i + 1], acc);
              acc = fmaf(ww.z, vs[4 * i + 2], acc); acc = fmaf(ww.w, vs[4 * i + 3], acc);
            }
          } else {
            for (int k = 0; k < B; ++k) acc = fmaf(w[k], v[k], acc);
          }
          if (MODE == 0) { const int64_t po = SMALLW ? ((pix0 + p) & 16383) : (pix0 + p); if (NT) __builtin_nontemporal_store(acc, &planes[b * stride + po]); else planes[b * stride + po] = acc; } else sink += acc;
        }
      }
    } else if (MODE == 4) {
      constexpr int SUB = 64 / P; const int p = lane % P; const int grp = wave * SUB + lane / P; constexpr int NG = NW * SUB;
#pragma unroll
      for (int j = 0; j < (NBMAX + NG - 1) / NG; ++j) { const int b = grp + NG * j; if (b < nb) planes[b * stride + pix0 + p] = tile[p * B + b]; }
    } else {
      sink += tile[(t * 37) % (P * B)];
    }
    if (RAWBAR) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    else __syncthreads();
  }
  if (MODE != 0 && MODE != 4 && sink == 12345.678f) planes[0] = sink;
}


// ---------------------------------------------------------------- diagnostic: phase stamps (shares, not speed)
__device__ __forceinline__ unsigned long long stamp() {
  unsigned long long t;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
  return t;
}
template <int P, int THREADS, int MINW>
__global__ __launch_bounds__(THREADS, MINW) void k_lds_timed(const float* __restrict__ cube, int64_t npix, const float* __restrict__ wn,
                                                 Bands bands, int nb, float* __restrict__ planes, int64_t stride,
                                                 unsigned long long* __restrict__ tout) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* tile = (float*)smem;
  uint32_t* flags = (uint32_t*)(smem + (size_t)P * B * 4);
  float* wl = (float*)(flags + 64);
  const int t = threadIdx.x, lane = t & 63;
  for (int b = 0; b < nb; ++b)
    for (int i = t; i < bands.klen[b]; i += THREADS) wl[bands.woff[b] + i] = wn[(size_t)b * B + bands.k0[b] + i];
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  constexpr int NW = THREADS / 64;
  constexpr int nchunk = P * B / 4;
  const int64_t ntiles = npix / P;
  unsigned long long tl = 0, tw = 0, ts = 0, tb2 = 0, tc = 0, tb3 = 0, ntile = 0;
  for (int64_t tileidx = blockIdx.x; tileidx < ntiles; tileidx += gridDim.x) {
    const int64_t pix0 = tileidx * P;
    const char* srcb = (const char*)(cube + pix0 * B);
    const unsigned long long t0 = stamp();
    if (t < P) flags[t] = 0u;
    for (int c0 = wave * 64; c0 < nchunk; c0 += THREADS) {
      const int c = c0 + lane;
      if (c < nchunk)
        __builtin_amdgcn_global_load_lds((gptr_t)(srcb + (size_t)c * 16), (lptr_t)(smem + (size_t)c0 * 16), 16, 0, 0);
    }
    const unsigned long long t1 = stamp();
    __syncthreads();
    const unsigned long long t2 = stamp();
    {
      const float4* t4 = (const float4*)smem;
      for (int c = t; c < nchunk; c += THREADS) {
        const float4 v = t4[c];
        float z = v.x * 0.0f; z = fmaf(v.y, 0.0f, z); z = fmaf(v.z, 0.0f, z); z = fmaf(v.w, 0.0f, z);
        if (z != z) { const int e = c * 4; flags[(e + 0) / B] = 1u; flags[(e + 3) / B] = 1u; }
      }
    }
    const unsigned long long t3 = stamp();
    __syncthreads();
    const unsigned long long t4s = stamp();
    {
      constexpr int SUB = 64 / P;
      const int p = lane % P;
      const int grp = wave * SUB + lane / P;
      constexpr int NG = NW * SUB;
      const bool slow = flags[p] != 0u;
      const float* v = tile + p * B;
#pragma unroll
      for (int j = 0; j < (NBMAX + NG - 1) / NG; ++j) {
        const int b = grp + NG * j;
        if (b < nb) {
          const float* w = wn + (size_t)b * B;
          float acc = 0.f;
          if (!slow) {
            const int k0 = bands.k0[b], n4 = bands.klen[b] >> 2;
            const float4* w4 = (const float4*)(wl + bands.woff[b]);
            const float* vs = v + k0;
            for (int i = 0; i < n4; ++i) {
              const float4 ww = w4[i];
              acc = fmaf(ww.x, vs[4 * i], acc); acc = fmaf(ww.y, vs[4 * i + 1], acc);
              acc = fmaf(ww.z, vs[4 * i + 2], acc); acc = fmaf(ww.w, vs[4 * i + 3], acc);
            }
          } else {
            for (int k = 0; k < B; ++k) acc = fmaf(w[k], v[k], acc);
          }
          planes[b * stride + pix0 + p] = acc;
        }
      }
    }
    const unsigned long long t5 = stamp();
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    const unsigned long long t6 = stamp();
    tl += t1 - t0; tw += t2 - t1; ts += t3 - t2; tb2 += t4s - t3; tc += t5 - t4s; tb3 += t6 - t5; ++ntile;
  }
  if (lane == 0) {
    unsigned long long* o = tout + ((size_t)blockIdx.x * NW + wave) * 8;
    o[0] = tl; o[1] = tw; o[2] = ts; o[3] = tb2; o[4] = tc; o[5] = tb3; o[6] = ntile;
  }
}


// ---------------------------------------------------------------- V_ring: double-buffered tiles, counted vmcnt
// One workgroup of T threads owns two 64-pixel LDS buffers; the DMA of tile k+1 is in flight while tile k
// is scanned and reduced.  vm ops per wave are issued in a fixed order (glds of tile k+1, then the plane
// stores of tile k), so "tile k landed" == all but the (stores + next glds) youngest ops are done.
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }
__device__ __forceinline__ void wait_vm_n(int n) {
  switch (n) {
    case 0: wait_vm<0>(); break; case 1: wait_vm<1>(); break; case 2: wait_vm<2>(); break; case 3: wait_vm<3>(); break;
    case 4: wait_vm<4>(); break; case 5: wait_vm<5>(); break; case 6: wait_vm<6>(); break; case 7: wait_vm<7>(); break;
    case 8: wait_vm<8>(); break; case 9: wait_vm<9>(); break; case 10: wait_vm<10>(); break; case 11: wait_vm<11>(); break;
    case 12: wait_vm<12>(); break; case 13: wait_vm<13>(); break; case 14: wait_vm<14>(); break; case 15: wait_vm<15>(); break;
    default: wait_vm<0>(); break;
  }
}
// LDS-DMA issued from inline asm: the compiler's waitcnt pass then does not know a DMA is pending and
// does not force vmcnt(0) in front of every ds_read; all vmcnt accounting for the DMA is done by hand.
__device__ __forceinline__ void glds16_asm(const void* gaddr, uint32_t lds_base) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off nt" ::"v"(gaddr), "s"(lds_base) : "memory");
}
template <int T, int MINW>
__global__ __launch_bounds__(T, MINW) void k_ring(const float* __restrict__ cube, int64_t npix, const float* __restrict__ wn,
                                                  Bands bands, int nb, float* __restrict__ planes, int64_t stride) {
  constexpr int P = 64, NW = T / 64, NG = T / P, SLOTS = (NBMAX + NG - 1) / NG;
  constexpr int nchunk = P * B / 4;
  constexpr int TILE_BYTES = P * B * 4;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint32_t* flags = (uint32_t*)(smem + 2 * TILE_BYTES);      // [2][64]
  float* wl = (float*)(flags + 128);
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  for (int b = 0; b < nb; ++b)
    for (int i = t; i < bands.klen[b]; i += T) wl[bands.woff[b] + i] = wn[(size_t)b * B + bands.k0[b] + i];
  if (t < 128) flags[t] = 0u;
  int bk0[SLOTS], bkl[SLOTS], bwo[SLOTS]; bool bval[SLOTS]; int nst = 0;
#pragma unroll
  for (int j = 0; j < SLOTS; ++j) {
    const int b = wave + NG * j; bval[j] = b < nb; const int bb = bval[j] ? b : 0;
    bk0[j] = bands.k0[bb]; bkl[j] = bval[j] ? bands.klen[bb] : 0; bwo[j] = bands.woff[bb]; nst += bval[j] ? 1 : 0;
  }
  nst = __builtin_amdgcn_readfirstlane(nst);
  // glds instructions this wave issues per tile (wave-instr index w, w+NW, ... < ceil(nchunk/64))
  constexpr int NINSTR = (nchunk + 63) / 64;
  const int nglds = (NINSTR - wave + NW - 1) / NW;
  const int64_t ntiles = npix / P;
  auto issue = [&](int64_t tileidx, int buf) {
    const char* srcb = (const char*)(cube + tileidx * P * B);
    const uint32_t dst = (uint32_t)(uintptr_t)(lptr_t)(smem) + (uint32_t)buf * TILE_BYTES;
    for (int c0 = wave * 64; c0 < nchunk; c0 += T) {
      const int c = c0 + lane;
      if (c < nchunk) glds16_asm(srcb + (size_t)c * 16, __builtin_amdgcn_readfirstlane(dst + (uint32_t)c0 * 16));
    }
  };
  int64_t tileidx = blockIdx.x;
  if (tileidx < ntiles) issue(tileidx, 0);
  int cur = 0;
  bool first = true;
  for (; tileidx < ntiles; tileidx += gridDim.x, cur ^= 1) {
    const int64_t nxt = tileidx + gridDim.x;
    const bool has_next = nxt < ntiles;
    if (has_next) issue(nxt, cur ^ 1);
    // ops younger than tile k's glds: previous tile's stores (if any) + next tile's glds (if issued)
    wait_vm_n((first ? 0 : nst) + (has_next ? nglds : 0));
    first = false;
    asm volatile("s_barrier" ::: "memory");
    const float* tile = (const float*)(smem + (size_t)cur * TILE_BYTES);
    uint32_t* fl = flags + 64 * cur;
    {
      const float4* t4 = (const float4*)tile;
      constexpr int U = (nchunk + T - 1) / T > 5 ? 5 : (nchunk + T - 1) / T;
      for (int c0 = t; c0 < nchunk; c0 += T * U) {
        float4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) { const int c = c0 + u * T; v[u] = t4[c < nchunk ? c : nchunk - 1]; }
        bool bad = false;
#pragma unroll
        for (int u = 0; u < U; ++u) { float z = v[u].x * 0.0f; z = fmaf(v[u].y, 0.0f, z); z = fmaf(v[u].z, 0.0f, z); z = fmaf(v[u].w, 0.0f, z); bad |= (z != z); }
        if (bad) {
#pragma unroll
          for (int u = 0; u < U; ++u) { const int c = c0 + u * T; const int e = (c < nchunk ? c : nchunk - 1) * 4;
            fl[(e + 0) / B] = 1u; fl[(e + 3) / B] = 1u; }
        }
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (t < 64) flags[64 * (cur ^ 1) + t] = 0u;     // for the next tile's scan (two barriers away)
    const bool slow = fl[lane] != 0u;
    const float* v = tile + lane * B;
    const int64_t pix0 = tileidx * P;
#pragma unroll
    for (int j = 0; j < SLOTS; ++j) {
      float acc = 0.f;
      const float* vs = v + bk0[j];
      const float4* w4 = (const float4*)(wl + bwo[j]);
      for (int i0 = 0; i0 < bkl[j]; i0 += 16) {
        float4 ww[4]; float xv[16];
#pragma unroll
        for (int u = 0; u < 4; ++u) ww[u] = w4[(i0 >> 2) + u];
#pragma unroll
        for (int u = 0; u < 16; ++u) xv[u] = vs[i0 + u];
#pragma unroll
        for (int u = 0; u < 4; ++u) { acc = fmaf(ww[u].x, xv[4*u], acc); acc = fmaf(ww[u].y, xv[4*u+1], acc); acc = fmaf(ww[u].z, xv[4*u+2], acc); acc = fmaf(ww[u].w, xv[4*u+3], acc); }
      }
      if (slow && bval[j]) { const float* w = wn + (size_t)(wave + NG * j) * B; acc = 0.f; for (int k = 0; k < B; ++k) acc = fmaf(w[k], v[k], acc); }
      if (bval[j]) __builtin_nontemporal_store(acc, &planes[(wave + NG * j) * stride + pix0 + lane]);
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  }
}

// ---------------------------------------------------------------- V_mfma: no LDS, dense f32 MFMA 16x16x4
// wave = 16 pixels per group; lane (i = l&15, kk = l>>4) loads 16 B at pixel i, elements 16s+4kk..+3.
// wperm[s][j][lane] = W[band = l&15][k = 16s + 4kk + j] (0 beyond B / nb)
constexpr int NS = (B + 15) / 16;   // 18

__device__ __forceinline__ void load_group(const float* __restrict__ cube, int64_t npix, int64_t g, int lane, f32x4 (&v)[NS]) {
  int64_t p = g * 16 + (lane & 15);
  if (p >= npix) p = npix - 1;
  const int kk = lane >> 4;
  const float* base = cube + p * B + 4 * kk;
#pragma unroll
  for (int s = 0; s < NS - 1; ++s) v[s] = *(const f4u*)(base + 16 * s);
  // last step: elements 16*(NS-1) + 4kk + j must stay inside the row (j < B - 272 - 4kk)
  constexpr int last = 16 * (NS - 1);
  f32x4 t = {0.f, 0.f, 0.f, 0.f};
  const int rem = B - last - 4 * kk;   // valid elements for this lane
  if (rem >= 4) t = *(const f4u*)(base + last);
  else {
    if (rem > 0) t.x = base[last];
    if (rem > 1) t.y = base[last + 1];
    if (rem > 2) t.z = base[last + 2];
  }
  v[NS - 1] = t;
}

template <bool WLDS, int MINW, bool PIPE>
__global__ __launch_bounds__(256, MINW) void k_mfma(const float* __restrict__ cube, int64_t npix, const float* __restrict__ wperm,
                                                    int nb, float* __restrict__ planes, int64_t stride) {
  __shared__ float wl[WLDS ? NS * 4 * 64 : 1];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  float wreg[WLDS ? 1 : NS][4];
  if (WLDS) {
    for (int i = t; i < NS * 4 * 64; i += 256) wl[i] = wperm[i];
    __syncthreads();
  } else {
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
      for (int j = 0; j < 4; ++j) wreg[s][j] = wperm[(s * 4 + j) * 64 + lane];
  }
  const int64_t ngroups = (npix + 15) / 16;
  const int64_t gstride = (int64_t)gridDim.x * 4;
  const int band = lane & 15;
  int64_t g = (int64_t)blockIdx.x * 4 + wave;
  f32x4 cur[NS], nxt[PIPE ? NS : 1];
  if (g < ngroups) load_group(cube, npix, g, lane, cur);
  for (; g < ngroups; g += gstride) {
    if (PIPE) { if (g + gstride < ngroups) load_group(cube, npix, g + gstride, lane, (f32x4(&)[NS])nxt); }
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < NS; ++s) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float bw = WLDS ? wl[(s * 4 + j) * 64 + lane] : wreg[s][j];
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(cur[s][j], bw, acc, 0, 0, 0);
      }
    }
    const int64_t p0 = g * 16 + 4 * (lane >> 4);
    if (band < nb) {
      if (p0 + 3 < npix) *(f32x4*)(planes + band * stride + p0) = acc;
      else for (int r = 0; r < 4; ++r) if (p0 + r < npix) planes[band * stride + p0 + r] = acc[r];
    }
    if (PIPE) {
#pragma unroll
      for (int s = 0; s < NS; ++s) cur[s] = ((f32x4(&)[NS])nxt)[s];
    } else {
      if (g + gstride < ngroups) load_group(cube, npix, g + gstride, lane, cur);
    }
  }
}

// ---------------------------------------------------------------- pure read probes
template <int UNROLL>
__global__ __launch_bounds__(256) void k_read(const float4* __restrict__ src, int64_t n16, float* __restrict__ sink) {
  float acc = 0.f;
  const int64_t stride = (int64_t)gridDim.x * 256;
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  for (; i + (UNROLL - 1) * stride < n16; i += UNROLL * stride) {
    float4 v[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) v[u] = src[i + u * stride];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) acc += v[u].x + v[u].y + v[u].z + v[u].w;
  }
  for (; i < n16; i += stride) { float4 a = src[i]; acc += a.x + a.y + a.z + a.w; }
  if (acc == 12345.678f) sink[0] = acc;
}

// ---------------------------------------------------------------- host
#include <functional>
struct Variant { std::string name; std::function<void()> run; std::vector<float> ms; bool check; };
#include <functional>

int main(int argc, char** argv) {
  const int H = argc > 1 ? atoi(argv[1]) : 1024, W = argc > 2 ? atoi(argv[2]) : 1024;
  const int rounds = argc > 3 ? atoi(argv[3]) : 7;
  const int64_t npix = (int64_t)H * W;
  const int nb = 12;
  printf("K1 lab: %d x %d x %d, nb=%d, cube %.1f MB\n", H, W, B, nb, npix * B * 4 / 1e6);
  // weights: Gaussian-ish supports similar to the S2 SRFs
  std::vector<float> wn((size_t)NBMAX * B, 0.f);
  Bands bands{};
  const int centres[12] = {8, 15, 24, 38, 44, 48, 54, 62, 65, 76, 166, 244};
  const int widths[12] = {6, 17, 10, 9, 5, 5, 6, 30, 6, 6, 24, 48};
  for (int b = 0; b < nb; ++b) {
    int k0 = std::max(0, centres[b] - widths[b] / 2), k1 = std::min(B, k0 + widths[b]);
    double sum = 0;
    for (int k = k0; k < k1; ++k) { double d = (k - centres[b]) / (widths[b] / 4.0 + 0.5); wn[b * B + k] = (float)exp(-0.5 * d * d); sum += wn[b * B + k]; }
    for (int k = k0; k < k1; ++k) wn[b * B + k] /= (float)sum;
    bands.k0[b] = k0; bands.klen[b] = k1 - k0;
  }
  Bands raw = bands;   // exact supports for the library call
  { int total = 0;
    for (int b = 0; b < nb; ++b) { int n16 = (bands.klen[b] + 15) / 16; int s0 = bands.k0[b]; if (s0 + 16 * n16 > B) s0 = B - 16 * n16;
      bands.k0[b] = s0; bands.klen[b] = 16 * n16; bands.woff[b] = total; total += 16 * n16; }
    bands.wtaps = total; printf("weight taps (padded): %d\n", total); }
  std::vector<float> wperm((size_t)NS * 4 * 64, 0.f);
  for (int s = 0; s < NS; ++s) for (int j = 0; j < 4; ++j) for (int l = 0; l < 64; ++l) {
    int n = l & 15, k = 16 * s + 4 * (l >> 4) + j;
    wperm[(s * 4 + j) * 64 + l] = (n < nb && k < B) ? wn[n * B + k] : 0.f;
  }
  float *d_cube, *d_wn, *d_wperm, *d_planes, *d_ref, *d_sink;
  CK(hipMalloc(&d_cube, npix * B * 4)); CK(hipMalloc(&d_wn, wn.size() * 4)); CK(hipMalloc(&d_wperm, wperm.size() * 4));
  CK(hipMalloc(&d_planes, (size_t)NBMAX * npix * 4)); CK(hipMalloc(&d_ref, (size_t)NBMAX * npix * 4)); CK(hipMalloc(&d_sink, 256));
  {
    std::vector<float> h((size_t)npix * B);
    uint32_t s = 12345;
    for (auto& x : h) { s = s * 1664525u + 1013904223u; x = (s >> 8) * (0.6f / 16777216.f); }
    h[5 * B + 100] = NAN; h[77 * B + 10] = INFINITY; h[(npix - 1) * B + 284] = NAN;
    CK(hipMemcpy(d_cube, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  }
  CK(hipMemcpy(d_wn, wn.data(), wn.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_wperm, wperm.data(), wperm.size() * 4, hipMemcpyHostToDevice));
  const int64_t stride = npix;

  auto lds_bytes = [&](int P) { return (size_t)P * B * 4 + 256 + (size_t)bands.wtaps * 4; };
#define SETLDS(kern, bytes) CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(bytes)))
  SETLDS((k_lds<64, 256, 2, 0>), lds_bytes(64)); SETLDS((k_lds<64, 256, 2, 1>), lds_bytes(64)); SETLDS((k_lds<64, 256, 2, 2>), lds_bytes(64));
  SETLDS((k_lds<32, 256, 4, 0>), lds_bytes(32)); SETLDS((k_lds<32, 256, 4, 1>), lds_bytes(32));
  SETLDS((k_lds<32, 128, 2, 0>), lds_bytes(32)); SETLDS((k_lds<32, 128, 2, 1>), lds_bytes(32));
  SETLDS((k_lds<64, 128, 1, 0>), lds_bytes(64));
  SETLDS((k_lds<16, 64, 2, 0>), lds_bytes(16)); SETLDS((k_lds<16, 64, 2, 1>), lds_bytes(16));
  SETLDS((k_lds<64, 256, 2, 0, true, 1, false>), lds_bytes(64)); SETLDS((k_lds<64, 256, 2, 0, true, 1, true>), lds_bytes(64)); SETLDS((k_lds<64, 256, 2, 0, true, 0, true>), lds_bytes(64));
  SETLDS((k_lds<64, 256, 2, 1, true, 1, false>), lds_bytes(64));
  SETLDS((k_lds<64, 256, 2, 0, true, 0, false, 2, false>), lds_bytes(64));
  SETLDS((k_lds<64, 256, 2, 0, true, 0, false, 16, false>), lds_bytes(64));
  SETLDS((k_lds<64, 256, 2, 0, true, 0, false, 1, false>), lds_bytes(64));
  SETLDS((k_lds<64, 256, 2, 0, true, 0, true, 2, false>), lds_bytes(64));
  SETLDS((k_lds<64, 256, 2, 0, true, 0, false, 17, false>), lds_bytes(64));
  SETLDS((k_lds<64, 256, 2, 0, true, 0, false, 0, true>), lds_bytes(64));
  SETLDS((k_lds<64, 256, 2, 1, true, 0, false, 2, false>), lds_bytes(64));
  SETLDS((k_lds<64, 256, 2, 3>), lds_bytes(64)); SETLDS((k_lds<64, 256, 2, 4>), lds_bytes(64));
  SETLDS((k_lds<64, 256, 2, 0, true>), lds_bytes(64)); SETLDS((k_lds<32, 256, 4, 0, true>), lds_bytes(32));
  SETLDS((k_lds<32, 128, 2, 0, true>), lds_bytes(32)); SETLDS((k_lds<16, 64, 2, 0, true>), lds_bytes(16));

  std::vector<Variant> vs;
  auto add = [&](const char* n, std::function<void()> f, bool check) { vs.push_back({n, f, {}, check}); };
  add("lds P64 T256 2/CU full (current)", [&] { hipLaunchKernelGGL((k_lds<64, 256, 2, 0>), dim3(512), dim3(256), lds_bytes(64), 0, d_cube, npix, d_wn, bands, nb, d_planes, stride); }, true);
  add("lds P64 T256 2/CU full RAWBAR", [&] { hipLaunchKernelGGL((k_lds<64, 256, 2, 0, true>), dim3(512), dim3(256), lds_bytes(64), 0, d_cube, npix, d_wn, bands, nb, d_planes, stride); }, true);
  add("lds P32 T256 4/CU full RAWBAR", [&] { hipLaunchKernelGGL((k_lds<32, 256, 4, 0, true>), dim3(1024), dim3(256), lds_bytes(32), 0, d_cube, npix, d_wn, bands, nb, d_planes, stride); }, true);
  add("lds P32 T128 4/CU full RAWBAR", [&] { hipLaunchKernelGGL((k_lds<32, 128, 2, 0, true>), dim3(1024), dim3(128), lds_bytes(32), 0, d_cube, npix, d_wn, bands, nb, d_planes, stride); }, true);
  add("lds P16 T64 8/CU full RAWBAR", [&] { hipLaunchKernelGGL((k_lds<16, 64, 2, 0, true>), dim3(2048), dim3(64), lds_bytes(16), 0, d_cube, npix, d_wn, bands, nb, d_planes, stride); }, true);
  add("lds P64 T256 2/CU full CONTIG", [&] { hipLaunchKernelGGL((k_lds<64, 256, 2, 0, true, 1, false>), dim3(512), dim3(256), lds_bytes(64), 0, d_cube, npix, d_wn, bands, nb, d_planes, stride); }, true);
  add("lds P64 T256 2/CU full CONTIG+NT", [&] { hipLaunchKernelGGL((k_lds<64, 256, 2, 0, true, 1, true>), dim3(512), dim3(256), lds_bytes(64), 0, d_cube, npix, d_wn, bands, nb, d_planes, stride); }, true);
  add("lds P64 T256 2/CU full strided+NT", [&] { hipLaunchKernelGGL((k_lds<64, 256, 2, 0, true, 0, true>), dim3(512), dim3(256), lds_bytes(64), 0, d_cube, npix, d_wn, bands, nb, d_planes, stride); }, true);
  add("lds P64 T256 2/CU load-only CONTIG", [&] { hipLaunchKernelGGL((k_lds<64, 256, 2, 1, true, 1, false>), dim3(512), dim3(256), lds_bytes(64), 0, d_cube, npix, d_wn, bands, nb, d_planes, stride); }, false);
  add("lds P64 T256 full aux nt(2)", [&] { hipLaunchKernelGGL((k_lds<64, 256, 2, 0, true, 0, false, 2, false>), dim3(512), dim3(256), lds_bytes(64), 0, d_cube, npix, d_wn, bands, nb, d_planes, stride); }, false);
  add("lds P64 T256 full aux sc1(16)", [&] { hipLaunchKernelGGL((k_lds<64, 256, 2, 0, true, 0, false, 16, false>), dim3(512), dim3(256), lds_bytes(64), 0, d_cube, npix, d_wn, bands, nb, d_planes, stride); }, false);
  add("lds P64 T256 full aux sc0(1)", [&] { hipLaunchKernelGGL((k_lds<64, 256, 2, 0, true, 0, false, 1, false>), dim3(512), dim3(256), lds_bytes(64), 0, d_cube, npix, d_wn, bands, nb, d_planes, stride); }, false);
  add("lds P64 T256 full aux nt + NT stores", [&] { hipLaunchKernelGGL((k_lds<64, 256, 2, 0, true, 0, true, 2, false>), dim3(512), dim3(256), lds_bytes(64), 0, d_cube, npix, d_wn, bands, nb, d_planes, stride); }, false);
  add("lds P64 T256 full aux sc0sc1(17)", [&] { hipLaunchKernelGGL((k_lds<64, 256, 2, 0, true, 0, false, 17, false>), dim3(512), dim3(256), lds_bytes(64), 0, d_cube, npix, d_wn, bands, nb, d_planes, stride); }, false);
  add("lds P64 T256 full stores to 64KB/plane region", [&] { hipLaunchKernelGGL((k_lds<64, 256, 2, 0, true, 0, false, 0, true>), dim3(512), dim3(256), lds_bytes(64), 0, d_cube, npix, d_wn, bands, nb, d_planes, stride); }, false);
  add("lds P64 T256 load-only aux nt(2)", [&] { hipLaunchKernelGGL((k_lds<64, 256, 2, 1, true, 0, false, 2, false>), dim3(512), dim3(256), lds_bytes(64), 0, d_cube, npix, d_wn, bands, nb, d_planes, stride); }, false);
  add("lds P64 T256 2/CU compute NO stores", [&] { hipLaunchKernelGGL((k_lds<64, 256, 2, 3>), dim3(512), dim3(256), lds_bytes(64), 0, d_cube, npix, d_wn, bands, nb, d_planes, stride); }, false);
  add("lds P64 T256 2/CU load-only + stores", [&] { hipLaunchKernelGGL((k_lds<64, 256, 2, 4>), dim3(512), dim3(256), lds_bytes(64), 0, d_cube, npix, d_wn, bands, nb, d_planes, stride); }, false);
  add("lds P64 T256 2/CU load-only", [&] { hipLaunchKernelGGL((k_lds<64, 256, 2, 1>), dim3(512), dim3(256), lds_bytes(64), 0, d_cube, npix, d_wn, bands, nb, d_planes, stride); }, false);
  add("lds P64 T256 2/CU load+scan", [&] { hipLaunchKernelGGL((k_lds<64, 256, 2, 2>), dim3(512), dim3(256), lds_bytes(64), 0, d_cube, npix, d_wn, bands, nb, d_planes, stride); }, false);
  add("lds P64 T128 2/CU full", [&] { hipLaunchKernelGGL((k_lds<64, 128, 1, 0>), dim3(512), dim3(128), lds_bytes(64), 0, d_cube, npix, d_wn, bands, nb, d_planes, stride); }, true);
  add("lds P32 T256 4/CU full", [&] { hipLaunchKernelGGL((k_lds<32, 256, 4, 0>), dim3(1024), dim3(256), lds_bytes(32), 0, d_cube, npix, d_wn, bands, nb, d_planes, stride); }, true);
  add("lds P32 T256 4/CU load-only", [&] { hipLaunchKernelGGL((k_lds<32, 256, 4, 1>), dim3(1024), dim3(256), lds_bytes(32), 0, d_cube, npix, d_wn, bands, nb, d_planes, stride); }, false);
  add("lds P32 T128 4/CU full", [&] { hipLaunchKernelGGL((k_lds<32, 128, 2, 0>), dim3(1024), dim3(128), lds_bytes(32), 0, d_cube, npix, d_wn, bands, nb, d_planes, stride); }, true);
  add("lds P32 T128 4/CU load-only", [&] { hipLaunchKernelGGL((k_lds<32, 128, 2, 1>), dim3(1024), dim3(128), lds_bytes(32), 0, d_cube, npix, d_wn, bands, nb, d_planes, stride); }, false);
  add("lds P16 T64 8/CU full", [&] { hipLaunchKernelGGL((k_lds<16, 64, 2, 0>), dim3(2048), dim3(64), lds_bytes(16), 0, d_cube, npix, d_wn, bands, nb, d_planes, stride); }, true);
  add("lds P16 T64 8/CU load-only", [&] { hipLaunchKernelGGL((k_lds<16, 64, 2, 1>), dim3(2048), dim3(64), lds_bytes(16), 0, d_cube, npix, d_wn, bands, nb, d_planes, stride); }, false);
  add("mfma wreg 2w/SIMD nopipe g512", [&] { hipLaunchKernelGGL((k_mfma<false, 2, false>), dim3(512), dim3(256), 0, 0, d_cube, npix, d_wperm, nb, d_planes, stride); }, true);
  add("mfma wreg 2w/SIMD pipe g512", [&] { hipLaunchKernelGGL((k_mfma<false, 2, true>), dim3(512), dim3(256), 0, 0, d_cube, npix, d_wperm, nb, d_planes, stride); }, true);
  add("mfma wlds 4w/SIMD nopipe g1024", [&] { hipLaunchKernelGGL((k_mfma<true, 4, false>), dim3(1024), dim3(256), 0, 0, d_cube, npix, d_wperm, nb, d_planes, stride); }, true);
  add("mfma wlds 3w/SIMD nopipe g768", [&] { hipLaunchKernelGGL((k_mfma<true, 3, false>), dim3(768), dim3(256), 0, 0, d_cube, npix, d_wperm, nb, d_planes, stride); }, true);
  add("mfma wlds 2w/SIMD pipe g512", [&] { hipLaunchKernelGGL((k_mfma<true, 2, true>), dim3(512), dim3(256), 0, 0, d_cube, npix, d_wperm, nb, d_planes, stride); }, true);
  add("mfma wlds 4w/SIMD nopipe g4096", [&] { hipLaunchKernelGGL((k_mfma<true, 4, false>), dim3(4096), dim3(256), 0, 0, d_cube, npix, d_wperm, nb, d_planes, stride); }, true);
  float* d_real; double* d_part; int slots = 0;
  CK(hipMalloc(&d_real, (size_t)NBMAX * npix * 4)); CK(hipMalloc(&d_part, hsr_partials_bytes(nb, 4)));
  CK(hipMemcpy(d_real, d_cube, (size_t)nb * npix * 4, hipMemcpyDeviceToDevice));
  auto ring_bytes = [&]() { return (size_t)2 * 64 * B * 4 + 512 + (size_t)bands.wtaps * 4; };
  SETLDS((k_ring<512, 2>), ring_bytes()); SETLDS((k_ring<1024, 4>), ring_bytes());
  add("ring P64x2 T512 1/CU g256", [&] { hipLaunchKernelGGL((k_ring<512, 2>), dim3(256), dim3(512), ring_bytes(), 0, d_cube, npix, d_wn, bands, nb, d_planes, stride); }, true);
  add("ring P64x2 T1024 1/CU g256", [&] { hipLaunchKernelGGL((k_ring<1024, 4>), dim3(256), dim3(1024), ring_bytes(), 0, d_cube, npix, d_wn, bands, nb, d_planes, stride); }, true);
  add("LIB K1 PIXMAJOR out", [&] { if (hsr_srf_integrate(d_cube, npix, B, d_wn, raw.k0, raw.klen, nb, d_planes, 1, 12, nullptr, 0)) { printf("%s\n", hsr_last_error()); exit(1);} }, false);
  add("LIB K1+K2 deg3 PIXMAJOR in/out", [&] { if (hsr_srf_integrate_moments(d_cube, npix, B, d_wn, raw.k0, raw.klen, nb, d_planes, 1, 12, d_real, 1, 12, nullptr, 0.f, 0.f, 3, d_part, &slots, nullptr, 0)) { printf("%s\n", hsr_last_error()); exit(1);} }, false);
  add("LIB K1+K2 deg3 PIXMAJOR out, PLANAR real", [&] { if (hsr_srf_integrate_moments(d_cube, npix, B, d_wn, raw.k0, raw.klen, nb, d_planes, 1, 12, d_real, npix, 1, nullptr, 0.f, 0.f, 3, d_part, &slots, nullptr, 0)) { printf("%s\n", hsr_last_error()); exit(1);} }, false);
  add("LIB hsr_srf_integrate (K1)", [&] { if (hsr_srf_integrate(d_cube, npix, B, d_wn, raw.k0, raw.klen, nb, d_planes, stride, 1, nullptr, 0)) { printf("%s\n", hsr_last_error()); exit(1);} }, true);
  for (int deg = 1; deg <= 4; ++deg) {
    static char names[5][64]; snprintf(names[deg], 64, "LIB hsr_srf_integrate_moments deg%d", deg);
    add(names[deg], [&, deg] { if (hsr_srf_integrate_moments(d_cube, npix, B, d_wn, raw.k0, raw.klen, nb, d_planes, stride, 1, d_real, npix, 1, nullptr, 0.f, 0.f, deg, d_part, &slots, nullptr, 0)) { printf("%s\n", hsr_last_error()); exit(1);} }, true);
  }
  const int64_t n16 = npix * B / 4;
  add("read probe unroll4 g2048", [&] { hipLaunchKernelGGL((k_read<4>), dim3(2048), dim3(256), 0, 0, (const float4*)d_cube, n16, d_sink); }, false);
  add("read probe unroll8 g2048", [&] { hipLaunchKernelGGL((k_read<8>), dim3(2048), dim3(256), 0, 0, (const float4*)d_cube, n16, d_sink); }, false);
  add("read probe unroll8 g4096", [&] { hipLaunchKernelGGL((k_read<8>), dim3(4096), dim3(256), 0, 0, (const float4*)d_cube, n16, d_sink); }, false);
  add("read probe unroll16 g1024", [&] { hipLaunchKernelGGL((k_read<16>), dim3(1024), dim3(256), 0, 0, (const float4*)d_cube, n16, d_sink); }, false);

  // reference output = variant 0
  vs[0].run(); CK(hipDeviceSynchronize());
  CK(hipMemcpy(d_ref, d_planes, (size_t)NBMAX * npix * 4, hipMemcpyDeviceToDevice));
  std::vector<float> href((size_t)nb * npix), hout((size_t)nb * npix);
  CK(hipMemcpy(href.data(), d_ref, href.size() * 4, hipMemcpyDeviceToHost));
  for (auto& v : vs) {
    if (!v.check) continue;
    CK(hipMemset(d_planes, 0xff, (size_t)NBMAX * npix * 4));
    v.run(); CK(hipDeviceSynchronize()); CK(hipGetLastError());
    CK(hipMemcpy(hout.data(), d_planes, hout.size() * 4, hipMemcpyDeviceToHost));
    double maxrel = 0; int64_t nanmis = 0;
    for (size_t i = 0; i < hout.size(); ++i) {
      const float a = hout[i], r = href[i];
      if (std::isnan(a) != std::isnan(r) || std::isinf(a) != std::isinf(r)) { ++nanmis; continue; }
      if (std::isfinite(r)) maxrel = std::max(maxrel, (double)fabsf(a - r) / (fabs(r) + 1e-6));
    }
    printf("check %-36s maxrel %.2e nan/inf mismatches %lld\n", v.name.c_str(), maxrel, (long long)nanmis);
  }
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int inner = 5;
  for (int r = 0; r < rounds + 1; ++r) {
    for (auto& v : vs) {
      CK(hipEventRecord(e0, 0));
      for (int i = 0; i < inner; ++i) v.run();
      CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (r > 0) v.ms.push_back(ms / inner);
    }
  }
  const double bytes = (double)npix * B * 4;
  printf("%-38s %9s %9s %9s\n", "variant", "med ms", "min ms", "GB/s(med)");
  for (auto& v : vs) {
    std::sort(v.ms.begin(), v.ms.end());
    const float med = v.ms[v.ms.size() / 2], mn = v.ms[0];
    printf("%-38s %9.4f %9.4f %9.1f\n", v.name.c_str(), med, mn, bytes / (med * 1e-3) / 1e9);
  }
  {
    unsigned long long* d_t; const int G = 512, NW = 4;
    CK(hipMalloc(&d_t, (size_t)G * NW * 8 * 8)); CK(hipMemset(d_t, 0, (size_t)G * NW * 8 * 8));
    SETLDS((k_lds_timed<64, 256, 2>), lds_bytes(64));
    for (int rep = 0; rep < 3; ++rep)
      hipLaunchKernelGGL((k_lds_timed<64, 256, 2>), dim3(G), dim3(256), lds_bytes(64), 0, d_cube, npix, d_wn, bands, nb, d_planes, stride, d_t);
    CK(hipDeviceSynchronize());
    std::vector<unsigned long long> ht((size_t)G * NW * 8);
    CK(hipMemcpy(ht.data(), d_t, ht.size() * 8, hipMemcpyDeviceToHost));
    double sum[6] = {0}; double nt = 0;
    for (int i = 0; i < G * NW; ++i) { for (int k = 0; k < 6; ++k) sum[k] += ht[i * 8 + k]; nt += ht[i * 8 + 6]; }
    const char* nm[6] = {"issue glds", "wait tile (sync1)", "scan", "sync2", "compute+store", "end barrier"};
    double tot = 0; for (int k = 0; k < 6; ++k) tot += sum[k];
    printf("phase stamps (P64 T256, cycles per tile per wave, 100 MHz-independent shader clocks):\n");
    for (int k = 0; k < 6; ++k) printf("  %-20s %9.0f  %5.1f %%\n", nm[k], sum[k] / nt, 100.0 * sum[k] / tot);
    printf("  total %9.0f cycles/tile\n", tot / nt);
  }
  return 0;
}
