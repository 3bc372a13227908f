// K2 (Vandermonde moments of materialised planes), the fixed-order slot reduction, the polynomial
// solve (np.polyfit semantics from moments) and K3 (polynomial apply) on gfx950.
//
// Reference: fit step np.polyfit per channel (s2_emit/poly_regression.py:59-60; all-pixel flavour
// Pairs_EMIT_S2_demo-2.ipynb cell 72), apply step apply_poly_rgb (poly_regression.py:65-84),
// stretch expression s2_emit/color.py:33.  All three kernels are streaming / HBM-bound; the only
// arithmetic of note is float64 (power sums, Horner), which the CDNA4 vector ALU does at half rate.
#include <math.h>
#include <string.h>

#include "hsr_common.h"
#include "hsr_solve.h"

namespace hsr {

// ------------------------------------------------------------------------------------------------
// K2: moments of planes
// ------------------------------------------------------------------------------------------------
struct MomArgs {
  const float* x;
  int64_t x_bs, x_ps;
  const float* y;
  int64_t y_bs, y_ps;
  const uint8_t* mask;
  int64_t npix;
  float min_x, min_y;
  const double* lohi_x;
  const double* lohi_y;
  double* partials;
  int32_t slots;
};

template <int DEG, typename T>
__device__ __forceinline__ void moment_add(double (&acc)[moment_count(DEG)], T xf, T yf) {
  const double xd = (double)xf, yd = (double)yf;
  acc[0] += 1.0;
  acc[2 * DEG + 1] += yd;
  double pw = 1.0;
#pragma unroll
  for (int k = 1; k <= 2 * DEG; ++k) {
    pw *= xd;
    acc[k] += pw;
    if (k <= DEG) acc[2 * DEG + 1 + k] += pw * yd;
  }
}

template <int DEG>
__global__ __launch_bounds__(256) void moments_kernel(const MomArgs a) {
  constexpr int M = moment_count(DEG);
  __shared__ double red[4][M];
  const int b = blockIdx.y;
  const float* x = a.x + (size_t)b * a.x_bs;
  const float* y = a.y + (size_t)b * a.y_bs;
  const bool sx = a.lohi_x != nullptr, sy = a.lohi_y != nullptr;
  const double lox = sx ? a.lohi_x[2 * b] : 0.0, hix = sx ? a.lohi_x[2 * b + 1] : 0.0;
  const double loy = sy ? a.lohi_y[2 * b] : 0.0, hiy = sy ? a.lohi_y[2 * b + 1] : 0.0;

  double acc[M];
#pragma unroll
  for (int m = 0; m < M; ++m) acc[m] = 0.0;

  // contiguous pixel range per slot: the summation tree depends on (npix, slots) only
  const int64_t per = (a.npix + a.slots - 1) / a.slots;
  const int64_t beg = (int64_t)blockIdx.x * per;
  int64_t end = beg + per;
  if (end > a.npix) end = a.npix;
  for (int64_t p = beg + threadIdx.x; p < end; p += 256) {
    float xv = ld_stream(x + p * a.x_ps), yv = ld_stream(y + p * a.y_ps);
    const bool m = a.mask ? a.mask[p] != 0 : true;
    const bool ok = m && finite_f32(xv) && finite_f32(yv) && xv > a.min_x && yv > a.min_y;
    if (ok) {
      if (sx) xv = stretch_f64(xv, lox, hix);
      if (sy) yv = stretch_f64(yv, loy, hiy);
      moment_add<DEG, float>(acc, xv, yv);
    }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int m = 0; m < M; ++m) {
    const double s = wave_sum(acc[m]);
    if (lane == 0) red[wave][m] = s;
  }
  __syncthreads();
  if (threadIdx.x < M) {
    const int m = threadIdx.x;
    const double s = ((red[0][m] + red[1][m]) + red[2][m]) + red[3][m];
    a.partials[((size_t)blockIdx.x * gridDim.y + b) * M + m] = s;     // slot-major: [slot][band][moment]
  }
}

// float64 sample columns (no mask, no stretch): the (X, Ybar) columns of fit_ot_poly_rgb
template <int DEG>
__global__ __launch_bounds__(256) void moments_f64_kernel(const double* __restrict__ xb, int64_t xs,
                                                          const double* __restrict__ yb, int64_t ys, int64_t npix,
                                                          double* __restrict__ partials, int slots) {
  constexpr int M = moment_count(DEG);
  __shared__ double red[4][M];
  const int b = blockIdx.y;
  const double* x = xb + (size_t)b * xs;
  const double* y = yb + (size_t)b * ys;
  double acc[M];
#pragma unroll
  for (int m = 0; m < M; ++m) acc[m] = 0.0;
  const int64_t per = (npix + slots - 1) / slots;
  const int64_t beg = (int64_t)blockIdx.x * per;
  int64_t end = beg + per;
  if (end > npix) end = npix;
  for (int64_t p = beg + threadIdx.x; p < end; p += 256) {
    const double xv = x[p], yv = y[p];
    if (isfinite(xv) && isfinite(yv)) moment_add<DEG, double>(acc, xv, yv);
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int m = 0; m < M; ++m) {
    const double s = wave_sum(acc[m]);
    if (lane == 0) red[wave][m] = s;
  }
  __syncthreads();
  if (threadIdx.x < M) {
    const int m = threadIdx.x;
    partials[((size_t)blockIdx.x * gridDim.y + b) * M + m] = ((red[0][m] + red[1][m]) + red[2][m]) + red[3][m];
  }
}

// one wave per (band, moment) row; lane-strided partial sums then the fixed butterfly
// Lane-strided sum of one slot row in a fixed order (slot = lane, lane+64, ...), the loads issued in
// independent batches of 8 so that the row costs one memory round trip instead of slots/64.
// The partials are slot-major, [slot][band][moment] (a K1 work unit writes its 2*M*... sums as whole cache lines; the
// first layout, [band][moment][slot], made every unit of a batch scatter 8-byte stores over nb*M lines shared with
// units running on other XCDs): row = address of (slot 0, band, moment), stride = nb * M doubles between slots.
__device__ __forceinline__ double row_sum(const double* __restrict__ row, int slots, int stride, int lane) {
  double s = 0.0;
  for (int i0 = lane; i0 < slots; i0 += 64 * 8) {
    double v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = i0 + 64 * u;
      v[u] = i < slots ? row[(size_t)i * stride] : 0.0;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) s += v[u];   // adding +0.0 for absent slots does not change the sum
  }
  return wave_sum(s);
}

// At most 16 workgroups of 4 waves: in the pipelined multi-GPU step this kernel runs on a side stream while the
// persistent K1 of the next tile owns the chip, and only workgroups that fit the few CUs K1 leaves free get
// dispatched before K1 ends (measured: 132 one-wave workgroups starved for the whole 0.22 ms with 4 free CUs,
// 16 workgroups run in 8 us).
__global__ __launch_bounds__(256) void reduce_kernel(const double* __restrict__ partials, int slots, int nrows,
                                                     double* __restrict__ moments) {
  const int lane = threadIdx.x & 63;
  for (int row = blockIdx.x * 4 + (threadIdx.x >> 6); row < nrows; row += gridDim.x * 4) {
    const double s = row_sum(partials + row, slots, nrows, lane);
    if (lane == 0) moments[row] = s;
  }
}

__global__ __launch_bounds__(64) void solve_kernel(const double* __restrict__ moments, int nb, int deg,
                                                   long long min_count, double* __restrict__ coeffs) {
  const int b = threadIdx.x;
  if (b < nb) solve_band(moments + (size_t)b * moment_count(deg), deg, min_count, coeffs + (size_t)b * (deg + 1));
}

// The 63 adds of wave_sum's xor butterfly (32, 16, .., 1) over 64 lane sums, done by one thread: t[l] = v[l] + v[l ^ off]
// for l < off is the value every lane of the pair holds after a level, so a[0] after the last level is wave_sum's result
// bit for bit.  ``col`` = address of lane 0's sum, ``ld`` doubles between lanes.
__device__ __forceinline__ double butterfly_tree64(const double* col, int ld) {
  double a[32];
#pragma unroll
  for (int l = 0; l < 32; ++l) a[l] = col[(size_t)l * ld] + col[(size_t)(l + 32) * ld];
#pragma unroll
  for (int off = 16; off >= 1; off >>= 1)
#pragma unroll
    for (int l = 0; l < off; ++l) a[l] = a[l] + a[l + off];
  return a[0];
}

// reduce + solve in one launch: one workgroup per band.  Same tree as row_sum / reduce_kernel - "lane" l adds slots
// l, l + 64, ... in order, then the butterfly over the 64 lane sums - but laid out for the slot-major partials: a
// 16-lane group reads the band's 3deg+2 moments of ONE slot (112 contiguous bytes for deg 3), so a load instruction
// touches 4-8 cache lines instead of 64 (the wave-per-moment gather cost ~8000 line requests through one CU's L1,
// 4 us of the kernel's 8.6).  The lane sums go through LDS; thread m < M walks the butterfly for moment m.
__global__ __launch_bounds__(1024) void reduce_solve_kernel(const double* __restrict__ partials, int slots, int deg,
                                                            long long min_count, double* __restrict__ moments,
                                                            double* __restrict__ coeffs) {
  __shared__ double lsum[64][16];
  __shared__ double mom[16];
  __shared__ double work[kSolveWork];
  const int M = moment_count(deg);
  const int b = blockIdx.x;
  const int stride = (int)gridDim.x * M;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l = wave * 4 + (lane >> 4), m = lane & 15;
  {
    const double* row = partials + (size_t)b * M + (m < M ? m : 0);
    double s = 0.0;
    for (int i0 = l; i0 < slots; i0 += 64 * 8) {
      double v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int i = i0 + 64 * u;
        v[u] = (i < slots && m < M) ? row[(size_t)i * stride] : 0.0;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) s += v[u];   // adding +0.0 for absent slots does not change the sum
    }
    lsum[l][m] = s;
  }
  __syncthreads();
  if (threadIdx.x < M) {
    const double s = butterfly_tree64(&lsum[0][threadIdx.x], 16);
    mom[threadIdx.x] = s;
    moments[(size_t)b * M + threadIdx.x] = s;
  }
  __syncthreads();
  if (threadIdx.x == 0) solve_band_work(mom, deg, min_count, coeffs + (size_t)b * (deg + 1), work);
}

// Batch form (hsr_moments_reduce_solve_batched): ONE workgroup per tile, the tile's slot block at slot0 * nb * M of the
// batch workspace with its own slot count.  Same tree as row_sum -> same bits as the single-tile launch on that tile -
// laid out for the slot-major partials like reduce_solve_kernel: a 16-lane group reads the M moments of one (slot,
// band) - contiguous - and every thread carries the lane sums of all bands, so the whole tile costs ~2 memory round
// trips.  Lane sum index l = (lane / 16) * 16 + wave: the butterfly's first two levels (l ^ 32, l ^ 16) are in-wave
// shuffles, the last four (over the wave index) run from LDS, one thread per (band, moment) row; then one thread per
// band solves.  (First version: a (band, tile) grid with thread 0 solving, 49 us for 256 tiles; second: one wave per
// row gathering 8 bytes per cache line, 14.2 us; this one 13.5 us, 11.1 us of it without the solve: 256 tiles of
// 100 x 100 pixels hold 157 slots x 1344 B each = 54 MB of partials, so the kernel is an HBM read at ~5 TB/s.)
// NBU: bands carried per thread (>= nb; 8, 12 or 16); TWO: two slots per pass (the registers allow it up to 12 bands).
template <int NBU, bool TWO>
__global__ __launch_bounds__(1024) void reduce_solve_batched_kernel(const hsr_batch_tile* __restrict__ tiles,
                                                                    const double* __restrict__ partials, int nb, int deg,
                                                                    long long min_count, double* __restrict__ moments,
                                                                    double* __restrict__ coeffs) {
  constexpr int RMAX = HSR_MAX_BANDS * (3 * HSR_MAX_DEG + 2);
  __shared__ double lsum[16][RMAX];
  __shared__ double mom[RMAX];
  __shared__ double work[HSR_MAX_BANDS * kSolveWork];   // Jacobi matrices of the rank-deficient bands (no scratch)
  const int M = moment_count(deg);
  const int R = nb * M;
  const int tile = blockIdx.x;
  const int64_t slot0 = tiles[tile].slot0;
  const int slots = tiles[tile].slots;
  const double* part = partials + (size_t)slot0 * R;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int q = lane >> 4, m = lane & 15;
  const int l = q * 16 + wave;
  const bool act = m < M;
  double s[NBU];
#pragma unroll
  for (int bnd = 0; bnd < NBU; ++bnd) s[bnd] = 0.0;
  for (int i0 = l; i0 < slots; i0 += TWO ? 128 : 64) {   // slots l, l + 64, ... in order
    double v0[NBU], v1[TWO ? NBU : 1];
    const double* p0 = part + (size_t)i0 * R + (act ? m : 0);
    const bool has1 = TWO && i0 + 64 < slots;
#pragma unroll
    for (int bnd = 0; bnd < NBU; ++bnd) {
      const bool on = act && bnd < nb;
      v0[bnd] = on ? p0[bnd * M] : 0.0;
      if (TWO) v1[bnd] = (on && has1) ? p0[(size_t)64 * R + bnd * M] : 0.0;
    }
#pragma unroll
    for (int bnd = 0; bnd < NBU; ++bnd) {
      s[bnd] += v0[bnd];
      if (TWO) s[bnd] += v1[bnd];                    // +0.0 for an absent slot does not change the sum
    }
  }
#pragma unroll
  for (int bnd = 0; bnd < NBU; ++bnd) {
    if (bnd < nb) {
      double t = s[bnd] + __shfl_xor(s[bnd], 32, 64);
      t = t + __shfl_xor(t, 16, 64);
      if (q == 0 && act) lsum[wave][bnd * M + m] = t;
    }
  }
  __syncthreads();
  if ((int)threadIdx.x < R) {
    double a[8];
#pragma unroll
    for (int w = 0; w < 8; ++w) a[w] = lsum[w][threadIdx.x] + lsum[w + 8][threadIdx.x];
#pragma unroll
    for (int off = 4; off >= 1; off >>= 1)
#pragma unroll
      for (int w = 0; w < off; ++w) a[w] = a[w] + a[w + off];
    mom[threadIdx.x] = a[0];
    moments[(size_t)tile * R + threadIdx.x] = a[0];
  }
  __syncthreads();
  if ((int)threadIdx.x < nb)
    solve_band_work(mom + threadIdx.x * M, deg, min_count, coeffs + ((size_t)tile * nb + threadIdx.x) * (deg + 1),
                    work + threadIdx.x * kSolveWork);
}

// ------------------------------------------------------------------------------------------------
// K3: polynomial apply
// ------------------------------------------------------------------------------------------------
struct ApplyArgs {
  const float* x;
  int64_t x_bs, x_ps;
  const uint8_t* mask;
  const double* coeffs;
  const double* lohi;
  int32_t nb, deg, clip;
  int64_t npix;
  float* out;
  int64_t out_bs, out_ps;
};

// np.polyval: y = 0; for c in coeffs: y = y*x + c  -- separate multiply and add in float64 (no FMA
// contraction) so the float32 store is bit-identical to the NumPy >= 2 result.
__device__ __forceinline__ float poly_eval(float xf, const double* c, int n) {
  const double x = (double)xf;
  double y = 0.0;
  for (int i = 0; i < n; ++i) y = __dadd_rn(__dmul_rn(y, x), c[i]);
  return (float)y;
}

__device__ __forceinline__ float clip01(float v) { return v < 0.0f ? 0.0f : (v > 1.0f ? 1.0f : v); }

// band-major planes, 16 bytes per lane along the pixel axis
template <bool VEC4>
__global__ __launch_bounds__(256) void apply_planar_kernel(const ApplyArgs a) {
  const int b = blockIdx.y;
  const int n = a.deg + 1;
  const bool has_poly = a.coeffs != nullptr;
  double c[HSR_MAX_APPLY_DEG + 1];
  for (int i = 0; i < n; ++i) c[i] = has_poly ? a.coeffs[(size_t)b * n + i] : 0.0;
  const bool st = a.lohi != nullptr;
  const double lo = st ? a.lohi[2 * b] : 0.0, hi = st ? a.lohi[2 * b + 1] : 0.0;
  const float* x = a.x + (size_t)b * a.x_bs;
  float* o = a.out + (size_t)b * a.out_bs;
  const int64_t tid = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t nthreads = (int64_t)gridDim.x * 256;
  if (VEC4) {
    const int64_t nv = a.npix >> 2;
    for (int64_t i = tid; i < nv; i += nthreads) {
      float4 v = ld_stream(reinterpret_cast<const float4*>(x) + i);
      uint32_t m = 0x01010101u;
      if (a.mask) m = reinterpret_cast<const uint32_t*>(a.mask)[i];
      float r[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float xv = r[j];
        if (st) xv = stretch_f64(xv, lo, hi);
        if (has_poly && ((m >> (8 * j)) & 0xffu)) xv = poly_eval(xv, c, n);
        r[j] = a.clip ? clip01(xv) : xv;
      }
      st_stream(reinterpret_cast<float4*>(o) + i, make_float4(r[0], r[1], r[2], r[3]));
    }
    for (int64_t p = (nv << 2) + tid; p < a.npix; p += nthreads) {
      float xv = x[p];
      if (st) xv = stretch_f64(xv, lo, hi);
      if (has_poly && (!a.mask || a.mask[p])) xv = poly_eval(xv, c, n);
      o[p] = a.clip ? clip01(xv) : xv;
    }
  } else {
    for (int64_t p = tid; p < a.npix; p += nthreads) {
      float xv = x[p * a.x_ps];
      if (st) xv = stretch_f64(xv, lo, hi);
      if (has_poly && (!a.mask || a.mask[p])) xv = poly_eval(xv, c, n);
      o[p * a.out_ps] = a.clip ? clip01(xv) : xv;
    }
  }
}

// pixel-major (band-last) images, the layout of the reference API (H, W, C) and of the fused path:
// a pure contiguous stream.  Rows of 4*Q floats, 16 bytes per lane (4 channels of one pixel); channels >= nb of a
// padded row pass through unchanged.
//
// apply_rows_kernel<Q, N>: the grid stride is a multiple of Q float4s, so a lane keeps its channel group
// (tid % Q) for the whole launch and holds the 4 x N coefficients of its channels (N = deg + 1 <= 5) and their
// stretch limits in registers.  The first version kept the table in LDS and read N doubles per element at
// cs + ch * n: lanes of a wave sit on Q channel groups whose rows collide in the banks - SQ_LDS_BANK_CONFLICT was
// half of SQ_LDS_IDX_ACTIVE (profiles/r01_rocprof_summary.md) and the kernel ran at 4.3 TB/s.
// BATCH: blockIdx.y = tile of a batch (hsr_poly_apply_batched), every tile with its own coefficients.
template <int Q, int N, bool BATCH>
__global__ __launch_bounds__(256) void apply_rows_kernel(const ApplyArgs a, const hsr_batch_tile* __restrict__ tiles,
                                                         int use_mask) {
#ifndef HSR_K3_U
#define HSR_K3_U 16
#endif
  // independent 16-byte loads in flight per thread.  A/B on one box at 1024 x 1024 x 12 (tools/dbg/k3_time.py):
  // U = 4: 25.0 us, 8: 22.1, 12: 22.7, 16: 19.0 (768 workgroups = 3 per CU in one round), 20: 21.7, 24: 24.0, 32: 23.2
  constexpr int U = HSR_K3_U;
  const bool has_poly = a.coeffs != nullptr;
  const bool st = a.lohi != nullptr;
  const float* x = a.x;
  float* out = a.out;
  const uint8_t* mask = a.mask;
  int64_t npix = a.npix;
  const double* coeffs = a.coeffs;
  if (BATCH) {
    const hsr_batch_tile tl = tiles[blockIdx.y];
    x = tl.pseudo_dev;
    out = tl.matched_dev;
    mask = (use_mask & 1) ? tl.mask_dev : nullptr;
    npix = tl.npix;
    coeffs = (use_mask & 2) ? a.coeffs : a.coeffs + (size_t)blockIdx.y * a.nb * N;   // bit 1: one polynomial set for all tiles
  }
  const uint32_t nv = (uint32_t)(npix * Q);          // host guarantees npix * Q < 2^31
  const uint32_t tid = blockIdx.x * 256u + threadIdx.x;
  const uint32_t nthreads = gridDim.x * 256u;        // host: a multiple of Q
  if (BATCH && tid >= nv) return;
  float4 v[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {          // first batch of loads goes out before the coefficient loads
    const uint32_t i = tid + u * nthreads;
    if (i < nv) v[u] = ld_stream(reinterpret_cast<const float4*>(x) + i);
  }
  const int c0 = (int)(tid % Q) * 4;     // first channel of this lane, fixed for the launch
  double c[4][N], lo[4], hi[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int ch = c0 + j < a.nb ? c0 + j : 0;
#pragma unroll
    for (int k = 0; k < N; ++k) c[j][k] = has_poly ? coeffs[ch * N + k] : 0.0;
    lo[j] = st ? a.lohi[2 * ch] : 0.0;
    hi[j] = st ? a.lohi[2 * ch + 1] : 0.0;
  }
  for (uint32_t i0 = tid; i0 < nv; i0 += nthreads * U) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const uint32_t i = i0 + u * nthreads;
      if (i >= nv) break;
      const uint32_t p = i / Q;
      const bool m = !mask || mask[p];
      float r[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (c0 + j < a.nb) {
          float xv = r[j];
          if (st) xv = stretch_f64(xv, lo[j], hi[j]);
          if (has_poly && m) {      // np.polyval: y = 0; y = y*x + c, separately rounded
            const double xd = (double)xv;
            double y = 0.0;
#pragma unroll
            for (int k = 0; k < N; ++k) y = __dadd_rn(__dmul_rn(y, xd), c[j][k]);
            xv = (float)y;
          }
          r[j] = a.clip ? clip01(xv) : xv;
        }
      }
      st_stream(reinterpret_cast<float4*>(out) + i, make_float4(r[0], r[1], r[2], r[3]));
    }
    // next batch (grid-stride; a single pass when the grid covers the image)
    const uint32_t inext = i0 + nthreads * U;
    if (inext < nv) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const uint32_t i = inext + u * nthreads;
        if (i < nv) v[u] = ld_stream(reinterpret_cast<const float4*>(x) + i);
      }
    }
  }
}

// General pixel-major fallback: any row stride, any degree up to HSR_MAX_APPLY_DEG, coefficient table in LDS.
__global__ __launch_bounds__(256) void apply_pixmajor_scalar_kernel(const ApplyArgs a) {
  __shared__ double cs[HSR_MAX_BANDS * (HSR_MAX_APPLY_DEG + 1)];
  __shared__ double lh[HSR_MAX_BANDS * 2];
  const int n = a.deg + 1;
  const bool has_poly = a.coeffs != nullptr;
  const bool st = a.lohi != nullptr;
  const int64_t tid = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t nthreads = (int64_t)gridDim.x * 256;
  if (has_poly)
    for (int i = threadIdx.x; i < a.nb * n; i += 256) cs[i] = a.coeffs[i];
  if (st)
    for (int i = threadIdx.x; i < a.nb * 2; i += 256) lh[i] = a.lohi[i];
  __syncthreads();
  const int64_t total = a.npix * a.nb;
  for (int64_t e = tid; e < total; e += nthreads) {
    const int64_t p = e / a.nb;
    const int ch = (int)(e - p * a.nb);
    float xv = a.x[p * a.x_ps + ch];
    if (st) xv = stretch_f64(xv, lh[2 * ch], lh[2 * ch + 1]);
    if (has_poly && (!a.mask || a.mask[p])) xv = poly_eval(xv, cs + ch * n, n);
    a.out[p * a.out_ps + ch] = a.clip ? clip01(xv) : xv;
  }
}

// rows of 4*Q floats with a degree above 4 (apply only; fits are <= 4): 16 bytes per lane, table in LDS
template <int Q>
__global__ __launch_bounds__(256) void apply_rows_lds_kernel(const ApplyArgs a) {
  __shared__ double cs[HSR_MAX_BANDS * (HSR_MAX_APPLY_DEG + 1)];
  __shared__ double lh[HSR_MAX_BANDS * 2];
  const int n = a.deg + 1;
  const bool has_poly = a.coeffs != nullptr;
  const bool st = a.lohi != nullptr;
  if (has_poly)
    for (int i = threadIdx.x; i < a.nb * n; i += 256) cs[i] = a.coeffs[i];
  if (st)
    for (int i = threadIdx.x; i < a.nb * 2; i += 256) lh[i] = a.lohi[i];
  __syncthreads();
  const uint32_t nv = (uint32_t)(a.npix * Q);
  for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < nv; i += gridDim.x * 256u) {
    const float4 v = ld_stream(reinterpret_cast<const float4*>(a.x) + i);
    const uint32_t p = i / Q;
    const int c0 = (int)(i - p * Q) * 4;
    const bool m = !a.mask || a.mask[p];
    float r[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int ch = c0 + j;
      if (ch < a.nb) {
        float xv = r[j];
        if (st) xv = stretch_f64(xv, lh[2 * ch], lh[2 * ch + 1]);
        if (has_poly && m) xv = poly_eval(xv, cs + ch * n, n);
        r[j] = a.clip ? clip01(xv) : xv;
      }
    }
    st_stream(reinterpret_cast<float4*>(a.out) + i, make_float4(r[0], r[1], r[2], r[3]));
  }
}

// ------------------------------------------------------------------------------------------------
// validity mask of the pipeline (poly_regression.py:106,118)
// ------------------------------------------------------------------------------------------------
// No short-circuit: every sample is loaded unconditionally (all addresses are valid), so the loads of a pixel are
// independent and in flight together; `a && load` made each load wait for the previous verdict (1.9 TB/s).
// Band-last rows of whole float4s are read 16 bytes at a time.
__device__ __forceinline__ uint32_t nonfinite_bits(float v) { return (__float_as_uint(v) & 0x7f800000u) == 0x7f800000u; }

__global__ __launch_bounds__(256) void valid_mask_kernel(const float* x, int64_t xbs, int64_t xps, int nbx,
                                                         int pos_band, const float* y, int64_t ybs, int64_t yps,
                                                         int nby, const uint8_t* min, int64_t npix, uint8_t* mout) {
  const bool xv = xbs == 1 && (xps & 3) == 0 && ((((uintptr_t)x) & 15) == 0);
  const bool yv = y && ybs == 1 && (yps & 3) == 0 && ((((uintptr_t)y) & 15) == 0);
  for (int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x; p < npix; p += (int64_t)gridDim.x * 256) {
    uint32_t bad = min ? (min[p] == 0) : 0u;
    float pos = 1.0f;
    if (xv) {
      const float4* r = reinterpret_cast<const float4*>(x + p * xps);
      for (int q = 0; q * 4 < nbx; ++q) {
        const float4 v = r[q];
        const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (q * 4 + j < nbx) bad |= nonfinite_bits(e[j]);
          if (q * 4 + j == pos_band) pos = e[j];
        }
      }
    } else {
      for (int b = 0; b < nbx; ++b) {
        const float v = x[b * xbs + p * xps];
        bad |= nonfinite_bits(v);
        if (b == pos_band) pos = v;
      }
    }
    if (pos_band >= 0) bad |= !(pos > 0.0f);
    if (yv) {
      const float4* r = reinterpret_cast<const float4*>(y + p * yps);
      for (int q = 0; q * 4 < nby; ++q) {
        const float4 v = r[q];
        const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (q * 4 + j < nby) bad |= nonfinite_bits(e[j]);
      }
    } else if (y) {
      for (int b = 0; b < nby; ++b) bad |= nonfinite_bits(y[b * ybs + p * yps]);
    }
    mout[p] = bad ? 0 : 1;
  }
}

static inline int stream_grid(int64_t work_items, int per_block) {
  int64_t g = (work_items + per_block - 1) / per_block;
  if (g < 1) g = 1;
  return (int)(g > 2048 ? 2048 : g);  // 256 CUs x 8 blocks, grid-stride beyond
}

template <bool BATCH, int Q>
static int launch_apply_rows_q(const ApplyArgs& a, const hsr_batch_tile* tiles, int use_mask, int deg, dim3 grid, hipStream_t s) {
  switch (deg) {
    case 0: hipLaunchKernelGGL((apply_rows_kernel<Q, 1, BATCH>), grid, dim3(256), 0, s, a, tiles, use_mask); break;
    case 1: hipLaunchKernelGGL((apply_rows_kernel<Q, 2, BATCH>), grid, dim3(256), 0, s, a, tiles, use_mask); break;
    case 2: hipLaunchKernelGGL((apply_rows_kernel<Q, 3, BATCH>), grid, dim3(256), 0, s, a, tiles, use_mask); break;
    case 3: hipLaunchKernelGGL((apply_rows_kernel<Q, 4, BATCH>), grid, dim3(256), 0, s, a, tiles, use_mask); break;
    case 4: hipLaunchKernelGGL((apply_rows_kernel<Q, 5, BATCH>), grid, dim3(256), 0, s, a, tiles, use_mask); break;
    default: set_error("apply_rows: deg=%d", deg); return HSR_ERR_UNSUPPORTED;
  }
  return HSR_OK;
}
template <bool BATCH>
static int launch_apply_rows(const ApplyArgs& a, const hsr_batch_tile* tiles, int use_mask, int q, int deg, dim3 grid, hipStream_t s) {
  switch (q) {
    case 1: return launch_apply_rows_q<BATCH, 1>(a, tiles, use_mask, deg, grid, s);
    case 2: return launch_apply_rows_q<BATCH, 2>(a, tiles, use_mask, deg, grid, s);
    case 3: return launch_apply_rows_q<BATCH, 3>(a, tiles, use_mask, deg, grid, s);
    case 4: return launch_apply_rows_q<BATCH, 4>(a, tiles, use_mask, deg, grid, s);
  }
  set_error("apply_rows: rows of %d floats", 4 * q);
  return HSR_ERR_UNSUPPORTED;
}

}  // namespace hsr

using namespace hsr;

static bool strides_ok(int64_t bs, int64_t ps, int64_t npix, int nb) {
  return (ps == 1 && bs >= npix) || (bs == 1 && ps >= nb);
}

extern "C" int hsr_poly_moments(const float* x_dev, int64_t x_bs, int64_t x_ps, const float* y_dev, int64_t y_bs,
                                int64_t y_ps, const uint8_t* mask_dev, int64_t npix, int32_t nb, int32_t deg, float min_x,
                                float min_y, const double* lohi_x_dev, const double* lohi_y_dev,
                                double* partials_dev, int32_t* slots_out, hsr_stream_t stream) {
  HSR_REQUIRE(x_dev && y_dev && partials_dev, HSR_ERR_INVALID, "hsr_poly_moments: NULL pointer");
  HSR_REQUIRE(npix > 0, HSR_ERR_INVALID, "hsr_poly_moments: npix must be > 0");
  HSR_REQUIRE(nb >= 1 && nb <= HSR_MAX_BANDS, HSR_ERR_UNSUPPORTED, "hsr_poly_moments: nb=%d outside [1,%d]", nb,
              HSR_MAX_BANDS);
  HSR_REQUIRE(deg >= 1 && deg <= HSR_MAX_DEG, HSR_ERR_UNSUPPORTED, "hsr_poly_moments: deg=%d outside [1,%d]", deg,
              HSR_MAX_DEG);
  HSR_REQUIRE(strides_ok(x_bs, x_ps, npix, nb) && strides_ok(y_bs, y_ps, npix, nb), HSR_ERR_INVALID,
              "hsr_poly_moments: strides are neither band-major nor pixel-major");
  MomArgs a{x_dev, x_bs, x_ps, y_dev, y_bs, y_ps, mask_dev, npix, min_x, min_y, lohi_x_dev, lohi_y_dev, partials_dev,
            partial_slots(npix)};
  const dim3 grid(a.slots, nb), block(256);
  hipStream_t s = (hipStream_t)stream;
  switch (deg) {
    case 1: hipLaunchKernelGGL(moments_kernel<1>, grid, block, 0, s, a); break;
    case 2: hipLaunchKernelGGL(moments_kernel<2>, grid, block, 0, s, a); break;
    case 3: hipLaunchKernelGGL(moments_kernel<3>, grid, block, 0, s, a); break;
    default: hipLaunchKernelGGL(moments_kernel<4>, grid, block, 0, s, a); break;
  }
  HSR_LAUNCH_CHECK("moments_kernel");
  if (slots_out) *slots_out = a.slots;
  return HSR_OK;
}

extern "C" int hsr_poly_moments_f64(const double* x_dev, int64_t x_stride, const double* y_dev, int64_t y_stride,
                                    int64_t npix, int32_t nb, int32_t deg, double* partials_dev,
                                    int32_t* slots_out, hsr_stream_t stream) {
  HSR_REQUIRE(x_dev && y_dev && partials_dev, HSR_ERR_INVALID, "hsr_poly_moments_f64: NULL pointer");
  HSR_REQUIRE(npix > 0 && x_stride >= npix && y_stride >= npix, HSR_ERR_INVALID, "hsr_poly_moments_f64: bad shape");
  HSR_REQUIRE(nb >= 1 && nb <= HSR_MAX_BANDS && deg >= 1 && deg <= HSR_MAX_DEG, HSR_ERR_UNSUPPORTED,
              "hsr_poly_moments_f64: nb=%d deg=%d", nb, deg);
  const int slots = partial_slots(npix);
  const dim3 grid(slots, nb), block(256);
  hipStream_t s = (hipStream_t)stream;
  switch (deg) {
    case 1: hipLaunchKernelGGL(moments_f64_kernel<1>, grid, block, 0, s, x_dev, x_stride, y_dev, y_stride, npix, partials_dev, slots); break;
    case 2: hipLaunchKernelGGL(moments_f64_kernel<2>, grid, block, 0, s, x_dev, x_stride, y_dev, y_stride, npix, partials_dev, slots); break;
    case 3: hipLaunchKernelGGL(moments_f64_kernel<3>, grid, block, 0, s, x_dev, x_stride, y_dev, y_stride, npix, partials_dev, slots); break;
    default: hipLaunchKernelGGL(moments_f64_kernel<4>, grid, block, 0, s, x_dev, x_stride, y_dev, y_stride, npix, partials_dev, slots); break;
  }
  HSR_LAUNCH_CHECK("moments_f64_kernel");
  if (slots_out) *slots_out = slots;
  return HSR_OK;
}

extern "C" int hsr_moments_reduce(const double* partials_dev, int32_t slots, int32_t nb, int32_t deg,
                                  double* moments_dev, hsr_stream_t stream) {
  HSR_REQUIRE(partials_dev && moments_dev, HSR_ERR_INVALID, "hsr_moments_reduce: NULL pointer");
  HSR_REQUIRE(slots >= 1 && slots <= HSR_MAX_PARTIALS, HSR_ERR_INVALID, "hsr_moments_reduce: slots=%d", slots);
  HSR_REQUIRE(nb >= 1 && nb <= HSR_MAX_BANDS && deg >= 1 && deg <= HSR_MAX_DEG, HSR_ERR_UNSUPPORTED,
              "hsr_moments_reduce: nb=%d deg=%d", nb, deg);
  const int nrows = nb * moment_count(deg);
  hipLaunchKernelGGL(reduce_kernel, dim3((nrows + 3) / 4 < 16 ? (nrows + 3) / 4 : 16), dim3(256), 0, (hipStream_t)stream, partials_dev,
                     slots, nrows, moments_dev);
  HSR_LAUNCH_CHECK("reduce_kernel");
  return HSR_OK;
}

extern "C" int hsr_poly_solve(const double* moments_dev, int32_t nb, int32_t deg, int64_t min_count,
                              double* coeffs_dev, hsr_stream_t stream) {
  HSR_REQUIRE(moments_dev && coeffs_dev, HSR_ERR_INVALID, "hsr_poly_solve: NULL pointer");
  HSR_REQUIRE(nb >= 1 && nb <= HSR_MAX_BANDS && deg >= 1 && deg <= HSR_MAX_DEG, HSR_ERR_UNSUPPORTED,
              "hsr_poly_solve: nb=%d deg=%d", nb, deg);
  hipLaunchKernelGGL(solve_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, moments_dev, nb, deg,
                     (long long)min_count, coeffs_dev);
  HSR_LAUNCH_CHECK("solve_kernel");
  return HSR_OK;
}

extern "C" int hsr_moments_reduce_solve(const double* partials_dev, int32_t slots, int32_t nb, int32_t deg,
                                        int64_t min_count, double* moments_dev, double* coeffs_dev,
                                        hsr_stream_t stream) {
  HSR_REQUIRE(partials_dev && moments_dev && coeffs_dev, HSR_ERR_INVALID, "hsr_moments_reduce_solve: NULL pointer");
  HSR_REQUIRE(slots >= 1 && slots <= HSR_MAX_PARTIALS, HSR_ERR_INVALID, "hsr_moments_reduce_solve: slots=%d", slots);
  HSR_REQUIRE(nb >= 1 && nb <= HSR_MAX_BANDS && deg >= 1 && deg <= HSR_MAX_DEG, HSR_ERR_UNSUPPORTED,
              "hsr_moments_reduce_solve: nb=%d deg=%d", nb, deg);
  hipLaunchKernelGGL(reduce_solve_kernel, dim3(nb), dim3(1024), 0, (hipStream_t)stream, partials_dev, slots, deg,
                     (long long)min_count, moments_dev, coeffs_dev);
  HSR_LAUNCH_CHECK("reduce_solve_kernel");
  return HSR_OK;
}

extern "C" int hsr_poly_solve_host(const double* moments, int32_t nb, int32_t deg, int64_t min_count,
                                   double* coeffs) {
  HSR_REQUIRE(moments && coeffs, HSR_ERR_INVALID, "hsr_poly_solve_host: NULL pointer");
  HSR_REQUIRE(nb >= 1 && deg >= 1 && deg <= HSR_MAX_DEG, HSR_ERR_UNSUPPORTED, "hsr_poly_solve_host: nb=%d deg=%d", nb,
              deg);
  for (int b = 0; b < nb; ++b)
    solve_band(moments + (size_t)b * moment_count(deg), deg, (long long)min_count, coeffs + (size_t)b * (deg + 1));
  return HSR_OK;
}

extern "C" int hsr_poly_apply(const float* x_dev, int64_t x_bs, int64_t x_ps, const uint8_t* mask_dev,
                              const double* coeffs_dev, int32_t nb, int32_t deg, int64_t npix,
                              const double* lohi_dev, int32_t clip, float* out_dev, int64_t out_bs, int64_t out_ps,
                              hsr_stream_t stream) {
  HSR_REQUIRE(x_dev && out_dev, HSR_ERR_INVALID, "hsr_poly_apply: NULL pointer");
  HSR_REQUIRE(npix >= 0, HSR_ERR_INVALID, "hsr_poly_apply: npix < 0");
  HSR_REQUIRE(nb >= 1 && nb <= HSR_MAX_BANDS && deg >= 0 && deg <= HSR_MAX_APPLY_DEG, HSR_ERR_UNSUPPORTED,
              "hsr_poly_apply: nb=%d deg=%d", nb, deg);
  HSR_REQUIRE(strides_ok(x_bs, x_ps, npix, nb) && strides_ok(out_bs, out_ps, npix, nb), HSR_ERR_INVALID,
              "hsr_poly_apply: strides are neither band-major nor pixel-major");
  if (npix == 0) return HSR_OK;
  ApplyArgs a{x_dev, x_bs, x_ps, mask_dev, coeffs_dev, lohi_dev, nb, deg, clip, npix, out_dev, out_bs, out_ps};
  hipStream_t s = (hipStream_t)stream;
  int rc = HSR_OK;
  const bool aligned = ((((uintptr_t)x_dev) | ((uintptr_t)out_dev)) & 15) == 0;
  if (x_bs == 1 && out_bs == 1 && x_ps == out_ps && !(x_ps == 1 && nb > 1)) {  // pixel-major in and out, same rows
    const int64_t q = x_ps >> 2;
    if (aligned && (x_ps & 3) == 0 && q >= 1 && q <= 4 && npix * q < ((int64_t)1 << 31)) {
      // one pass when it fits (4 x 16 B per thread): no grid-stride tail imbalance on a ~20 us kernel.  The
      // grid is a multiple of 3 workgroups so that the stride is a multiple of every Q (lane <-> channel group fixed).
      // Small images (a 128 x 1024 row block of a strong-scaling run: 393 216 float4) got 96 workgroups of 16 loads per
      // thread - 10.6 us for 12.6 MB (rocprofv3, r03): latency bound on a third of the CUs.  The kernel skips loads past the
      // image, so a larger grid simply means fewer loads per thread: aim for 3 workgroups per CU (768) and 1 .. U loads.
      const int64_t nv4 = npix * q;
      int64_t per_thread = (nv4 + 768 * 256 - 1) / (768 * 256);
      per_thread = per_thread < 1 ? 1 : (per_thread > HSR_K3_U ? HSR_K3_U : per_thread);
      int64_t gb = (nv4 + 256 * per_thread - 1) / (256 * per_thread);
      if (gb > 8190) gb = 2046;
      gb = (gb + 2) / 3 * 3;
      const dim3 grid((unsigned)gb);
      if (deg <= HSR_MAX_DEG) {
        rc = launch_apply_rows<false>(a, nullptr, 0, (int)q, deg, grid, s);
        if (rc != HSR_OK) return rc;
      } else {
        switch (q) {
          case 1: hipLaunchKernelGGL(apply_rows_lds_kernel<1>, grid, dim3(256), 0, s, a); break;
          case 2: hipLaunchKernelGGL(apply_rows_lds_kernel<2>, grid, dim3(256), 0, s, a); break;
          case 3: hipLaunchKernelGGL(apply_rows_lds_kernel<3>, grid, dim3(256), 0, s, a); break;
          default: hipLaunchKernelGGL(apply_rows_lds_kernel<4>, grid, dim3(256), 0, s, a); break;
        }
      }
    } else {
      hipLaunchKernelGGL(apply_pixmajor_scalar_kernel, dim3(stream_grid(npix * nb, 256 * 4)), dim3(256), 0, s, a);
    }
    HSR_LAUNCH_CHECK("apply_rows_kernel");
    return HSR_OK;
  }
  const bool vec = aligned && x_ps == 1 && out_ps == 1 && (x_bs & 3) == 0 && (out_bs & 3) == 0 &&
                   (((uintptr_t)mask_dev) & 3) == 0;
  const dim3 grid(stream_grid(npix, 256 * 4), nb);
  if (vec)
    hipLaunchKernelGGL(apply_planar_kernel<true>, grid, dim3(256), 0, s, a);
  else
    hipLaunchKernelGGL(apply_planar_kernel<false>, grid, dim3(256), 0, s, a);  // any mix of strides
  HSR_LAUNCH_CHECK("apply_planar_kernel");
  return HSR_OK;
}

extern "C" int hsr_valid_mask(const float* x_dev, int64_t x_bs, int64_t x_ps, int32_t nbx, int32_t pos_band,
                              const float* y_dev, int64_t y_bs, int64_t y_ps, int32_t nby,
                              const uint8_t* mask_in_dev, int64_t npix, uint8_t* mask_out_dev,
                              hsr_stream_t stream) {
  HSR_REQUIRE(x_dev && mask_out_dev, HSR_ERR_INVALID, "hsr_valid_mask: NULL pointer");
  HSR_REQUIRE(nbx >= 1 && pos_band < nbx && npix >= 0, HSR_ERR_INVALID, "hsr_valid_mask: bad shape");
  if (npix == 0) return HSR_OK;
  hipLaunchKernelGGL(valid_mask_kernel, dim3(stream_grid(npix, 256)), dim3(256), 0, (hipStream_t)stream, x_dev,
                     x_bs, x_ps, nbx, pos_band, y_dev, y_bs, y_ps, y_dev ? nby : 0, mask_in_dev, npix, mask_out_dev);
  HSR_LAUNCH_CHECK("valid_mask_kernel");
  return HSR_OK;
}

extern "C" int hsr_moments_reduce_solve_batched(const hsr_batch_tile* tiles_dev, int32_t ntiles, const double* partials_dev,
                                                int32_t nb, int32_t deg, int64_t min_count, double* moments_dev,
                                                double* coeffs_dev, hsr_stream_t stream) {
  HSR_REQUIRE(tiles_dev && partials_dev && moments_dev && coeffs_dev, HSR_ERR_INVALID, "hsr_moments_reduce_solve_batched: NULL pointer");
  HSR_REQUIRE(ntiles >= 1 && ntiles <= 65535, HSR_ERR_UNSUPPORTED, "hsr_moments_reduce_solve_batched: ntiles=%d outside [1,65535]", ntiles);
  HSR_REQUIRE(nb >= 1 && nb <= HSR_MAX_BANDS && deg >= 1 && deg <= HSR_MAX_DEG, HSR_ERR_UNSUPPORTED,
              "hsr_moments_reduce_solve_batched: nb=%d deg=%d", nb, deg);
  auto* kern = nb <= 8 ? reduce_solve_batched_kernel<8, true> : nb <= 12 ? reduce_solve_batched_kernel<12, true>
                                                                           : reduce_solve_batched_kernel<16, false>;
  hipLaunchKernelGGL(kern, dim3(ntiles), dim3(1024), 0, (hipStream_t)stream, tiles_dev, partials_dev, nb, deg,
                     (long long)min_count, moments_dev, coeffs_dev);
  HSR_LAUNCH_CHECK("reduce_solve_batched_kernel");
  return HSR_OK;
}

extern "C" int hsr_poly_apply_batched(const hsr_batch_tile* tiles_dev, int32_t ntiles, int64_t max_npix, const double* coeffs_dev,
                                      int32_t nb, int32_t deg, int32_t row, int32_t use_mask, int32_t clip, hsr_stream_t stream) {
  HSR_REQUIRE(tiles_dev && coeffs_dev, HSR_ERR_INVALID, "hsr_poly_apply_batched: NULL pointer");
  HSR_REQUIRE(ntiles >= 1 && ntiles <= 65535, HSR_ERR_UNSUPPORTED, "hsr_poly_apply_batched: ntiles=%d outside [1,65535]", ntiles);
  HSR_REQUIRE(nb >= 1 && nb <= HSR_MAX_BANDS && deg >= 0 && deg <= HSR_MAX_DEG, HSR_ERR_UNSUPPORTED,
              "hsr_poly_apply_batched: nb=%d deg=%d", nb, deg);
  HSR_REQUIRE(row >= nb && (row & 3) == 0 && row <= HSR_MAX_BANDS, HSR_ERR_UNSUPPORTED,
              "hsr_poly_apply_batched: row=%d must be a multiple of 4 in [nb,%d]", row, HSR_MAX_BANDS);
  const int64_t q = row >> 2;
  HSR_REQUIRE(max_npix >= 1 && max_npix * q < ((int64_t)1 << 31), HSR_ERR_UNSUPPORTED, "hsr_poly_apply_batched: max_npix=%lld", (long long)max_npix);
  ApplyArgs a{nullptr, 1, row, nullptr, coeffs_dev, nullptr, nb, deg, clip, 0, nullptr, 1, row};
  int64_t gb = (max_npix * q + 256 * HSR_K3_U - 1) / (256 * HSR_K3_U);
  if (gb > 2046) gb = 2046;
  gb = (gb + 2) / 3 * 3;
  int rc = launch_apply_rows<true>(a, tiles_dev, use_mask, (int)q, deg, dim3((unsigned)gb, (unsigned)ntiles), (hipStream_t)stream);
  if (rc != HSR_OK) return rc;
  HSR_LAUNCH_CHECK("apply_rows_kernel (batched)");
  return HSR_OK;
}
