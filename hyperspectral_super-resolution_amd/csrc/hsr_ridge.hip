// K4 (variant a9): multivariate polynomial-ridge fusion S2 (n_in bands) -> EMIT (T bands) on gfx950.
//
// Reference: legacy_notebooks/Spectral_matching.ipynb - Pipeline(StandardScaler, PolynomialFeatures(3,
// include_bias=False), Ridge(alpha=1)) fitted on logit(EMIT) at 60 m (raw lines 475-490) and applied at
// 10 m by predict_cube_logit (raw lines 192-213) through sigmoid(clip(z, +-50)).
// This is the only stage of the path that is a genuine dense contraction, hence the only one on MFMA:
//   fit      G = P^T [P | Y]  with P = [1 | 285 monomials of the standardised inputs], float64,
//            v_mfma_f64_16x16x4_f64 (contraction over pixels), fixed-order reduction of the pixel chunks;
//   predict  out[T][pixels] = W^T Phi^T with the 285 features of each 64-pixel tile expanded ON CHIP into
//            LDS (never written to HBM: 1.2 GB per Mpixel otherwise), v_mfma_f32_32x32x2_f32 (exact
//            float32 fma chain), epilogue intercept + clip + sigmoid, coalesced band-major stores.
// Feature order = sklearn's: degree-major, combinations_with_replacement (x0..x9, x0^2, x0x1, ..., x9^3).
#include <vector>

#include "hsr_common.h"

namespace hsr {

constexpr int kMaxIn = 16;        // input bands
constexpr int kMaxFeat = 1024;    // monomials (10 inputs, degree 3 -> 285)

typedef double f64x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

struct FeatTable {                // monomial f = z[a] * z[b] * z[c]; index n_in means the constant 1
  uint8_t idx[kMaxFeat][3];
};
static FeatTable g_table_host;
static int g_table_nin = -1, g_table_deg = -1, g_table_nfeat = 0;
static uint8_t* g_table_dev = nullptr;   // [nfeat][4] bytes (a, b, c, pad) - allocated once, outside launches

static int build_table(int n_in, int degree) {
  int f = 0;
  for (int d = 1; d <= degree; ++d) {
    int c[3] = {0, 0, 0};   // non-decreasing index tuple of length d
    while (true) {
      if (f >= kMaxFeat) return -1;
      for (int k = 0; k < 3; ++k) g_table_host.idx[f][k] = (uint8_t)(k < d ? c[k] : n_in);
      ++f;
      int pos = d - 1;
      while (pos >= 0 && c[pos] == n_in - 1) --pos;
      if (pos < 0) break;
      const int v = c[pos] + 1;
      for (int k = pos; k < d; ++k) c[k] = v;
    }
  }
  return f;
}

// ------------------------------------------------------------------------------------------------
// expand: X (N, n_in) float32 rows -> P (N, ldp) float64 = [1 | monomials of (x - mean)/scale | 0 pad]
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void expand_f64_kernel(const float* __restrict__ x, int64_t x_rs, int64_t x_cs,
                                                         const double* __restrict__ mean,
                                                         const double* __restrict__ scale, int64_t n, int n_in,
                                                         int nfeat, const uint8_t* __restrict__ table,
                                                         double* __restrict__ P, int64_t ldp, int ncols) {
  __shared__ double z[32][kMaxIn + 1];
  const int rows_per_block = 32;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
  for (int i = threadIdx.x; i < rows_per_block * (n_in + 1); i += 256) {
    const int r = i / (n_in + 1), c = i % (n_in + 1);
    double v = 1.0;
    if (c < n_in && r0 + r < n) v = ((double)x[(r0 + r) * x_rs + c * x_cs] - mean[c]) / scale[c];
    z[r][c] = v;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < rows_per_block * ncols; i += 256) {
    const int r = i / ncols, c = i % ncols;
    if (r0 + r >= n) continue;
    double v = 0.0;
    if (c == 0) v = 1.0;
    else if (c <= nfeat) {
      const uint8_t* t = table + (size_t)(c - 1) * 4;
      v = z[r][t[0]] * z[r][t[1]] * z[r][t[2]];
    }
    P[(r0 + r) * ldp + c] = v;
  }
}

// ------------------------------------------------------------------------------------------------
// gram: C (na x nb) = A^T B over n rows, float64 MFMA 16x16x4, one wave per 16x16 tile and row chunk
// ------------------------------------------------------------------------------------------------
// A (n, lda), B (n, ldb) row-major float64 with na, nb multiples of 16 inside lda/ldb.  Grid:
// (tiles_i * tiles_j, chunks).  v_mfma_f64_16x16x4_f64: lane l supplies A[i = l&15][k = l>>4] and
// B[k = l>>4][j = l&15]; D[row = (l>>4) + 4*reg][col = l&15] (the f64 map, NOT the f32 one).
__global__ __launch_bounds__(64) void gram_f64_kernel(const double* __restrict__ A, int64_t lda, int tiles_i,
                                                      const double* __restrict__ B, int64_t ldb, int tiles_j,
                                                      int64_t n, int64_t rows_per_chunk,
                                                      double* __restrict__ partials) {
  // One wave = a 32 x 32 output block (2 x 2 MFMA tiles): two A and two B operands per k-step feed four
  // MFMAs, half the operand loads per MFMA of the one-tile-per-wave version.
  const int bj = (tiles_j + 1) / 2;
  const int blk = blockIdx.x;
  const int ti0 = (blk / bj) * 2, tj0 = (blk % bj) * 2;
  const bool i1 = ti0 + 1 < tiles_i, j1 = tj0 + 1 < tiles_j;
  const int lane = threadIdx.x;
  const int col = lane & 15, kk = lane >> 4;
  const int64_t r0 = (int64_t)blockIdx.y * rows_per_chunk;
  int64_t r1 = r0 + rows_per_chunk;
  if (r1 > n) r1 = n;
  f64x4 acc[2][2];
#pragma unroll
  for (int x = 0; x < 2; ++x)
#pragma unroll
    for (int y = 0; y < 2; ++y) acc[x][y] = f64x4{0.0, 0.0, 0.0, 0.0};
  const double* ap = A + ti0 * 16 + col;
  const double* bp = B + tj0 * 16 + col;
  const int ao = i1 ? 16 : 0, bo = j1 ? 16 : 0;   // second tile aliases the first when it does not exist
  constexpr int KU = 8;   // k-steps whose operand loads are in flight together (32 independent loads)
  for (int64_t r = r0; r < r1; r += 4 * KU) {
    double a0[KU], a1[KU], b0[KU], b1[KU];
#pragma unroll
    for (int u = 0; u < KU; ++u) {
      const int64_t rr = r + 4 * u + kk;
      const bool ok = rr < r1;                    // ragged tail: rows beyond r1 contribute zeros
      const int64_t rc = ok ? rr : r0;            // clamped address, value masked below
      const double va0 = ap[rc * lda], va1 = ap[rc * lda + ao], vb0 = bp[rc * ldb], vb1 = bp[rc * ldb + bo];
      a0[u] = ok ? va0 : 0.0;
      a1[u] = ok ? va1 : 0.0;
      b0[u] = ok ? vb0 : 0.0;
      b1[u] = ok ? vb1 : 0.0;
    }
#pragma unroll
    for (int u = 0; u < KU; ++u) {
      acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[u], b0[u], acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[u], b1[u], acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[u], b0[u], acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[u], b1[u], acc[1][1], 0, 0, 0);
    }
  }
  const int ntiles = tiles_i * tiles_j;
#pragma unroll
  for (int x = 0; x < 2; ++x) {
#pragma unroll
    for (int y = 0; y < 2; ++y) {
      if ((x == 1 && !i1) || (y == 1 && !j1)) continue;
      const int tile = (ti0 + x) * tiles_j + (tj0 + y);
      double* out = partials + ((size_t)blockIdx.y * ntiles + tile) * 256;
#pragma unroll
      for (int g = 0; g < 4; ++g) out[(kk + 4 * g) * 16 + col] = acc[x][y][g];
    }
  }
}

// chunks summed in index order -> C[(ti*16 + r) * ldc + tj*16 + c]
__global__ __launch_bounds__(256) void gram_reduce_kernel(const double* __restrict__ partials, int ntiles, int chunks,
                                                          int tiles_j, double* __restrict__ C, int64_t ldc) {
  const int tile = blockIdx.x, e = threadIdx.x;
  double s = 0.0;
  for (int c = 0; c < chunks; ++c) s += partials[((size_t)c * ntiles + tile) * 256 + e];
  const int ti = tile / tiles_j, tj = tile % tiles_j;
  C[(size_t)(ti * 16 + (e >> 4)) * ldc + tj * 16 + (e & 15)] = s;
}

// ------------------------------------------------------------------------------------------------
// predict: out[t][p] = act( sum_f W[f][t] * phi_f(z_p) + b[t] ),  fused expand + f32 MFMA + epilogue
// ------------------------------------------------------------------------------------------------
struct PredArgs {
  const float* x;      // inputs: element (pixel p, band c) at x[p * x_ps + c * x_cs]
  int64_t x_ps, x_cs;
  const float* mean;   // [n_in]
  const float* inv;    // [n_in] 1/scale
  int64_t npix;
  int32_t n_in, nfeat, kpad;   // kpad = nfeat rounded up to even
  const uint8_t* table;        // [nfeat][4]
  const float* W;              // (kpad, ldw) float32, rows >= nfeat are zero
  int64_t ldw;
  const float* bias;           // [T]
  int32_t T, ttiles;           // ttiles = ceil(T / 32)
  int32_t act;                 // 1: sigmoid(clip(z, +-50)); 0: raw
  float* out;                  // (T, out_stride)
  int64_t out_stride;
};

constexpr int kPredPix = 64;      // pixels per workgroup tile
constexpr int kPredThreads = 256;

template <int TT>   // number of 32-wide target tiles held by a wave (accumulators: TT * 16 VGPRs)
__global__ __launch_bounds__(kPredThreads) void predict_kernel(const PredArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int ldphi = a.kpad + 1;                      // odd row stride -> conflict-free column walks
  float* phi = reinterpret_cast<float*>(smem);       // [64][ldphi]
  float* zt = phi + kPredPix * ldphi;                // [64][n_in + 1]
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int nz = a.n_in + 1;
  for (int64_t tile = blockIdx.x; tile * kPredPix < a.npix; tile += gridDim.x) {
    const int64_t p0 = tile * kPredPix;
    // standardised inputs (+ the constant 1 at index n_in); non-finite inputs propagate to NaN outputs
    for (int i = t; i < kPredPix * nz; i += kPredThreads) {
      const int p = i / nz, c = i % nz;
      float v = 1.0f;
      if (c < a.n_in) v = (p0 + p < a.npix) ? (a.x[(p0 + p) * a.x_ps + c * a.x_cs] - a.mean[c]) * a.inv[c] : 0.0f;
      zt[p * nz + c] = v;
    }
    __syncthreads();
    for (int i = t; i < kPredPix * a.kpad; i += kPredThreads) {
      const int f = i / kPredPix, p = i % kPredPix;       // consecutive threads -> consecutive pixels
      float v = 0.0f;
      if (f < a.nfeat) {
        const uint8_t* tb = a.table + (size_t)f * 4;
        v = zt[p * nz + tb[0]] * zt[p * nz + tb[1]] * zt[p * nz + tb[2]];
      }
      phi[p * ldphi + f] = v;
    }
    __syncthreads();
    // wave w: pixel half (w & 1) * 32, target tiles (w >> 1), (w >> 1) + 2, ...
    const int ph = (wave & 1) * 32;
    const int j = lane & 31, kh = lane >> 5;
    for (int tt0 = wave >> 1; tt0 < a.ttiles; tt0 += 2 * TT) {
      f32x16 acc[TT];
#pragma unroll
      for (int q = 0; q < TT; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[q][r] = 0.0f;
      const float* brow = phi + (ph + j) * ldphi + kh;   // B[k][pixel j] = phi[pixel][k]
      for (int k = 0; k < a.kpad; k += 2) {
        const float bv = brow[k];
#pragma unroll
        for (int q = 0; q < TT; ++q) {
          const int tt = tt0 + 2 * q;
          const int tcol = tt * 32 + j;                  // A[i = target][k] = W[k][target]
          const float av = (tt < a.ttiles && tcol < a.T) ? a.W[(size_t)(k + kh) * a.ldw + tcol] : 0.0f;
          acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[q], 0, 0, 0);
        }
      }
      // D[row = target, col = pixel]: row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), col = lane & 31
#pragma unroll
      for (int q = 0; q < TT; ++q) {
        const int tt = tt0 + 2 * q;
        if (tt >= a.ttiles) continue;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int trg = tt * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
          const int64_t p = p0 + ph + j;
          if (trg < a.T && p < a.npix) {
            float v = acc[q][r] + a.bias[trg];
            if (a.act) {
              v = v < -50.0f ? -50.0f : (v > 50.0f ? 50.0f : v);   // NaN falls through, like np.clip
              v = 1.0f / (1.0f + __expf(-v));
            }
            a.out[(size_t)trg * a.out_stride + p] = v;
          }
        }
      }
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------------
// predict, specialised for the notebook's shape (10 inputs, degree 3 -> 285 monomials): MFMA-bound
// ------------------------------------------------------------------------------------------------
// The generic kernel above stages a 64 x 286 feature tile in LDS (73 KB -> one workgroup per CU) and fetches
// W from global memory per MFMA: 10-12 % of the f32 matrix peak.  Here every lane keeps the 10 standardised
// inputs of its pixel in registers and the monomials are generated by fully unrolled code from a
// compile-time table (2-5 VALU ops per MFMA, hidden under the 64-cycle v_mfma_f32_32x32x2_f32), W lives in
// LDS (whole when it fits, otherwise double-buffered 32-row chunks), and a workgroup covers 128 pixels with
// all four waves busy.  Targets sit on the M axis so each accumulator register is 32 consecutive pixels of
// one target: coalesced band-major stores.
struct Tab103 {
  uint8_t v[286][3];
};
constexpr Tab103 make_tab103() {
  Tab103 t{};
  int f = 0;
  for (int d = 1; d <= 3; ++d) {
    int c[3] = {0, 0, 0};
    while (true) {
      for (int k = 0; k < 3; ++k) t.v[f][k] = (uint8_t)(k < d ? c[k] : 10);
      ++f;
      int pos = d - 1;
      while (pos >= 0 && c[pos] == 9) --pos;
      if (pos < 0) break;
      const int nv = c[pos] + 1;
      for (int k = pos; k < d; ++k) c[k] = nv;
    }
  }
  t.v[285][0] = t.v[285][1] = t.v[285][2] = 10;   // padding monomial (its W row is zero)
  return t;
}
constexpr Tab103 kTab103 = make_tab103();
constexpr int kSteps103 = 143;      // 286 / 2
constexpr int kChunkRows = 32;      // W rows per LDS chunk (16 MFMA steps)

template <int S0, int S1, int TT>
__device__ __forceinline__ void mfma_steps103(const float (&z)[11], int kh, const float* __restrict__ wl, int ldwl,
                                              int row0, int j, f32x16 (&acc)[TT]) {
  // Hand-pipelined: the W operands of step s+1 are read from LDS before the MFMAs of step s, and a
  // scheduling barrier per step stops hipcc from hoisting all 286 monomial products ahead of the MFMA
  // chain (which cost 230-256 VGPRs and spills).  Per step: 2-4 v_mul + 1 v_cndmask + TT ds_read_b32 under
  // TT x 64 cycles of MFMA.
  float wc[TT], wn[TT];
  {
    const float* wr = wl + (2 * S0 + kh - row0) * ldwl + j;   // A[i = target][k] = W[k][target]
#pragma unroll
    for (int q = 0; q < TT; ++q) wc[q] = wr[q * 32];
  }
#pragma unroll
  for (int s = S0; s < S1; ++s) {
    const float p0 = z[kTab103.v[2 * s][0]] * z[kTab103.v[2 * s][1]] * z[kTab103.v[2 * s][2]];
    const float p1 = z[kTab103.v[2 * s + 1][0]] * z[kTab103.v[2 * s + 1][1]] * z[kTab103.v[2 * s + 1][2]];
    const float bv = kh ? p1 : p0;                            // B[k = 2s + kh][pixel j]
    if (s + 1 < S1) {
      const float* wr = wl + (2 * (s + 1) + kh - row0) * ldwl + j;
#pragma unroll
      for (int q = 0; q < TT; ++q) wn[q] = wr[q * 32];
    }
#pragma unroll
    for (int q = 0; q < TT; ++q) acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(wc[q], bv, acc[q], 0, 0, 0);
#pragma unroll
    for (int q = 0; q < TT; ++q) wc[q] = wn[q];
    __builtin_amdgcn_sched_barrier(0);
  }
}

template <int TT, bool WHOLE>
__global__ __launch_bounds__(256, 2) void predict103_kernel(const PredArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* wl = reinterpret_cast<float*>(smem);
  constexpr int Tp = TT * 32;                       // padded target count held by every wave
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int j = lane & 31, kh = lane >> 5;
  if (WHOLE) {                                      // W (286 x Tp) resident for the whole launch
    for (int i = t; i < 286 * Tp; i += 256) {
      const int r = i / Tp, c = i % Tp;
      wl[i] = c < a.T ? a.W[(size_t)r * a.ldw + c] : 0.0f;
    }
    __syncthreads();
  }
  for (int64_t tile = blockIdx.x; tile * 128 < a.npix; tile += gridDim.x) {
    const int64_t p = tile * 128 + wave * 32 + j;
    const int64_t pc = p < a.npix ? p : a.npix - 1;
    float z[11];
#pragma unroll
    for (int c = 0; c < 10; ++c) z[c] = (a.x[pc * a.x_ps + c * a.x_cs] - a.mean[c]) * a.inv[c];
    z[10] = 1.0f;
    f32x16 acc[TT];
#pragma unroll
    for (int q = 0; q < TT; ++q)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[q][r] = 0.0f;
    if (WHOLE) {
      mfma_steps103<0, kSteps103, TT>(z, kh, wl, Tp, 0, j, acc);
    } else {
      // 9 chunks of 32 W rows (the last has 30), double-buffered: chunk c+1 is copied while chunk c is used
      auto stage = [&](int c, int buf) {
        float* dst = wl + buf * kChunkRows * Tp;
        const int rows = c == 8 ? 286 - 8 * kChunkRows : kChunkRows;
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll 2
        for (int i = t; i < rows * Tp; i += 256) {
          const int r = i / Tp, col = i % Tp;
          dst[i] = col < a.T ? a.W[(size_t)(c * kChunkRows + r) * a.ldw + col] : 0.0f;
        }
        __builtin_amdgcn_sched_barrier(0);
      };
      stage(0, 0);
      __syncthreads();
#define HSR_CHUNK(C)                                                                                   \
  do {                                                                                                 \
    if (C < 8) stage(C + 1, (C + 1) & 1);                                                              \
    mfma_steps103<C * 16, (C == 8 ? kSteps103 : C * 16 + 16), TT>(z, kh, wl + (C & 1) * kChunkRows * Tp, Tp, \
                                                                  C * kChunkRows, j, acc);               \
    __syncthreads();                                                                                   \
  } while (0)
      HSR_CHUNK(0); HSR_CHUNK(1); HSR_CHUNK(2); HSR_CHUNK(3); HSR_CHUNK(4);
      HSR_CHUNK(5); HSR_CHUNK(6); HSR_CHUNK(7); HSR_CHUNK(8);
#undef HSR_CHUNK
    }
    if (p < a.npix) {
      // keep the epilogue's addressing inside the tile loop: hoisted out of it (LICM) the 64-bit offsets of
      // all 16*TT accumulator rows cost up to 288 VGPRs and spilled the accumulators
      int64_t ostride = a.out_stride;
      int tmax = a.T;
      asm volatile("" : "+s"(ostride), "+s"(tmax));
      float* orow = a.out + (size_t)(4 * kh) * ostride + p;
#pragma unroll
      for (int q = 0; q < TT; ++q) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int tu = q * 32 + (r & 3) + 8 * (r >> 2);      // wave-uniform part of the target index
          const int trg = tu + 4 * kh;
          if (trg < tmax) {
            float v = acc[q][r] + a.bias[trg];
            if (a.act) {
              v = v < -50.0f ? -50.0f : (v > 50.0f ? 50.0f : v);
              v = 1.0f / (1.0f + __expf(-v));
            }
            orow[(size_t)tu * ostride] = v;
          }
        }
      }
    }
  }
}

template <int TT, bool WHOLE>
static void launch_predict103(const PredArgs& a, hipStream_t s) {
  const size_t lds = WHOLE ? (size_t)286 * TT * 32 * 4 : (size_t)2 * kChunkRows * TT * 32 * 4;
  static thread_local bool configured = false;
  if (!configured) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(predict103_kernel<TT, WHOLE>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipGetLastError();
    configured = true;
  }
  int64_t tiles = (a.npix + 127) / 128;
  const int grid = (int)(tiles < 512 ? tiles : 512);
  hipLaunchKernelGGL((predict103_kernel<TT, WHOLE>), dim3(grid), dim3(256), lds, s, a);
}

// returns false when the shape is not covered (caller falls back to the generic kernel)
static bool try_predict103(const PredArgs& a, hipStream_t s) {
  if (a.n_in != 10 || a.nfeat != 285) return false;
  const int tt = a.ttiles;
  if (tt == 1) launch_predict103<1, true>(a, s);          // T <= 32: W resident (36.6 KB)
  else if (tt == 2) launch_predict103<2, false>(a, s);
  else if (tt <= 3) launch_predict103<3, false>(a, s);
  else if (tt <= 5) launch_predict103<5, false>(a, s);
  else if (tt <= 9) launch_predict103<9, false>(a, s);     // T <= 288 (EMIT's 285 bands)
  else return false;
  return true;
}

static int ensure_table(int n_in, int degree) {
  if (g_table_nin == n_in && g_table_deg == degree && g_table_dev) return g_table_nfeat;
  const int nf = build_table(n_in, degree);
  if (nf < 0) return -1;
  std::vector<uint8_t> packed((size_t)nf * 4, 0);
  for (int f = 0; f < nf; ++f)
    for (int k = 0; k < 3; ++k) packed[(size_t)f * 4 + k] = g_table_host.idx[f][k];
  if (g_table_dev) (void)hipFree(g_table_dev);
  if (hipMalloc(&g_table_dev, packed.size()) != hipSuccess) return -1;
  if (hipMemcpy(g_table_dev, packed.data(), packed.size(), hipMemcpyHostToDevice) != hipSuccess) return -1;
  g_table_nin = n_in;
  g_table_deg = degree;
  g_table_nfeat = nf;
  return nf;
}

}  // namespace hsr

using namespace hsr;

extern "C" int hsr_polyfeat_count(int32_t n_in, int32_t degree) {
  if (n_in < 1 || n_in > kMaxIn || degree < 1 || degree > 3) return -1;
  int64_t total = 0, c = 1;   // sum_{d=1..deg} C(n_in + d - 1, d)
  for (int d = 1; d <= degree; ++d) {
    c = c * (n_in + d - 1) / d;
    total += c;
  }
  return total <= kMaxFeat ? (int)total : -1;
}

extern "C" int hsr_polyfeat_table(int32_t n_in, int32_t degree, uint8_t* idx_out /* [nfeat][3] */) {
  const int nf = build_table(n_in, degree);
  HSR_REQUIRE(nf > 0 && idx_out, HSR_ERR_UNSUPPORTED, "hsr_polyfeat_table: n_in=%d degree=%d", n_in, degree);
  for (int f = 0; f < nf; ++f)
    for (int k = 0; k < 3; ++k) idx_out[f * 3 + k] = g_table_host.idx[f][k];
  g_table_nin = -1;   // host table was rebuilt: force the device copy to refresh on next use
  return HSR_OK;
}

// Not a launch-path call: uploads the monomial table once per (n_in, degree) (hipMalloc + copy).
extern "C" int hsr_polyfeat_prepare(int32_t n_in, int32_t degree) {
  HSR_REQUIRE(hsr_polyfeat_count(n_in, degree) > 0, HSR_ERR_UNSUPPORTED, "hsr_polyfeat_prepare: n_in=%d degree=%d",
              n_in, degree);
  HSR_REQUIRE(ensure_table(n_in, degree) > 0, HSR_ERR_HIP, "hsr_polyfeat_prepare: table upload failed");
  return HSR_OK;
}

extern "C" int hsr_polyfeat_expand_f64(const float* x_dev, int64_t x_rs, int64_t x_cs, const double* mean_dev,
                                       const double* scale_dev, int64_t n, int32_t n_in, int32_t degree,
                                       double* p_dev, int64_t ldp, int32_t ncols, hsr_stream_t stream) {
  HSR_REQUIRE(x_dev && mean_dev && scale_dev && p_dev && n > 0, HSR_ERR_INVALID, "hsr_polyfeat_expand_f64: bad argument");
  HSR_REQUIRE(g_table_nin == n_in && g_table_deg == degree && g_table_dev, HSR_ERR_INVALID,
              "hsr_polyfeat_expand_f64: call hsr_polyfeat_prepare(%d, %d) first", n_in, degree);
  HSR_REQUIRE(ncols >= g_table_nfeat + 1 && ldp >= ncols && ncols <= 4096, HSR_ERR_INVALID,
              "hsr_polyfeat_expand_f64: ncols=%d ldp=%lld (need ncols >= %d)", ncols, (long long)ldp, g_table_nfeat + 1);
  hipLaunchKernelGGL(expand_f64_kernel, dim3((unsigned)((n + 31) / 32)), dim3(256), 0, (hipStream_t)stream, x_dev, x_rs,
                     x_cs, mean_dev, scale_dev, n, n_in, g_table_nfeat, g_table_dev, p_dev, ldp, ncols);
  HSR_LAUNCH_CHECK("expand_f64_kernel");
  return HSR_OK;
}

extern "C" size_t hsr_gram_work_bytes(int32_t na, int32_t nb, int64_t n) {
  if (na < 16 || nb < 16 || n < 1) return 0;
  int64_t chunks = (n + 1023) / 1024;
  if (chunks > 256) chunks = 256;
  return (size_t)chunks * (na / 16) * (nb / 16) * 256 * sizeof(double);
}

extern "C" int hsr_gram_f64(const double* a_dev, int64_t lda, int32_t na, const double* b_dev, int64_t ldb,
                            int32_t nb, int64_t n, double* work_dev, double* c_dev, int64_t ldc,
                            hsr_stream_t stream) {
  HSR_REQUIRE(a_dev && b_dev && work_dev && c_dev && n > 0, HSR_ERR_INVALID, "hsr_gram_f64: bad argument");
  HSR_REQUIRE(na >= 16 && nb >= 16 && na % 16 == 0 && nb % 16 == 0 && lda >= na && ldb >= nb && ldc >= nb,
              HSR_ERR_INVALID, "hsr_gram_f64: na=%d nb=%d must be multiples of 16 inside the leading dimensions", na, nb);
  int64_t chunks = (n + 1023) / 1024;
  if (chunks > 256) chunks = 256;
  int64_t rows = (n + chunks - 1) / chunks;
  rows = (rows + 3) / 4 * 4;                       // whole k-steps inside a chunk
  chunks = (n + rows - 1) / rows;
  const int ti = na / 16, tj = nb / 16;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(gram_f64_kernel, dim3(((ti + 1) / 2) * ((tj + 1) / 2), (unsigned)chunks), dim3(64), 0, s, a_dev, lda,
                     ti, b_dev, ldb, tj, n, rows, work_dev);
  hipLaunchKernelGGL(gram_reduce_kernel, dim3(ti * tj), dim3(256), 0, s, work_dev, ti * tj, (int)chunks, tj, c_dev, ldc);
  HSR_LAUNCH_CHECK("gram_f64_kernel");
  return HSR_OK;
}

extern "C" int hsr_polyfeat_predict(const float* x_dev, int64_t x_ps, int64_t x_cs, const float* mean_dev,
                                    const float* inv_scale_dev, int64_t npix, int32_t n_in, int32_t degree,
                                    const float* w_dev, int64_t ldw, const float* bias_dev, int32_t T,
                                    int32_t activation, float* out_dev, int64_t out_stride, hsr_stream_t stream) {
  HSR_REQUIRE(x_dev && mean_dev && inv_scale_dev && w_dev && bias_dev && out_dev, HSR_ERR_INVALID,
              "hsr_polyfeat_predict: NULL pointer");
  HSR_REQUIRE(npix > 0 && T >= 1 && ldw >= T && out_stride >= npix, HSR_ERR_INVALID, "hsr_polyfeat_predict: bad shape");
  HSR_REQUIRE(g_table_nin == n_in && g_table_deg == degree && g_table_dev, HSR_ERR_INVALID,
              "hsr_polyfeat_predict: call hsr_polyfeat_prepare(%d, %d) first", n_in, degree);
  PredArgs a{};
  a.x = x_dev;
  a.x_ps = x_ps;
  a.x_cs = x_cs;
  a.mean = mean_dev;
  a.inv = inv_scale_dev;
  a.npix = npix;
  a.n_in = n_in;
  a.nfeat = g_table_nfeat;
  a.kpad = (g_table_nfeat + 1) & ~1;
  a.table = g_table_dev;
  a.W = w_dev;
  a.ldw = ldw;
  a.bias = bias_dev;
  a.T = T;
  a.ttiles = (T + 31) / 32;
  a.act = activation;
  a.out = out_dev;
  a.out_stride = out_stride;
  if (degree == 3 && try_predict103(a, (hipStream_t)stream)) {
    HSR_LAUNCH_CHECK("predict103_kernel");
    return HSR_OK;
  }
  const size_t lds = ((size_t)kPredPix * (a.kpad + 1) + (size_t)kPredPix * (n_in + 1)) * sizeof(float);
  HSR_REQUIRE(lds <= 150 * 1024, HSR_ERR_UNSUPPORTED, "hsr_polyfeat_predict: %zu bytes of LDS needed", lds);
  int64_t tiles = (npix + kPredPix - 1) / kPredPix;
  const int grid = (int)(tiles < 512 ? tiles : 512);
  hipStream_t s = (hipStream_t)stream;
  static thread_local size_t configured[3] = {0, 0, 0};
  const int per_wave = (a.ttiles + 1) / 2;     // target tiles a wave pair must cover
  const int variant = per_wave <= 1 ? 0 : (per_wave <= 2 ? 1 : 2);
#define HSR_PRED_LAUNCH(TT, slot)                                                                                   \
  do {                                                                                                              \
    if (lds > configured[slot]) {                                                                                   \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(predict_kernel<TT>),                                  \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                              \
      (void)hipGetLastError();                                                                                      \
      configured[slot] = lds;                                                                                       \
    }                                                                                                               \
    hipLaunchKernelGGL(predict_kernel<TT>, dim3(grid), dim3(kPredThreads), lds, s, a);                              \
  } while (0)
  if (variant == 0) HSR_PRED_LAUNCH(1, 0);
  else if (variant == 1) HSR_PRED_LAUNCH(2, 1);
  else HSR_PRED_LAUNCH(4, 2);
#undef HSR_PRED_LAUNCH
  HSR_LAUNCH_CHECK("predict_kernel");
  return HSR_OK;
}
