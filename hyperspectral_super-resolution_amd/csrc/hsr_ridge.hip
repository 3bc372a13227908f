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
#include <mutex>
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
// One wave = a 48 x 48 output block (R x R = 3 x 3 MFMA tiles): three A and three B operands per k-step feed
// nine MFMAs.  The kernel is bound by operand traffic from L2, not by the matrix pipe (the 2 x 2 version:
// one operand load per MFMA, 50 % MFMA busy), so the lever is operands per MFMA: 0.67 here.  With `sym`
// (B's first tiles_i tile columns are A itself, the Gram of the fit) blocks strictly below the diagonal
// are skipped and mirrored by the reduction: 15 of the 36 symmetric blocks at 288 features.
constexpr int kGramR = 3;

__device__ __forceinline__ bool gram_block_skipped(int bi, int bj, int sym) { return sym && bj < bi; }

__global__ __launch_bounds__(256) void gram_f64_kernel(const double* __restrict__ A, int64_t lda, int tiles_i,
                                                       const double* __restrict__ B, int64_t ldb, int tiles_j,
                                                       int64_t n, int64_t rows_per_chunk, int sym,
                                                       double* __restrict__ partials) {
  constexpr int R = kGramR;
  __shared__ double red[R * R][256];   // cross-wave reduction of the block's output tiles (18 KB)
  const int nbj = (tiles_j + R - 1) / R;
  const int bi = blockIdx.x / nbj, bj = blockIdx.x % nbj;
  if (gram_block_skipped(bi, bj, sym)) return;
  const int ti0 = bi * R, tj0 = bj * R;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int col = lane & 15, kk = lane >> 4;
  // the chunk's rows are dealt to the 4 waves in quarters (whole k-steps); their sums are combined in wave
  // order below, so the result does not depend on timing
  const int64_t c0 = (int64_t)blockIdx.y * rows_per_chunk;
  const int64_t quarter = ((rows_per_chunk / 4) + 3) / 4 * 4;
  const int64_t r0 = c0 + wave * quarter;
  int64_t cend = c0 + rows_per_chunk;
  if (cend > n) cend = n;
  int64_t r1 = wave == 3 ? cend : r0 + quarter;
  if (r1 > cend) r1 = cend;
  f64x4 acc[R][R];
#pragma unroll
  for (int x = 0; x < R; ++x)
#pragma unroll
    for (int y = 0; y < R; ++y) acc[x][y] = f64x4{0.0, 0.0, 0.0, 0.0};
  const double* ap = A + ti0 * 16 + col;
  const double* bp = B + tj0 * 16 + col;
  int ao[R], bo[R];   // tiles past the edge alias the first one (loaded, multiplied, never stored)
#pragma unroll
  for (int x = 0; x < R; ++x) {
    ao[x] = ti0 + x < tiles_i ? 16 * x : 0;
    bo[x] = tj0 + x < tiles_j ? 16 * x : 0;
  }
  constexpr int KU = 4;   // k-steps per operand batch (24 loads); two batches alternate: one in flight, one in the MFMAs
  auto load = [&](double (&av)[KU][R], double (&bv)[KU][R], int64_t r) {
#pragma unroll
    for (int u = 0; u < KU; ++u) {
      const int64_t rr = r + 4 * u + kk;
      const bool ok = rr < r1;                    // ragged tail / past the end: zeros
      const int64_t rc = ok ? rr : c0;            // clamped address, value masked below
#pragma unroll
      for (int x = 0; x < R; ++x) {
        const double va = ap[rc * lda + ao[x]], vb = bp[rc * ldb + bo[x]];
        av[u][x] = ok ? va : 0.0;
        bv[u][x] = ok ? vb : 0.0;
      }
    }
  };
  auto mma = [&](const double (&av)[KU][R], const double (&bv)[KU][R]) {
#pragma unroll
    for (int u = 0; u < KU; ++u)
#pragma unroll
      for (int x = 0; x < R; ++x)
#pragma unroll
        for (int y = 0; y < R; ++y)
          acc[x][y] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u][x], bv[u][y], acc[x][y], 0, 0, 0);
  };
  if (r0 < r1) {
    double a0[KU][R], b0[KU][R], a1[KU][R], b1[KU][R];
    load(a0, b0, r0);
    for (int64_t r = r0; r < r1; r += 8 * KU) {
      load(a1, b1, r + 4 * KU);
      mma(a0, b0);
      load(a0, b0, r + 8 * KU);
      if (r + 4 * KU < r1) mma(a1, b1);
    }
  }
  // ordered cross-wave sum: wave 0 stores, waves 1..3 add in turn
#pragma unroll 1
  for (int w = 0; w < 4; ++w) {
    if (wave == w) {
#pragma unroll
      for (int x = 0; x < R; ++x)
#pragma unroll
        for (int y = 0; y < R; ++y)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            double* p = &red[x * R + y][(kk + 4 * g) * 16 + col];
            *p = w == 0 ? acc[x][y][g] : *p + acc[x][y][g];
          }
    }
    __syncthreads();
  }
  const int ntiles = tiles_i * tiles_j;
  for (int q = 0; q < R * R; ++q) {
    const int x = q / R, y = q % R;
    if (ti0 + x >= tiles_i || tj0 + y >= tiles_j) continue;
    const int tile = (ti0 + x) * tiles_j + (tj0 + y);
    partials[((size_t)blockIdx.y * ntiles + tile) * 256 + threadIdx.x] = red[q][threadIdx.x];
  }
}

// LDS-panel form of the Gram kernel (the fast path).  Measured on the way here: f64 MFMA does not overlap with
// the wave's own VALU work (a variant that built the feature panels on chip, 2 v_mul_f64 per operand, ran the
// MFMAs at exactly MFMA time + VALU time), and the register-operand kernel above stalls on its 24 global loads
// per batch.  So the operands take the one route that costs no VALU and no VGPRs: global_load_lds_dwordx4
// (scalar row base + one constant per-lane offset) straight into two 8-row x 96-column panels (A and B) per
// batch, ring-buffered, and conflict-free ds_read_b64 from there.  A workgroup (4 waves = 2 x 2 blocks of 3 x 3
// MFMA tiles) owns a 96 x 96 output block over a chunk of rows; blocks below the diagonal of the symmetric part
// are not launched.
//
// r03 (rocprofv3 PMC on the r02 kernel: matrix pipe busy 58 %, and 9 launched 96 x 96 blocks for 207 useful tiles
// of 324 at 288 features + 32 targets):
//  * a last column strip of <= 32 columns (the 32 targets of the notebook) is no longer a ragged 96-wide block
//    with two thirds of its MFMAs on padding: it is a NARROW block, 96 x 32, 3 x 1 tiles per wave, over chunks
//    5/2 as long (a third of the MFMAs per row, but the same DMA / barrier / address work: measured 0.72 us against
//    1.80 us per batch), so that every workgroup takes the same time (9 -> 7.2 block equivalents at T = 32);
//  * the DMA address is a scalar (global_load_lds with an SGPR base), one exec region covers a wave's four DMAs,
//    and only batches at the ragged end of the last chunk take the row-checked path;
//  * ONE workgroup of kGramGroups x 4 waves per CU instead of two of 4 waves: the groups take the batches of the
//    chunk in turn (group g: batches g, g + G, ...), each with its own panel ring, and their accumulators are added
//    in group order through LDS at the end.  Three waves per SIMD instead of two, in lockstep - with independent
//    workgroups (3 x 4 waves per CU, also measured) the oldest workgroup of a CU wins the matrix pipe, finishes at
//    86 us and leaves the youngest alone until 139 us - and a third of the partial sums: every workgroup writes
//    72 KB and the reduction reads them again (55 MB each way with 768 workgroups, 18 MB with 256);
//  * software pipeline over the barrier: the operands of batch b + 1 are read from LDS while the MFMAs of batch b
//    run, and after the barrier every wave first issues MFMAs and only then its ~60 scalar / DMA / LDS
//    instructions (4 240 -> 3 750 shader cycles per batch of 3 x 18 MFMAs = 3 456).
//  * the diagonal blocks of the symmetric part are a third kind (gram_diag_block below): upper tiles only.
//  What is left: the shader clock runs at 2.10 GHz under this kernel (s_memtime against s_memrealtime), not 2.4.
#ifndef HSR_GRAM_BUFS
#define HSR_GRAM_BUFS 4
#endif
#ifndef HSR_GRAM_GROUPS
#define HSR_GRAM_GROUPS 3
#endif
#ifndef HSR_GRAM_WGS
#define HSR_GRAM_WGS 1
#endif
constexpr int kGpCols = 96;          // panel width = 6 MFMA tiles
constexpr int kGpNarrow = 32;        // widest last strip that becomes a narrow block
constexpr int kGpRows = 8;           // rows per batch = 2 k-steps
constexpr int kGpBufs = HSR_GRAM_BUFS;   // panel ring: batch b lives in slot b % kGpBufs, the DMA runs kGpBufs - 1 batches ahead
constexpr int kGpAhead = kGpBufs - 1;
constexpr int kGpStride = 208;       // doubles per LDS row = [A 96 | B 96 | 16 spare]: 1664 B = 128 B mod 256 B -> kk rows 0/1 and 2/3 on disjoint banks
constexpr int kGpDmaPerWave = 2 * kGpRows / 4;   // panel rows each wave moves per batch (waves 0, 1: A; waves 2, 3: B)
constexpr int kGramGroups = HSR_GRAM_GROUPS;     // 4-wave groups per workgroup
constexpr int kGramWgs = HSR_GRAM_WGS;           // workgroups per CU
constexpr int kGramSlots = 256 * kGramWgs;       // resident workgroups of this kernel on the chip
constexpr int kGramThreads = 256 * kGramGroups;
constexpr int kGpRingDoubles = kGpBufs * kGpRows * kGpStride;       // one group's panel ring (4 slots: 53 248 B; three groups: 159 744 B)
constexpr int kGpDumpDoubles = 4 * 9 * 4 * 64;                      // one group's accumulators (72 KB)
constexpr int kGramLdsDoubles = (kGramGroups == 1 || kGramGroups * kGpRingDoubles > kGpDumpDoubles)
                                    ? kGramGroups * kGpRingDoubles : kGpDumpDoubles;
constexpr int kGpMaxBlocks = 64;

struct GramCore {                     // scalars only: handed to the block routine by value (a struct with a dynamically
  const double* A;                    // indexed array would be copied to scratch, and everything read from it would count
  const double* B;                    // as divergent)
  int64_t lda, ldb, n;
  int64_t rows_wide, rows_narrow;    // rows per chunk of a 96-wide / a narrow block
  int32_t na, nb, tiles_i, tiles_j;
  int32_t nwide, nnarrow;            // launched blocks of either kind (wide ones first in `blocks`, then diagonal, then narrow)
  int32_t chunks_wide, chunks_narrow;
  int32_t ndiag, chunks_diag;        // diagonal blocks of the symmetric part (upper tiles only)
  int64_t rows_diag;
  int32_t narrow_col, narrow_width;  // first column and width of the narrow strip of B
  int32_t total, per_xcd;            // workgroups with work; ceil(total / 8)
  double* partials;                  // [chunk][tile][256]
#ifdef HSR_GRAM_STAMPS
  unsigned long long* stamps;        // [workgroup][8] s_memrealtime at the phase boundaries (diagnostic builds only)
#endif
};
#ifdef HSR_GRAM_STAMPS
__device__ __forceinline__ unsigned long long gram_realtime() {   // 100 MHz
  unsigned long long t;
  asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  return t;
}
static unsigned long long* g_gram_stamps = nullptr;
extern "C" void hsr_dbg_gram_stamps(unsigned long long* dev) { g_gram_stamps = dev; }
#define GRAM_STAMP(i)                                                                                   \
  do {                                                                                                  \
    if (a.stamps && threadIdx.x == 0) a.stamps[(size_t)blockIdx.x * 8 + (i)] = gram_realtime();           \
  } while (0)
#else
#define GRAM_STAMP(i) \
  do {                \
  } while (0)
#endif
struct GramLdsArgs {
  GramCore c;
  uint8_t blocks[kGpMaxBlocks][2];   // (bi, bj) in 96-column units; bj is ignored for narrow blocks
};

__device__ __forceinline__ void glds16_s(uint32_t voff, const void* sbase, uint32_t lds_base) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(lds_base)
               : "memory");
}

__device__ __forceinline__ const double* uniform_ptr(const double* p) {   // a wave-uniform pointer, provably in SGPRs
  const uint64_t v = (uint64_t)(uintptr_t)p;
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
  return (const double*)(uintptr_t)(((uint64_t)hi << 32) | lo);
}

template <int RY>
__device__ __forceinline__ void gram_block(const GramCore a, double* pan_base, int acol0, int bcol0, int bw,
                                           int64_t c0, int64_t cend, int chunk) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane((tid >> 6) & 3);        // wave inside its group
  const int grp = __builtin_amdgcn_readfirstlane(tid >> 8);
  constexpr int G = kGramGroups;
  const int col = lane & 15, kk = lane >> 4;
  const int wx = wave >> 1, wy = wave & 1;
  // the chunk's batches are dealt to the groups in turn; all groups run the same number of local batches (a batch
  // past the end of the chunk is zero-filled), so that every wave meets the same barriers
  const int nbatch_all = (int)((cend - c0 + kGpRows - 1) / kGpRows);
  const int nbatch = (nbatch_all + G - 1) / G;              // local batches per group
  const int nfull = (int)((cend - c0) / kGpRows) / G;       // local batches whose rows all exist in every group
  pan_base += grp * kGpRingDoubles;                         // this group's ring
  typedef double Slot[kGpRows][kGpStride];
  Slot* pan = reinterpret_cast<Slot*>(pan_base);            // [slot][row][A cols | B cols]
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)(pan_base);

  // 16 panel rows per batch (8 of A, 8 of B), 4 per wave; a row is up to 96 doubles = 48 lanes x 16 bytes.  Lanes whose
  // columns lie past the block or the matrix stay off: their LDS words only feed tiles that are never stored.
  const int p = wave >> 1;                 // which panel this wave fills
  const int pr0 = (wave & 1) * kGpDmaPerWave;
  const int mcol0 = p ? bcol0 : acol0;
  const bool on = p ? (2 * lane < bw && mcol0 + 2 * lane < a.nb) : (lane < 48 && mcol0 + 2 * lane < a.na);
  const int64_t ld = p ? a.ldb : a.lda;
  const double* mbase = uniform_ptr((p ? a.B : a.A) + mcol0);
  const uint32_t voff = (uint32_t)lane * 16u;
  auto issue_full = [&](int b) {           // every row of batch b exists
    const int slot = b % kGpBufs;
    const double* src = mbase + (c0 + (int64_t)(b * G + grp) * kGpRows + pr0) * ld;
    const uint32_t dst = lds0 + (uint32_t)(((slot * kGpRows + pr0) * kGpStride + p * kGpCols) * 8);
    if (on) {
#pragma unroll
      for (int i = 0; i < kGpDmaPerWave; ++i) glds16_s(voff, src + i * ld, dst + (uint32_t)(i * kGpStride * 8));
    }
  };
  auto issue_any = [&](int b) {            // rows past the chunk are zero-filled by hand
    const int slot = b % kGpBufs;
#pragma unroll
    for (int i = 0; i < kGpDmaPerWave; ++i) {
      const int64_t row = c0 + (int64_t)(b * G + grp) * kGpRows + pr0 + i;
      const uint32_t dst = lds0 + (uint32_t)(((slot * kGpRows + pr0 + i) * kGpStride + p * kGpCols) * 8);
      if (row < cend) {
        if (on) glds16_s(voff, mbase + row * ld, dst);
      } else if (lane < 48) {
        pan[slot][pr0 + i][p * kGpCols + 2 * lane] = 0.0;
        pan[slot][pr0 + i][p * kGpCols + 2 * lane + 1] = 0.0;
      }
    }
  };

  constexpr int R = 3, KU = kGpRows / 4;
  f64x4 acc[R][RY];
#pragma unroll
  for (int x = 0; x < R; ++x)
#pragma unroll
    for (int y = 0; y < RY; ++y) acc[x][y] = f64x4{0.0, 0.0, 0.0, 0.0};
  const int asub = wx * 48;
  const int bsub = kGpCols + (RY == 3 ? wy * 48 : wy * 16);     // this wave's first B column inside the row
  struct Ops {
    double a[KU][R], b[KU][RY];
  };
  auto fetch = [&](Ops& o, int b) {        // the operands of local batch b: LDS -> registers
    const double (*ps)[kGpStride] = pan[b % kGpBufs];
#pragma unroll
    for (int u = 0; u < KU; ++u) {
#pragma unroll
      for (int x = 0; x < R; ++x) o.a[u][x] = ps[4 * u + kk][asub + 16 * x + col];
#pragma unroll
      for (int y = 0; y < RY; ++y) o.b[u][y] = ps[4 * u + kk][bsub + 16 * y + col];
    }
  };
  auto mma_row = [&](const Ops& o, int u, int x) {
#pragma unroll
    for (int y = 0; y < RY; ++y)
      acc[x][y] = __builtin_amdgcn_mfma_f64_16x16x4f64(o.a[u][x], o.b[u][y], acc[x][y], 0, 0, 0);
  };
  // One local batch.  On entry the panels of batches <= b + 1 have landed (for every wave of the workgroup) and `cur`
  // holds the operands of batch b.  The DMA of batch b + 3 goes to the slot of batch b - 1, whose LDS reads were issued
  // during step b - 2 and had completed before the barrier that ended it.  The operands of batch b + 1 are read while
  // the MFMAs of batch b run, so that no wave starts a batch by waiting for LDS behind the barrier; the step ends when
  // the panel of batch b + 2 has landed (the 4 DMAs of batch b + 3 may stay in flight).
  // The three waves of a SIMD leave the barrier together: each first feeds the matrix pipe (RY MFMAs whose operands are
  // in registers) and only then runs its ~60 scalar / DMA / LDS instructions, in the shadow of those MFMAs - with the
  // address work first the pipe stood idle for ~800 of the 4 240 cycles of a batch.
  auto step = [&](Ops& cur, Ops& nxt, int b) {
    const bool steady = b + kGpAhead < nfull;
    mma_row(cur, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    if (steady) issue_full(b + kGpAhead);
    else if (b + kGpAhead < nbatch) issue_any(b + kGpAhead);
    if (b + 1 < nbatch) fetch(nxt, b + 1);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < KU; ++u)
#pragma unroll
      for (int x = 0; x < R; ++x)
        if (u || x) mma_row(cur, u, x);
    if (steady) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kGpDmaPerWave) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  };

  GRAM_STAMP(0);
  for (int b = 0; b < kGpAhead && b < nbatch; ++b) issue_any(b);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  GRAM_STAMP(1);
#ifdef HSR_GRAM_STAMPS
  if (a.stamps && threadIdx.x == 0) a.stamps[(size_t)blockIdx.x * 8 + 6] = __builtin_readcyclecounter();   // s_memtime: shader clock
#endif
  {
    Ops o0, o1;
    fetch(o0, 0);
#pragma unroll 1
    for (int b = 0; b < nbatch; b += 2) {
      step(o0, o1, b);
      if (b + 1 < nbatch) step(o1, o0, b + 1);
    }
  }

#ifdef HSR_GRAM_STAMPS
  if (a.stamps && threadIdx.x == 0) a.stamps[(size_t)blockIdx.x * 8 + 7] = __builtin_readcyclecounter();
#endif
  GRAM_STAMP(2);
  // the groups' sums, added in group order through LDS (the rings are free: the loop ended in a barrier)
  if (G > 1) {
    double* dump = pan_base - grp * kGpRingDoubles + (wave * 9 * 4) * 64 + lane;
#pragma unroll 1
    for (int g = 1; g < G; ++g) {
      if (grp == g) {
#pragma unroll
        for (int x = 0; x < R; ++x)
#pragma unroll
          for (int y = 0; y < RY; ++y)
#pragma unroll
            for (int r = 0; r < 4; ++r) dump[((x * RY + y) * 4 + r) * 64] = acc[x][y][r];
      }
      __syncthreads();
      if (grp == 0) {
#pragma unroll
        for (int x = 0; x < R; ++x)
#pragma unroll
          for (int y = 0; y < RY; ++y)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[x][y][r] += dump[((x * RY + y) * 4 + r) * 64];
      }
      __syncthreads();
    }
    if (grp != 0) return;
  }
  GRAM_STAMP(3);

  const int ntiles = a.tiles_i * a.tiles_j;
#pragma unroll
  for (int x = 0; x < R; ++x) {
#pragma unroll
    for (int y = 0; y < RY; ++y) {
      const int ti = acol0 / 16 + wx * 3 + x, tjl = RY == 3 ? wy * 3 + y : wy;
      const int tj = bcol0 / 16 + tjl;
      if (ti >= a.tiles_i || tj >= a.tiles_j || 16 * tjl >= bw) continue;
      double* out = a.partials + ((size_t)chunk * ntiles + (size_t)ti * a.tiles_j + tj) * 256;
#pragma unroll
      for (int g = 0; g < 4; ++g) out[(kk + 4 * g) * 16 + col] = acc[x][y][g];
    }
  }
#ifdef HSR_GRAM_STAMPS
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  GRAM_STAMP(4);
  if (a.stamps && threadIdx.x == 0) a.stamps[(size_t)blockIdx.x * 8 + 5] = (unsigned long long)(RY * 1000000 + nbatch);
#endif
}

// Diagonal 96 x 96 block of the symmetric part (A's panel against itself): only its 21 upper tiles are needed - the reduction
// mirrors every tile below the diagonal - so the four waves of a group do not take a 3 x 3 quadrant each (36 tiles, one
// quadrant wasted, two half wasted) but a share of the upper triangle:
//     role 0: (0,0) (0,1) (0,2) (0,3) (0,4) (0,5)        role 1: (1,1) (1,2) (1,3) (1,4) (1,5)
//     role 2: (2,2) (2,3) (2,4) (2,5) (5,5)              role 3: (3,3) (3,4) (3,5) (4,4) (4,5)
// (at most two distinct A operands and six B operands per k-step), role = (wave + group) mod 4 so that every SIMD carries
// 15-16 MFMAs per k-step from its three waves instead of 27.  One panel per batch (2 rows per wave), both operands read
// from it.  Everything else - rings, split of the batches over the groups, software pipeline, combine - as gram_block.
constexpr int kGdTiles = 6;
struct GramDiagRole { int8_t n, arow[2], sel[kGdTiles], tj[kGdTiles]; };
constexpr GramDiagRole kGdRoles[4] = {{6, {0, 0}, {0, 0, 0, 0, 0, 0}, {0, 1, 2, 3, 4, 5}},
                                      {5, {1, 1}, {0, 0, 0, 0, 0, 0}, {1, 2, 3, 4, 5, 5}},
                                      {5, {2, 5}, {0, 0, 0, 0, 1, 1}, {2, 3, 4, 5, 5, 5}},
                                      {5, {3, 4}, {0, 0, 0, 1, 1, 1}, {3, 4, 5, 4, 5, 5}}};
constexpr int kGdDmaPerWave = kGpRows / 4;       // 8 panel rows per batch over 4 waves

template <int ROLE>
__device__ __forceinline__ void gram_diag_run(const GramCore a, double* pan_base, int acol0, int64_t c0, int64_t cend, int chunk,
                                              int wave, int grp) {
  constexpr GramDiagRole role = kGdRoles[ROLE];
  const int lane = threadIdx.x & 63;
  constexpr int G = kGramGroups;
  const int col = lane & 15, kk = lane >> 4;
  const int nbatch_all = (int)((cend - c0 + kGpRows - 1) / kGpRows);
  const int nbatch = (nbatch_all + G - 1) / G;
  const int nfull = (int)((cend - c0) / kGpRows) / G;
  double* ring = pan_base + grp * kGpRingDoubles;
  typedef double Slot[kGpRows][kGpStride];
  Slot* pan = reinterpret_cast<Slot*>(ring);
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)(ring);
  const int pr0 = wave * kGdDmaPerWave;
  const bool on = lane < 48 && acol0 + 2 * lane < a.nb;    // (A and B are the same matrix here: the columns of a ragged last block
  const double* mbase = uniform_ptr(a.A + acol0);          //  that lie past na are B's, and tiles in them are stored, too)
  const uint32_t voff = (uint32_t)lane * 16u;
  auto issue_full = [&](int b) {
    const int slot = b % kGpBufs;
    const double* src = mbase + (c0 + (int64_t)(b * G + grp) * kGpRows + pr0) * a.lda;
    const uint32_t dst = lds0 + (uint32_t)((slot * kGpRows + pr0) * kGpStride * 8);
    if (on) {
#pragma unroll
      for (int i = 0; i < kGdDmaPerWave; ++i) glds16_s(voff, src + i * a.lda, dst + (uint32_t)(i * kGpStride * 8));
    }
  };
  auto issue_any = [&](int b) {
    const int slot = b % kGpBufs;
#pragma unroll
    for (int i = 0; i < kGdDmaPerWave; ++i) {
      const int64_t row = c0 + (int64_t)(b * G + grp) * kGpRows + pr0 + i;
      const uint32_t dst = lds0 + (uint32_t)((slot * kGpRows + pr0 + i) * kGpStride * 8);
      if (row < cend) {
        if (on) glds16_s(voff, mbase + row * a.lda, dst);
      } else if (lane < 48) {
        pan[slot][pr0 + i][2 * lane] = 0.0;
        pan[slot][pr0 + i][2 * lane + 1] = 0.0;
      }
    }
  };
  constexpr int KU = kGpRows / 4;
  f64x4 acc[kGdTiles];
#pragma unroll
  for (int k = 0; k < kGdTiles; ++k) acc[k] = f64x4{0.0, 0.0, 0.0, 0.0};
  struct Ops {
    double a[KU][2], b[KU][kGdTiles];
  };
  auto fetch = [&](Ops& o, int b) {
    const double (*ps)[kGpStride] = pan[b % kGpBufs];
#pragma unroll
    for (int u = 0; u < KU; ++u) {
      o.a[u][0] = ps[4 * u + kk][16 * role.arow[0] + col];
      if (role.arow[1] != role.arow[0]) o.a[u][1] = ps[4 * u + kk][16 * role.arow[1] + col];
#pragma unroll
      for (int k = 0; k < role.n; ++k)
        if (k == 0 || role.tj[k] != role.tj[k - 1]) o.b[u][k] = ps[4 * u + kk][16 * role.tj[k] + col];
    }
  };
  auto mma_one = [&](const Ops& o, int u, int k) {
    // (a B operand shared by two consecutive tiles of the table - (2,5) (5,5) - was fetched once, for the first of them)
    const int kb = (k > 0 && role.tj[k] == role.tj[k - 1]) ? k - 1 : k;
    acc[k] = __builtin_amdgcn_mfma_f64_16x16x4f64(o.a[u][role.sel[k]], o.b[u][kb], acc[k], 0, 0, 0);
  };
  auto step = [&](Ops& cur, Ops& nxt, int b) {
    const bool steady = b + kGpAhead < nfull;
    mma_one(cur, 0, 0);
    mma_one(cur, 0, 1);
    __builtin_amdgcn_sched_barrier(0);
    if (steady) issue_full(b + kGpAhead);
    else if (b + kGpAhead < nbatch) issue_any(b + kGpAhead);
    if (b + 1 < nbatch) fetch(nxt, b + 1);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < KU; ++u)
#pragma unroll
      for (int k = 0; k < role.n; ++k)
        if (u || k > 1) mma_one(cur, u, k);
    if (steady) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kGdDmaPerWave) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  };
  GRAM_STAMP(0);
  for (int b = 0; b < kGpAhead && b < nbatch; ++b) issue_any(b);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  GRAM_STAMP(1);
#ifdef HSR_GRAM_STAMPS
  if (a.stamps && threadIdx.x == 0) a.stamps[(size_t)blockIdx.x * 8 + 6] = __builtin_readcyclecounter();
#endif
  {
    Ops o0, o1;
    fetch(o0, 0);
#pragma unroll 1
    for (int b = 0; b < nbatch; b += 2) {
      step(o0, o1, b);
      if (b + 1 < nbatch) step(o1, o0, b + 1);
    }
  }
#ifdef HSR_GRAM_STAMPS
  if (a.stamps && threadIdx.x == 0) a.stamps[(size_t)blockIdx.x * 8 + 7] = __builtin_readcyclecounter();
#endif
  GRAM_STAMP(2);
  // the groups' sums, added in group order through LDS; a role sits in a different wave in every group, so the dump is
  // indexed by role
  if (G > 1) {
    double* dump = pan_base + (ROLE * kGdTiles * 4) * 64 + lane;
#pragma unroll 1
    for (int g = 1; g < G; ++g) {
      if (grp == g) {
#pragma unroll
        for (int k = 0; k < role.n; ++k)
#pragma unroll
          for (int r = 0; r < 4; ++r) dump[(k * 4 + r) * 64] = acc[k][r];
      }
      __syncthreads();
      if (grp == 0) {
#pragma unroll
        for (int k = 0; k < role.n; ++k)
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[k][r] += dump[(k * 4 + r) * 64];
      }
      __syncthreads();
    }
    if (grp != 0) return;
  }
  GRAM_STAMP(3);
  const int ntiles = a.tiles_i * a.tiles_j;
#pragma unroll
  for (int k = 0; k < role.n; ++k) {
    const int ti = acol0 / 16 + role.arow[role.sel[k]], tj = acol0 / 16 + role.tj[k];
    if (ti >= a.tiles_i || tj >= a.tiles_j) continue;
    double* out = a.partials + ((size_t)chunk * ntiles + (size_t)ti * a.tiles_j + tj) * 256;
#pragma unroll
    for (int g = 0; g < 4; ++g) out[(kk + 4 * g) * 16 + col] = acc[k][g];
  }
#ifdef HSR_GRAM_STAMPS
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  GRAM_STAMP(4);
  if (a.stamps && threadIdx.x == 0) a.stamps[(size_t)blockIdx.x * 8 + 5] = (unsigned long long)(2 * 1000000 + nbatch);   // kind 2 = diagonal
#endif
}

__device__ __forceinline__ void gram_diag_block(const GramCore a, double* pan_base, int acol0, int64_t c0, int64_t cend, int chunk) {
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane((tid >> 6) & 3);
  const int grp = __builtin_amdgcn_readfirstlane(tid >> 8);
  switch ((wave + grp) & 3) {      // wave-uniform; every role runs the same sequence of barriers
    case 0: gram_diag_run<0>(a, pan_base, acol0, c0, cend, chunk, wave, grp); break;
    case 1: gram_diag_run<1>(a, pan_base, acol0, c0, cend, chunk, wave, grp); break;
    case 2: gram_diag_run<2>(a, pan_base, acol0, c0, cend, chunk, wave, grp); break;
    default: gram_diag_run<3>(a, pan_base, acol0, c0, cend, chunk, wave, grp); break;
  }
}

// Workgroup id -> (block, chunk): consecutive ids are the blocks of one chunk of rows, and the eight XCDs take
// contiguous runs of ids (hardware deals workgroup w to XCD w % 8), so that the workgroups that read the same rows
// of A sit behind the same L2.
__global__ __launch_bounds__(kGramThreads, HSR_GRAM_WGS) void gram_f64_lds_kernel(const GramLdsArgs args) {
  extern __shared__ __attribute__((aligned(16))) double pan[];   // kGramLdsDoubles
  const GramCore a = args.c;
  const int id = (int)(blockIdx.x % 8) * a.per_xcd + (int)(blockIdx.x / 8);
  if (id >= a.total) return;
  const int wide_ids = a.nwide * a.chunks_wide;
  if (id < wide_ids) {
    const int k = id % a.nwide, chunk = id / a.nwide;
    const int64_t c0 = (int64_t)chunk * a.rows_wide;
    int64_t cend = c0 + a.rows_wide;
    if (cend > a.n) cend = a.n;
    gram_block<3>(a, pan, args.blocks[k][0] * kGpCols, args.blocks[k][1] * kGpCols, kGpCols, c0, cend, chunk);
  } else if (id < wide_ids + a.ndiag * a.chunks_diag) {
    const int k = a.nwide + (id - wide_ids) % a.ndiag, chunk = (id - wide_ids) / a.ndiag;
    const int64_t c0 = (int64_t)chunk * a.rows_diag;
    int64_t cend = c0 + a.rows_diag;
    if (cend > a.n) cend = a.n;
    gram_diag_block(a, pan, args.blocks[k][0] * kGpCols, c0, cend, chunk);
  } else {
    const int nid = id - wide_ids - a.ndiag * a.chunks_diag;
    const int k = a.nwide + a.ndiag + nid % a.nnarrow, chunk = nid / a.nnarrow;
    const int64_t c0 = (int64_t)chunk * a.rows_narrow;
    int64_t cend = c0 + a.rows_narrow;
    if (cend > a.n) cend = a.n;
    gram_block<1>(a, pan, args.blocks[k][0] * kGpCols, a.narrow_col, a.narrow_width, c0, cend, chunk);
  }
}

// chunks summed in index order -> C[(ti*16 + r) * ldc + tj*16 + c]; with `sym` every tile below the diagonal is the transpose
// of its mirror tile (whole skipped blocks, and the lower tiles of diagonal blocks).  Tiles from column tile `narrow_tj` on
// have `chunks_narrow` chunks, tiles of diagonal blocks (edge `blk` tiles) `chunks_diag` when that is > 0.
__global__ __launch_bounds__(256) void gram_reduce_kernel(const double* __restrict__ partials, int ntiles, int chunks,
                                                          int tiles_j, int sym, int blk, int narrow_tj,
                                                          int chunks_narrow, int chunks_diag, double* __restrict__ C, int64_t ldc) {
  const int tile = blockIdx.x, e = threadIdx.x;
  const int ti = tile / tiles_j, tj = tile % tiles_j;
  const bool mirror = sym && tj < ti;
  const int si = mirror ? tj : ti, sj = mirror ? ti : tj;           // the tile that was computed
  const int src_tile = si * tiles_j + sj;
  const int src_e = mirror ? (e & 15) * 16 + (e >> 4) : e;
  const int nc = sj >= narrow_tj ? chunks_narrow : (sym && chunks_diag > 0 && si / blk == sj / blk) ? chunks_diag : chunks;
  const double* p = partials + (size_t)src_tile * 256 + src_e;
  const size_t step = (size_t)ntiles * 256;
  double s = 0.0;
  int c = 0;
  for (; c + 8 <= nc; c += 8) {          // eight loads in flight, the sum stays in chunk order
    double v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = p[(size_t)(c + i) * step];
#pragma unroll
    for (int i = 0; i < 8; ++i) s += v[i];
  }
  for (; c < nc; ++c) s += p[(size_t)c * step];
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
  int32_t nan_bad;             // 1: a pixel with a non-finite (or nodata) input comes out NaN in every target
  int32_t use_nodata;
  float nodata;
  float* out;                  // (T, out_stride)
  int64_t out_stride;
};

// predict_cube_logit's rule for unusable pixels (Spectral_matching.ipynb raw lines 197-203): any input non-finite, or
// close to the nodata value in torch.isclose's sense (|x - nd| <= 1e-8 + 1e-5 |nd|, equal infinities close, NaN never).
// Epilogue of the predict103 kernels.  r03 (rocprofv3 PMC: matrix pipe busy 57 % of the SIMD cycles, 88 k wave-cycles per
// tile of which 55 k are MFMA): the first version loaded the bias of each of a lane's 16 x TT targets inside the tile loop,
// every load in its own exec-masked block behind `trg < T` with its own s_waitcnt - ~0.5 k cycles of exposed latency each,
// 24 k per tile - and evaluated 1 / (1 + e) as an IEEE division (10 instructions).  The targets of a lane do not depend on
// the tile, so the bias values are loaded ONCE before the tile loop and the accumulators start from them; the sigmoid uses
// v_rcp_f32 (1 ulp; the bar is 1e-4 in reflectance); only the store stays behind `trg < T`.
__device__ __forceinline__ float predict_activation(float v, int act) {
  if (act) {
    v = v < -50.0f ? -50.0f : (v > 50.0f ? 50.0f : v);   // NaN falls through, like np.clip
    v = __builtin_amdgcn_rcpf(1.0f + __expf(-v));
  }
  return v;
}

__device__ __forceinline__ bool pred_bad_input(float x, int use_nodata, float nd) {
  const bool nonfinite = (__float_as_uint(x) & 0x7f800000u) == 0x7f800000u;
  const bool close = use_nodata && (x == nd || fabsf(x - nd) <= 1e-8f + 1e-5f * fabsf(nd));
  return nonfinite || close;
}

constexpr int kPredPix = 64;      // pixels per workgroup tile
constexpr int kPredThreads = 256;

template <int TT>   // number of 32-wide target tiles held by a wave (accumulators: TT * 16 VGPRs)
__global__ __launch_bounds__(kPredThreads) void predict_kernel(const PredArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int ldphi = a.kpad + 1;                      // odd row stride -> conflict-free column walks
  float* phi = reinterpret_cast<float*>(smem);       // [64][ldphi]
  float* zt = phi + kPredPix * ldphi;                // [64][n_in + 1]
  uint32_t* badl = reinterpret_cast<uint32_t*>(zt + kPredPix * (a.n_in + 1));   // [64]
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
    if (t < kPredPix) {
      uint32_t bad = 0u;
      if (a.nan_bad && p0 + t < a.npix)
        for (int c = 0; c < a.n_in; ++c) bad |= pred_bad_input(a.x[(p0 + t) * a.x_ps + c * a.x_cs], a.use_nodata, a.nodata) ? 1u : 0u;
      badl[t] = bad;
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
            float v = predict_activation(acc[q][r] + a.bias[trg], a.act);
            if (badl[ph + j]) v = __uint_as_float(0x7fc00000u);
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


// (r04, measured and dropped: two accumulator tiles per wave for T <= 32 - even / odd steps, added at the end - so that a wave always
// has an independent MFMA to issue: 0.2039 vs 0.2054 ms per Mpixel.  The chain's latency is not what idles the matrix pipe.)
// the 10 inputs of one pixel.  Pixel-major rows (x_cs == 1, the (N, 10) arrays of predict()) with an even pitch and an 8-byte
// aligned base are read as five float2 instead of ten scalars: a wave's 32 pixels then touch their 40-byte rows once per 8 bytes
__device__ __forceinline__ void pred_load10(const PredArgs& a, int64_t pc, float (&x)[10]) {
  if (a.x_cs == 1 && (a.x_ps & 1) == 0 && (((uintptr_t)a.x) & 7) == 0) {       // launch-uniform
    const float2* r = reinterpret_cast<const float2*>(a.x + pc * a.x_ps);
#pragma unroll
    for (int c = 0; c < 5; ++c) {
      const float2 v = r[c];
      x[2 * c] = v.x;
      x[2 * c + 1] = v.y;
    }
  } else {
#pragma unroll
    for (int c = 0; c < 10; ++c) x[c] = a.x[pc * a.x_ps + c * a.x_cs];
  }
}

// workgroups per CU: a wave of the T <= 32 kernel carries ONE accumulator chain (143 dependent MFMAs per tile) and a 16-value
// sigmoid epilogue; with four waves per SIMD instead of two the matrix pipe has something to run while a wave is in its
// epilogue or waits for its next operand (r03: 0.214 -> 0.199 ms per Mpixel, 111 VGPRs, W resident in 36.6 KB of LDS)
#ifndef HSR_PRED_OCC1
#define HSR_PRED_OCC1 4
#endif
#ifndef HSR_PRED_OCC2
#define HSR_PRED_OCC2 2
#endif
constexpr int pred103_occupancy(int tt) { return tt == 1 ? HSR_PRED_OCC1 : tt == 2 ? HSR_PRED_OCC2 : 2; }
template <int TT, bool WHOLE>
__global__ __launch_bounds__(256, pred103_occupancy(TT)) void predict103_kernel(const PredArgs a) {
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
  float bv[TT][16];                                 // bias of this lane's targets: tile-invariant
#pragma unroll
  for (int q = 0; q < TT; ++q)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int trg = q * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
      bv[q][r] = trg < a.T ? a.bias[trg] : 0.0f;
    }
  // the 10 inputs of the NEXT tile are loaded while this tile's MFMA chain runs (a fresh load at the top of every tile left
  // the matrix pipe idle for one HBM latency per tile: ~2 us of a 9-27 us tile)
  float xn[10];
  auto load_inputs = [&](int64_t tile_) {
    const int64_t p_ = tile_ * 128 + wave * 32 + j;
    const int64_t pc_ = p_ < a.npix ? p_ : a.npix - 1;
    pred_load10(a, pc_, xn);
  };
  if ((int64_t)blockIdx.x * 128 < a.npix) load_inputs(blockIdx.x);
  for (int64_t tile = blockIdx.x; tile * 128 < a.npix; tile += gridDim.x) {
    const int64_t p = tile * 128 + wave * 32 + j;
    float z[11];
    bool bad = false;
#pragma unroll
    for (int c = 0; c < 10; ++c) {
      const float xr = xn[c];
      bad = bad || pred_bad_input(xr, a.use_nodata, a.nodata);
      z[c] = (xr - a.mean[c]) * a.inv[c];
    }
    bad = bad && a.nan_bad != 0;
    z[10] = 1.0f;
    if ((tile + gridDim.x) * 128 < a.npix) load_inputs(tile + gridDim.x);
    f32x16 acc[TT];
#pragma unroll
    for (int q = 0; q < TT; ++q)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[q][r] = bv[q][r];
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
          float v = predict_activation(acc[q][r], a.act);
          if (bad) v = __uint_as_float(0x7fc00000u);
          if (trg < tmax) orow[(size_t)tu * ostride] = v;
        }
      }
    }
  }
}

typedef float f32x4 __attribute__((ext_vector_type(4)));
// ---- the K axis in ORBIT order (r04) ------------------------------------------------------------------------------------------
// An MFMA's K index is spread over lane groups (two for v_mfma_f32_32x32x2_f32, four for v_mfma_f32_16x16x4_f32), so the lane
// groups of one instruction need DIFFERENT monomials - compile-time index triples - of their pixel.  Rounds 1-3 formed both
// products in every lane and picked one (v_cndmask): ~5 VALU per step, and the measurements of r04 say the SIMD does not hide
// VALU work under its own MFMAs (T <= 32: 64 % matrix-pipe busy with 20 cycles of VALU per 64-cycle MFMA; a 16-target variant with
// 20 VALU per 64 cycles of MFMA ran SLOWER than the 32-target kernel).  The order of the K axis is free (a sum): the 286 index
// multisets {a <= b <= c} over the 11 symbols (10 inputs + the constant 1) are grouped into orbits of a symbol permutation sigma
// of order G = number of lane groups; step s takes one orbit, and lane group g evaluates the orbit's REPRESENTATIVE on its own copy
// of the inputs, permuted once per tile (Z_g[m] = z[sigma^g(m)]): Z[a] Z[b] Z[c] in group g is the monomial sigma^g(a, b, c).  One code
// path, ~1.3 v_mul per step, no pick.  Members an orbit repeats (short orbits) meet a zero row of W; the staging loop gathers W's
// rows in orbit order.   G = 2: sigma = (0 1)(2 3)(4 5)(6 7)(8 9): 146 steps for 143.   G = 4: sigma = (0 1 2 3)(4 5 6 7): 82 for 71.5.
template <int G>
constexpr int perm_sym(int m, int g) {
  if (G == 2) return (m < 10 && (g & 1)) ? (m ^ 1) : m;
  return m < 4 ? (m + g) & 3 : (m < 8 ? 4 + ((m - 4 + g) & 3) : m);
}
template <int G>
struct Orbits103 {
  uint8_t rep[160][3];     // representative triple of step s (symbols 0 .. 10; 10 = the constant 1)
  int16_t src[160][G];     // feature row of W that lane group g's member of the orbit multiplies, -1: none (a repeat, or 1 * 1 * 1)
  int steps;
};
template <int G>
constexpr Orbits103<G> make_orbits103() {
  Orbits103<G> o{};
  bool seen[11][11][11] = {};
  int n = 0;
  for (int a = 0; a <= 10; ++a)
    for (int b = a; b <= 10; ++b)
      for (int c = b; c <= 10; ++c) {
        if (seen[a][b][c]) continue;
        o.rep[n][0] = (uint8_t)a; o.rep[n][1] = (uint8_t)b; o.rep[n][2] = (uint8_t)c;
        for (int g = 0; g < G; ++g) {
          int t0 = perm_sym<G>(a, g), t1 = perm_sym<G>(b, g), t2 = perm_sym<G>(c, g);
          if (t0 > t1) { const int x = t0; t0 = t1; t1 = x; }
          if (t1 > t2) { const int x = t1; t1 = t2; t2 = x; }
          if (t0 > t1) { const int x = t0; t0 = t1; t1 = x; }
          int row = -1;
          if (!seen[t0][t1][t2]) {
            seen[t0][t1][t2] = true;
            for (int f = 0; f < 285; ++f)      // kTab103 lists a monomial's indices in ascending order, padded with 10
              if (kTab103.v[f][0] == t0 && kTab103.v[f][1] == t1 && kTab103.v[f][2] == t2) row = f;
          }
          o.src[n][g] = (int16_t)row;
        }
        ++n;
      }
  o.steps = n;
  return o;
}
constexpr Orbits103<4> kOrb103 = make_orbits103<4>();
constexpr Orbits103<2> kOrb2 = make_orbits103<2>();
static_assert(kOrb103.steps == 82, "orbits of (0 1 2 3)(4 5 6 7) on the 286 index multisets");
static_assert(kOrb2.steps == 146, "orbits of (0 1)(2 3)(4 5)(6 7)(8 9) on the 286 index multisets");
constexpr int kStepsOrb = kOrb103.steps, kStepsOrb2 = kOrb2.steps;
static int16_t* g_orb_rows_dev = nullptr;              // kOrb103.src then kOrb2.src on the device (uploaded by hsr_polyfeat_prepare, not on a launch path)

// the slice kernels' MFMA chain over the orbit-ordered K axis (G = 2): Z is the lane half's permuted copy of the inputs
template <int TT>
__device__ __forceinline__ void mfma_steps_orb2(const float (&Z)[11], int kh, const float* __restrict__ wl, int ldwl, int j, f32x16 (&acc)[TT]) {
  float wc[TT], wn[TT];
  const float* wr = wl + kh * ldwl + j;              // A[i = target][k] = staged row 2 s + kh
#pragma unroll
  for (int q = 0; q < TT; ++q) wc[q] = wr[q * 32];
#pragma unroll
  for (int s = 0; s < kStepsOrb2; ++s) {
    const float bv = Z[kOrb2.rep[s][0]] * Z[kOrb2.rep[s][1]] * Z[kOrb2.rep[s][2]];      // B[k = 2 s + kh][pixel j]
    if (s + 1 < kStepsOrb2) {
#pragma unroll
      for (int q = 0; q < TT; ++q) wn[q] = wr[(size_t)2 * (s + 1) * ldwl + q * 32];
    }
#pragma unroll
    for (int q = 0; q < TT; ++q) acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(wc[q], bv, acc[q], 0, 0, 0);
#pragma unroll
    for (int q = 0; q < TT; ++q) wc[q] = wn[q];
    __builtin_amdgcn_sched_barrier(0);
  }
}

// Many targets (T > 96, e.g. EMIT's 285 bands): the chunked path above re-stages the whole 329 KB of W through LDS for
// every 128-pixel tile, with nine barriers per tile, and sat at 39 % of the f32-MFMA peak against 50 % for T <= 32 where
// W is resident.  W does not fit one LDS, but a SLICE of 96 targets does (286 x 96 x 4 = 110 KB): blockIdx.y picks the
// slice, the slice is staged once per workgroup and stays for the whole launch, and the workgroup (12 waves = 3 per
// SIMD, each wave its own 32 pixels) walks the pixel tiles with no staging and no barrier in the loop.  The monomials
// of a pixel are recomputed once per slice (2-4 v_mul per MFMA step of 3 x 64 cycles: free) and its 10 inputs re-read
// (40 B per slice: nothing).  3 accumulator tiles per wave instead of 9.
// r03: (a) 8 -> 12 waves on the one 96-target slice a CU holds (3 per SIMD, 168 VGPRs, no spill; 16 waves = 128 VGPRs spill):
// 1.675 -> 1.603 ms at T = 285 on one box; (b) the kernel is a template on the slice width, and it also serves 33 <= T <= 96 with
// ONE slice: the chunked predict103_kernel<2 / 3, false> (nine barriers and a re-staged W per 128-pixel tile) took 0.83-0.87 ms
// per Mpixel for 65-96 targets, the 96-wide slice kernel 0.51-0.52 ms; (c) slices are as narrow as the target count allows
// (T = 97: two slices of 64 instead of two of 96).  64-target slices run 16 waves (4 per SIMD) on their 73 KB of W.
// Wave priorities of the slice and 16-target kernels (r04, after they paid in the uint16 K1): level of a wave during its MFMA chain /
// outside it (inputs, standardisation, activation, stores).  A/B on one box, ms per Mpixel at T = 16 / 32 / 96 / 285
// (profiles/r04_k4_ridge.md): none 0.129 / 0.203 / 0.528 / 1.523; REST 1 (shipped): 0.123 / 0.198 / 0.520 / 1.491; REST 3: the same;
// CHAIN 3 or 1: no change; STAGGER (the four waves of a SIMD at four different levels during the chain, so that they drift apart and
// one wave's epilogue meets another's chain - either guess of the wave -> SIMD mapping): T = 32 unchanged at 0.200-0.205.
#ifndef HSR_PRED_PRIO_CHAIN
#define HSR_PRED_PRIO_CHAIN 0
#endif
#ifndef HSR_PRED_PRIO_REST
#define HSR_PRED_PRIO_REST 1
#endif
#ifndef HSR_PRED_PRIO_STAGGER     // 1: chain level = (wave >> 2) & 3, 2: wave & 3 - the waves of a SIMD at DIFFERENT levels drift apart
#define HSR_PRED_PRIO_STAGGER 0
#endif
__device__ __forceinline__ void pred_chain_prio(int wave) {
  if (HSR_PRED_PRIO_STAGGER == 0) {
    if (HSR_PRED_PRIO_CHAIN != HSR_PRED_PRIO_REST) __builtin_amdgcn_s_setprio(HSR_PRED_PRIO_CHAIN);
    return;
  }
  const int k = __builtin_amdgcn_readfirstlane(HSR_PRED_PRIO_STAGGER == 1 ? (wave >> 2) & 3 : wave & 3);
  if (k == 3) __builtin_amdgcn_s_setprio(3);
  else if (k == 2) __builtin_amdgcn_s_setprio(2);
  else if (k == 1) __builtin_amdgcn_s_setprio(1);
  else __builtin_amdgcn_s_setprio(0);
}
__device__ __forceinline__ void pred_rest_prio() {
  if (HSR_PRED_PRIO_STAGGER != 0 || HSR_PRED_PRIO_CHAIN != HSR_PRED_PRIO_REST) __builtin_amdgcn_s_setprio(HSR_PRED_PRIO_REST);
}
#ifndef HSR_SLICE_WAVES
#define HSR_SLICE_WAVES 12
#endif
#ifndef HSR_SLICE2_WAVES
#define HSR_SLICE2_WAVES 16
#endif
#ifndef HSR_SLICE1_WAVES
#define HSR_SLICE1_WAVES 16
#endif
#ifndef HSR_SLICE1_WGS
#define HSR_SLICE1_WGS 1
#endif
constexpr int slice_waves(int tt) { return tt == 3 ? HSR_SLICE_WAVES : (tt == 1 ? HSR_SLICE1_WAVES : HSR_SLICE2_WAVES); }
constexpr int slice_wgs(int tt) { return tt == 1 ? HSR_SLICE1_WGS : 1; }       // workgroups per CU
template <int TT>
__global__ __launch_bounds__(64 * slice_waves(TT), (slice_waves(TT) * slice_wgs(TT) + 3) / 4) void predict103_slice_kernel(const PredArgs a, const int16_t* __restrict__ src_rows) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* wl = reinterpret_cast<float*>(smem);       // [2 * kStepsOrb2][Tp]: W's rows in orbit order (kOrb2)
  constexpr int Tp = TT * 32;
  constexpr int kSliceThreads = 64 * slice_waves(TT);      // waves per workgroup = waves per CU (one workgroup per CU)
  constexpr int kSlicePix = 32 * slice_waves(TT);          // pixels per tile: 32 per wave
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int j = lane & 31, kh = lane >> 5;
  const int t0 = blockIdx.y * Tp;                   // first target of this workgroup's slice
  for (int i = t; i < 2 * kStepsOrb2 * Tp; i += kSliceThreads) {
    const int r = i / Tp, c = i % Tp;
    const int f = src_rows[r];
    wl[i] = (f >= 0 && t0 + c < a.T) ? a.W[(size_t)f * a.ldw + t0 + c] : 0.0f;
  }
  __syncthreads();
  float bv[TT][16];                                    // bias of this lane's targets: tile-invariant
#pragma unroll
  for (int q = 0; q < TT; ++q)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int trg = t0 + q * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
      bv[q][r] = trg < a.T ? a.bias[trg] : 0.0f;
    }
  float xn[10];                                        // inputs of the next tile, loaded under this tile's MFMA chain
  auto load_inputs = [&](int64_t tile_) {
    const int64_t p_ = tile_ * kSlicePix + wave * 32 + j;
    const int64_t pc_ = p_ < a.npix ? p_ : a.npix - 1;
    pred_load10(a, pc_, xn);
  };
  if ((int64_t)blockIdx.x * kSlicePix < a.npix) load_inputs(blockIdx.x);
  if (HSR_PRED_PRIO_REST != 0) __builtin_amdgcn_s_setprio(HSR_PRED_PRIO_REST);
  for (int64_t tile = blockIdx.x; tile * kSlicePix < a.npix; tile += gridDim.x) {
    const int64_t p = tile * kSlicePix + wave * 32 + j;
    float z[11];
    bool bad = false;
#pragma unroll
    for (int c = 0; c < 10; ++c) {
      const float xr = xn[c];
      bad = bad || pred_bad_input(xr, a.use_nodata, a.nodata);
      z[c] = (xr - a.mean[c]) * a.inv[c];
    }
    bad = bad && a.nan_bad != 0;
    z[10] = 1.0f;
#pragma unroll
    for (int c = 0; c < 10; c += 2) {                   // the lane half's copy: kh = 1 swaps the inputs pairwise (sigma of kOrb2)
      const float lo = z[c], hi = z[c + 1];
      z[c] = kh ? hi : lo;
      z[c + 1] = kh ? lo : hi;
    }
    if ((tile + gridDim.x) * kSlicePix < a.npix) load_inputs(tile + gridDim.x);
    f32x16 acc[TT];
#pragma unroll
    for (int q = 0; q < TT; ++q)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[q][r] = bv[q][r];
    pred_chain_prio(wave);
    mfma_steps_orb2<TT>(z, kh, wl, Tp, j, acc);
    pred_rest_prio();
    if (p < a.npix) {
      int64_t ostride = a.out_stride;
      int tmax = a.T;
      asm volatile("" : "+s"(ostride), "+s"(tmax));        // keep the addressing inside the loop (see predict103_kernel)
      float* orow = a.out + (size_t)(t0 + 4 * kh) * ostride + p;
#pragma unroll
      for (int q = 0; q < TT; ++q) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int tu = q * 32 + (r & 3) + 8 * (r >> 2);
          const int trg = t0 + tu + 4 * kh;
          float v = predict_activation(acc[q][r], a.act);
          if (bad) v = __uint_as_float(0x7fc00000u);
          if (trg < tmax) orow[(size_t)tu * ostride] = v;
        }
      }
    }
  }
}

// Few targets (r04).  v_mfma_f32_32x32x2_f32 puts 32 targets on the M axis, so T = 16 cost what T = 32 costs (r03: 0.189 ms per
// Mpixel for both).  v_mfma_f32_16x16x4_f32 has the same rate (16 x 16 x 4 x 2 flop in 32 cycles) with 16 targets per tile; a wave
// holds one tile of 16 targets x two sets of 16 pixels.
//   A[i = target][k] = W[k][target]: lane (i = lane & 15, g = lane >> 4) reads row 4 s + g of the staged W (64 consecutive LDS words);
//   B[k][j = pixel]:  lane (j, g) must supply "monomial 4 s + g" of ITS pixel - four DIFFERENT compile-time index triples in the four
//     lane groups of one instruction.  Two versions that pick per lane were measured and dropped: a ?: chain over the four products
//     (hipcc wraps every pick in exec-masked branches: 0.173 ms) and bit-mask picks (20 VALU per step against 64 cycles of matrix
//     pipe, and the SIMD does not hide them: 0.201 ms).  What is built: the ORDER of the K axis is free (a sum), so the 286 index
//     multisets {a <= b <= c} over the 11 symbols (10 inputs + the constant) are grouped into orbits of the symbol permutation
//     sigma = (0 1 2 3)(4 5 6 7); step s takes one orbit: lane group g computes the orbit's representative on ITS OWN copy of the
//     inputs, permuted once per tile (Z_g[m] = z[sigma^g(m)]): Z[a] Z[b] Z[c] in group g IS the monomial sigma^g(a, b, c).  One code
//     path, two v_mul per step and pixel set, no pick.  82 orbits (66 of four, 6 of two, 10 fixed) instead of 286 / 4 = 71.5 steps:
//     the repeated members of short orbits meet a zero row of W.  The staging loop gathers W's rows in orbit order.
//   D[i][j]: lane (j, g) holds targets 4 g + r, r = 0 .. 3, of pixel j: 64-byte store segments per target.
// One 16-wave workgroup per CU, W staged once.  (Three such tiles for 33 <= T <= 48 were measured too: 82 x 6 MFMAs of 32 cycles are
// no better than the 64-target slice's 143 x 2 of 64; those T keep the slice kernel.)
#ifndef HSR_X16_WAVES
#define HSR_X16_WAVES 12
#endif
#ifndef HSR_X16_WGS
#define HSR_X16_WGS 2
#endif
constexpr int kX16Waves = HSR_X16_WAVES, kX16Wgs = HSR_X16_WGS;     // 2 x 12 waves per CU = 6 per SIMD at <= 80 VGPRs (r04: 16 x 1: 0.132 ms per Mpixel)
__global__ __launch_bounds__(64 * kX16Waves, kX16Wgs) void predict103_x16_kernel(const PredArgs a, const int16_t* __restrict__ src_rows) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* wl = reinterpret_cast<float*>(smem);          // [4 * kStepsOrb][16]
  constexpr int kPix = 32 * kX16Waves;                 // pixels per workgroup tile: 32 per wave
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int i = lane & 15, g = lane >> 4;
  for (int e = t; e < 4 * kStepsOrb * 16; e += 64 * kX16Waves) {
    const int r = e >> 4, c = e & 15;
    const int f = src_rows[r];
    wl[e] = (f >= 0 && c < a.T) ? a.W[(size_t)f * a.ldw + c] : 0.0f;
  }
  __syncthreads();
  float bias[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) bias[r] = 4 * g + r < a.T ? a.bias[4 * g + r] : 0.0f;
  float xn[2][10];
  auto load_inputs = [&](int64_t tile_) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int64_t p_ = tile_ * kPix + wave * 32 + h * 16 + i;
      pred_load10(a, p_ < a.npix ? p_ : a.npix - 1, xn[h]);
    }
  };
  if ((int64_t)blockIdx.x * kPix < a.npix) load_inputs(blockIdx.x);
  if (HSR_PRED_PRIO_REST != 0) __builtin_amdgcn_s_setprio(HSR_PRED_PRIO_REST);
  for (int64_t tile = blockIdx.x; tile * kPix < a.npix; tile += gridDim.x) {
    float Z[2][11];                                    // this lane group's permuted copy of the standardised inputs
    bool bad[2] = {false, false};
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      float z[8];
#pragma unroll
      for (int c = 0; c < 10; ++c) {
        const float xr = xn[h][c];
        bad[h] = bad[h] || pred_bad_input(xr, a.use_nodata, a.nodata);
        const float zc = (xr - a.mean[c]) * a.inv[c];
        if (c < 8) z[c] = zc; else Z[h][c] = zc;
      }
      bad[h] = bad[h] && a.nan_bad != 0;
      Z[h][10] = 1.0f;
#pragma unroll
      for (int m = 0; m < 8; ++m) {                    // Z[m] = z[sigma^g(m)]: a rotation inside each block of four
        const int b4 = m & 4, r4 = m & 3;
        const float v0 = z[b4 + r4], v1 = z[b4 + ((r4 + 1) & 3)], v2 = z[b4 + ((r4 + 2) & 3)], v3 = z[b4 + ((r4 + 3) & 3)];
        Z[h][m] = g == 0 ? v0 : (g == 1 ? v1 : (g == 2 ? v2 : v3));
      }
    }
    if ((tile + gridDim.x) * kPix < a.npix) load_inputs(tile + gridDim.x);
    f32x4 acc[2];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[h][r] = bias[r];
    const float* wr = wl + g * 16 + i;
    float wc = wr[0], wn = 0.0f;
    pred_chain_prio(wave);
#pragma unroll
    for (int s = 0; s < kStepsOrb; ++s) {
      const float b0 = Z[0][kOrb103.rep[s][0]] * Z[0][kOrb103.rep[s][1]] * Z[0][kOrb103.rep[s][2]];
      const float b1 = Z[1][kOrb103.rep[s][0]] * Z[1][kOrb103.rep[s][1]] * Z[1][kOrb103.rep[s][2]];
      if (s + 1 < kStepsOrb) wn = wr[(s + 1) * 64];
      acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(wc, b0, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(wc, b1, acc[1], 0, 0, 0);
      wc = wn;
      __builtin_amdgcn_sched_barrier(0);
    }
    pred_rest_prio();
    int64_t ostride = a.out_stride;
    int tmax = a.T;
    asm volatile("" : "+s"(ostride), "+s"(tmax));      // keep the addressing inside the loop (see predict103_kernel)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int64_t p = tile * kPix + wave * 32 + h * 16 + i;
      if (p < a.npix) {
        float* orow = a.out + (size_t)(4 * g) * ostride + p;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float v = predict_activation(acc[h][r], a.act);
          if (bad[h]) v = __uint_as_float(0x7fc00000u);
          if (4 * g + r < tmax) orow[(size_t)r * ostride] = v;
        }
      }
    }
  }
}

static void launch_predict103_x16(const PredArgs& a, hipStream_t s) {
  const size_t lds = (size_t)4 * kStepsOrb * 16 * 4;
  constexpr int pix = 32 * kX16Waves;
  int64_t tiles = (a.npix + pix - 1) / pix;
  const int gx = (int)(tiles < 256 * kX16Wgs ? tiles : 256 * kX16Wgs);
  hipLaunchKernelGGL(predict103_x16_kernel, dim3(gx), dim3(64 * kX16Waves), lds, s, a, g_orb_rows_dev);
}

template <int TT>
static void launch_predict103_slices(const PredArgs& a, int slices, hipStream_t s) {
  const size_t lds = (size_t)2 * kStepsOrb2 * TT * 32 * 4;
  static thread_local bool configured = false;
  if (!configured) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(predict103_slice_kernel<TT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipGetLastError();
    configured = true;
  }
  constexpr int pix = 32 * slice_waves(TT);
  int64_t tiles = (a.npix + pix - 1) / pix;
  int gx = 256 * slice_wgs(TT) / slices;       // one workgroup per CU in all (73 / 110 KB of LDS each; T <= 32: slice_wgs)
  if (gx < 1) gx = 1;
  if (tiles < gx) gx = (int)tiles;
  hipLaunchKernelGGL(predict103_slice_kernel<TT>, dim3(gx, slices), dim3(64 * slice_waves(TT)), lds, s, a, g_orb_rows_dev + 4 * kStepsOrb);
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
  const int resident = 256 * pred103_occupancy(TT);
  const int grid = (int)(tiles < resident ? tiles : resident);
  hipLaunchKernelGGL((predict103_kernel<TT, WHOLE>), dim3(grid), dim3(256), lds, s, a);
}

// returns false when the shape is not covered (caller falls back to the generic kernel)
static bool try_predict103(const PredArgs& a, hipStream_t s) {
  if (a.n_in != 10 || a.nfeat != 285 || g_orb_rows_dev == nullptr) return false;
  const int tt = a.ttiles;
#ifndef HSR_PRED_NO_X16
  if (a.T <= 16) {                                        // 16-target tiles (v_mfma_f32_16x16x4_f32)
    launch_predict103_x16(a, s);
    return true;
  }
#endif
#ifdef HSR_PRED_CHUNKED
  if (tt == 1) launch_predict103<1, true>(a, s);          // diagnostic builds: four 4-wave workgroups per CU, each with its own copy of W
#else
  if (tt == 1) launch_predict103_slices<1>(a, 1, s);      // T <= 32: one 16-wave workgroup per CU, W (36.6 KB) staged once per CU (0.209 -> 0.199 ms)
#endif
#ifdef HSR_PRED_CHUNKED
  else if (tt == 2) launch_predict103<2, false>(a, s);    // diagnostic builds: the chunked kernels of rounds 1-2
  else if (tt == 3) launch_predict103<3, false>(a, s);
#endif
  else if (tt <= 16) {                                    // T <= 512: slices of 64 or 96 targets, as few and as narrow as T allows
    const int slices = (tt + 2) / 3;
    const int per = (tt + slices - 1) / slices;           // 32-target tiles per slice: 2 or 3 (1 only for tt == 1)
    if (per <= 2) launch_predict103_slices<2>(a, slices, s);
    else launch_predict103_slices<3>(a, slices, s);
  }
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
  if (g_orb_rows_dev == nullptr) {                     // the orbit orders of the predict103 kernels' K axis, once per process
    std::vector<int16_t> rows((size_t)4 * kStepsOrb + (size_t)2 * kStepsOrb2);
    for (int st = 0; st < kStepsOrb; ++st)
      for (int g = 0; g < 4; ++g) rows[(size_t)4 * st + g] = kOrb103.src[st][g];
    for (int st = 0; st < kStepsOrb2; ++st)
      for (int g = 0; g < 2; ++g) rows[(size_t)4 * kStepsOrb + 2 * st + g] = kOrb2.src[st][g];
    if (hipMalloc(&g_orb_rows_dev, rows.size() * sizeof(int16_t)) != hipSuccess ||
        hipMemcpy(g_orb_rows_dev, rows.data(), rows.size() * sizeof(int16_t), hipMemcpyHostToDevice) != hipSuccess) {
      (void)hipGetLastError();
      g_orb_rows_dev = nullptr;                        // without it the notebook's shape takes the generic kernel
    }
  }
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

// Decomposition of the LDS-panel kernel.  Blocks: every 96 x 96 block of the result above the diagonal of the
// symmetric part (wide), the diagonal blocks of the symmetric part (upper tiles only), plus - when B ends in a strip
// of <= 32 columns - one narrow block per 96 rows of A.  Chunks: rows per chunk as small as fills the resident
// workgroup slots in ONE round (257 workgroups take as long as 512), scaled per kind for equal time; <= 256 chunks,
// as the work buffer is sized for.
static bool gram_lds_plan(int na, int nb, int sym, int64_t n, GramLdsArgs* g) {
  const int nbi = (na + kGpCols - 1) / kGpCols;
  const int rem = nb % kGpCols;
  const bool strip = rem > 0 && rem <= kGpNarrow;
  const int nbj = strip ? nb / kGpCols : (nb + kGpCols - 1) / kGpCols;
  int k = 0;
  auto put = [&](int bi, int bj) {
    if (k < kGpMaxBlocks) {
      g->blocks[k][0] = (uint8_t)bi;
      g->blocks[k][1] = (uint8_t)bj;
    }
    ++k;
  };
  for (int bi = 0; bi < nbi; ++bi)                 // wide: every launched block that is not a diagonal block of the symmetric part
    for (int bj = 0; bj < nbj; ++bj)
      if (!(sym && bj <= bi)) put(bi, bj);
  g->c.nwide = k;
  for (int bi = 0; sym && bi < nbi && bi < nbj; ++bi) put(bi, bi);
  g->c.ndiag = k - g->c.nwide;
  for (int bi = 0; strip && bi < nbi; ++bi) put(bi, nbj);
  g->c.nnarrow = strip ? nbi : 0;
  if (k > kGpMaxBlocks) return false;
  g->c.narrow_col = strip ? nbj * kGpCols : nb;
  g->c.narrow_width = strip ? rem : 0;
  // Rows per chunk by kind, for equal time per workgroup (measured per 8-row batch: wide 1.80 us, narrow 0.72 us - a third of the
  // MFMAs but the same DMA / barrier / address work -, diagonal 1.08 us: 16 of 27 MFMAs per SIMD and k-step):
  // narrow 5/2 and diagonal 5/3 of the rows of a wide block; rows in multiples of 48 keep all three whole batches.
  auto narrow_rows = [](int64_t rows) { return rows / 2 * 5; };
  auto diag_rows = [](int64_t rows) { return rows / 3 * 5; };
  auto count = [&](int64_t rows, int64_t* cw, int64_t* cd, int64_t* cn) {
    *cw = g->c.nwide ? (n + rows - 1) / rows : 0;
    *cd = g->c.ndiag ? (n + diag_rows(rows) - 1) / diag_rows(rows) : 0;
    *cn = g->c.nnarrow ? (n + narrow_rows(rows) - 1) / narrow_rows(rows) : 0;
    return g->c.nwide * *cw + g->c.ndiag * *cd + g->c.nnarrow * *cn;
  };
  // start from the even split and grow until the count fits
  int64_t rows = (int64_t)((double)n * (g->c.nwide + g->c.ndiag * 0.6 + g->c.nnarrow / 2.5) / kGramSlots);
  rows = (rows + 47) / 48 * 48;
  const int64_t floor_rows = g->c.nwide ? 240 : (g->c.ndiag ? 144 : 96);   // >= 240 rows per chunk whatever the kind
  if (rows < floor_rows) rows = floor_rows;
  int64_t cw = 0, cd = 0, cn = 0;
  while (count(rows, &cw, &cd, &cn) > kGramSlots || cw > 256 || cd > 256 || cn > 256) rows += 48;
  g->c.rows_wide = rows;
  g->c.rows_diag = diag_rows(rows);
  g->c.rows_narrow = narrow_rows(rows);
  g->c.chunks_wide = (int32_t)cw;
  g->c.chunks_diag = (int32_t)cd;
  g->c.chunks_narrow = (int32_t)cn;
  g->c.total = (int32_t)(g->c.nwide * cw + g->c.ndiag * cd + g->c.nnarrow * cn);
  g->c.per_xcd = (g->c.total + 7) / 8;
  return true;
}

static int64_t gram_reg_chunks(int64_t n, int64_t* rows_out) {
  int64_t chunks = (n + 1023) / 1024;
  if (chunks > 256) chunks = 256;
  int64_t rows = (n + chunks - 1) / chunks;
  rows = (rows + 15) / 16 * 16;                    // whole k-steps inside every wave's quarter of a chunk
  chunks = (n + rows - 1) / rows;
  if (rows_out) *rows_out = rows;
  return chunks;
}

extern "C" size_t hsr_gram_work_bytes(int32_t na, int32_t nb, int64_t n) {
  if (na < 16 || nb < 16 || n < 1) return 0;
  // both kernels use at most 256 chunks; size for the larger count so that either path can run
  GramLdsArgs g{};
  int64_t c1 = gram_reg_chunks(n, nullptr), c2 = 0;
  if (gram_lds_plan(na, nb, nb >= na, n, &g)) {
    c2 = g.c.chunks_wide > g.c.chunks_narrow ? g.c.chunks_wide : g.c.chunks_narrow;
    if (g.c.chunks_diag > c2) c2 = g.c.chunks_diag;
  }
  GramLdsArgs g0{};                                   // the same matrices as two different pointers: no symmetric skip
  if (gram_lds_plan(na, nb, 0, n, &g0) && g0.c.chunks_wide > c2) c2 = g0.c.chunks_wide;
  const int64_t chunks = c1 > c2 ? c1 : c2;
  return (size_t)chunks * (na / 16) * (nb / 16) * 256 * sizeof(double);
}

extern "C" int hsr_gram_f64(const double* a_dev, int64_t lda, int32_t na, const double* b_dev, int64_t ldb,
                            int32_t nb, int64_t n, double* work_dev, double* c_dev, int64_t ldc,
                            hsr_stream_t stream) {
  HSR_REQUIRE(a_dev && b_dev && work_dev && c_dev && n > 0, HSR_ERR_INVALID, "hsr_gram_f64: bad argument");
  HSR_REQUIRE(na >= 16 && nb >= 16 && na % 16 == 0 && nb % 16 == 0 && lda >= na && ldb >= nb && ldc >= nb,
              HSR_ERR_INVALID, "hsr_gram_f64: na=%d nb=%d must be multiples of 16 inside the leading dimensions", na, nb);
  const int ti = na / 16, tj = nb / 16;
  // Gram of one matrix with itself (the fit: A == B, same leading dimension): the first na columns of the
  // result are symmetric, compute the upper block triangle only
  const int sym = (a_dev == b_dev && lda == ldb && nb >= na) ? 1 : 0;
  hipStream_t s = (hipStream_t)stream;
  const bool dma_ok = (lda % 2 == 0) && (ldb % 2 == 0) && (((uintptr_t)a_dev | (uintptr_t)b_dev) & 15) == 0;
  GramLdsArgs g{};
  if (dma_ok && gram_lds_plan(na, nb, sym, n, &g)) {
    g.c.A = a_dev;
    g.c.B = b_dev;
    g.c.lda = lda;
    g.c.ldb = ldb;
    g.c.n = n;
    g.c.na = na;
    g.c.nb = nb;
    g.c.tiles_i = ti;
    g.c.tiles_j = tj;
    g.c.partials = work_dev;
#ifdef HSR_GRAM_STAMPS
    g.c.stamps = g_gram_stamps;
#endif
    static std::once_flag lds_once;
    constexpr size_t lds_bytes = (size_t)kGramLdsDoubles * sizeof(double);
    std::call_once(lds_once, [] {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gram_f64_lds_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds_bytes);
    });
    hipLaunchKernelGGL(gram_f64_lds_kernel, dim3(8 * (unsigned)g.c.per_xcd), dim3(kGramThreads), lds_bytes, s, g);
    HSR_LAUNCH_CHECK("gram_f64_lds_kernel");
    hipLaunchKernelGGL(gram_reduce_kernel, dim3(ti * tj), dim3(256), 0, s, work_dev, ti * tj, g.c.chunks_wide, tj, sym,
                       kGpCols / 16, g.c.narrow_col / 16, g.c.chunks_narrow, g.c.chunks_diag, c_dev, ldc);
    HSR_LAUNCH_CHECK("gram_reduce_kernel");
    return HSR_OK;
  }
  int64_t rows = 0;
  const int64_t chunks = gram_reg_chunks(n, &rows);
  constexpr int R = hsr::kGramR;
  hipLaunchKernelGGL(gram_f64_kernel, dim3(((ti + R - 1) / R) * ((tj + R - 1) / R), (unsigned)chunks), dim3(256), 0, s,
                     a_dev, lda, ti, b_dev, ldb, tj, n, rows, sym, work_dev);
  hipLaunchKernelGGL(gram_reduce_kernel, dim3(ti * tj), dim3(256), 0, s, work_dev, ti * tj, (int)chunks, tj, sym, R, tj,
                     (int)chunks, 0, c_dev, ldc);
  HSR_LAUNCH_CHECK("gram_f64_kernel");
  return HSR_OK;
}

// ------------------------------------------------------------------------------------------------
// The small steps of PolyRidge.fit around Gram and Cholesky, as three kernels instead of ~40 torch launches
// (200 us of a 700 us fit): StandardScaler statistics, assembly of the ridge system, model read-out.
// ------------------------------------------------------------------------------------------------
constexpr int kStatsBlocks = 64;

// per-block sums of d = x - K and d^2 (K = the column's first sample: the "shifted data" form, so that
// M2 = sum d^2 - (sum d)^2 / n loses nothing to a large mean); fixed order inside the block: every thread walks its rows
// once with all columns in registers, lanes are joined by the xor butterfly, the four waves in wave order
__global__ __launch_bounds__(256) void ridge_stats_partial_kernel(const float* __restrict__ x, int64_t x_rs, int64_t x_cs,
                                                                  int64_t n, int n_in, double* __restrict__ work) {
  __shared__ double red[4][32];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int64_t rows = (n + gridDim.x - 1) / gridDim.x;
  const int64_t r0 = (int64_t)blockIdx.x * rows;
  const int64_t r1 = r0 + rows < n ? r0 + rows : n;
  double s1[16], s2[16], K[16];
#pragma unroll
  for (int c = 0; c < 16; ++c) {
    s1[c] = s2[c] = 0.0;
    K[c] = c < n_in ? (double)x[c * x_cs] : 0.0;
  }
  // four rows per pass with all their loads issued before the first sum (a block walks <= 1024 rows: one exposed round trip
  // instead of four; r03 trace: 11.9 us for 1.2 MB); the sums keep their row order
  for (int64_t rb = r0 + t; rb < r1; rb += 4 * 256) {
    float v[4][16];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int64_t r = rb + q * 256;
#pragma unroll
      for (int c = 0; c < 16; ++c)
        if (c < n_in && r < r1) v[q][c] = x[r * x_rs + c * x_cs];
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (rb + q * 256 >= r1) break;
#pragma unroll
      for (int c = 0; c < 16; ++c)
        if (c < n_in) {
          const double d = (double)v[q][c] - K[c];
          s1[c] += d;
          s2[c] += d * d;
        }
    }
  }
#pragma unroll
  for (int c = 0; c < 16; ++c) {
    if (c < n_in) {
      const double a1 = wave_sum(s1[c]), a2 = wave_sum(s2[c]);
      if (lane == 0) {
        red[wave][2 * c] = a1;
        red[wave][2 * c + 1] = a2;
      }
    }
  }
  __syncthreads();
  if (t < 2 * n_in) work[(size_t)blockIdx.x * n_in * 2 + t] = ((red[0][t] + red[1][t]) + red[2][t]) + red[3][t];
}

// the blocks' sums joined (one wave per column, a lane per block, xor butterfly: a fixed order) -> [n, mean.., M2..] (the
// layout of PolyRidge.local_stats), mean and scale (zero variance -> 1)
__global__ __launch_bounds__(1024) void ridge_stats_finish_kernel(const float* __restrict__ x, int64_t x_cs, int64_t n, int n_in,
                                                                  int nblocks, const double* __restrict__ work,
                                                                  double* __restrict__ stats, double* __restrict__ mean_out,
                                                                  double* __restrict__ scale_out) {
  const int c = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (threadIdx.x == 0) stats[0] = (double)n;
  if (c >= n_in) return;
  const double s1 = wave_sum(lane < nblocks ? work[((size_t)lane * n_in + c) * 2] : 0.0);
  const double s2 = wave_sum(lane < nblocks ? work[((size_t)lane * n_in + c) * 2 + 1] : 0.0);
  if (lane != 0) return;
  const double K = (double)x[c * x_cs];
  const double mean = K + s1 / (double)n;
  double m2 = s2 - s1 * s1 / (double)n;
  if (m2 < 0.0) m2 = 0.0;
  stats[1 + c] = mean;
  stats[1 + n_in + c] = m2;
  mean_out[c] = mean;
  const double sc = sqrt(m2 / (double)n);
  scale_out[c] = sc == 0.0 ? 1.0 : sc;
}

// G (na, na + tp) = [1 | Phi]^T [1 | Phi | Y]  ->  A = Phi_c^T Phi_c + alpha I padded to npad with an identity block,
// B = Phi_c^T (Y - ybar) padded with zero rows (the centred normal equations of Ridge(fit_intercept=True)):
//   A_ij = G[1+i][1+j] - s_i s_j / cnt (+ alpha on the diagonal),  B_it = G[1+i][na+t] - s_i ybar_t,
//   s = G[0][1..nf] (column sums), cnt = G[0][0], ybar_t = G[0][na+t] / cnt.   Also clears the Cholesky status word.
__global__ __launch_bounds__(256) void ridge_assemble_kernel(const double* __restrict__ G, int64_t ldg, int na, int nf, int T,
                                                             double alpha, double* __restrict__ A, int npad,
                                                             double* __restrict__ Bm, int64_t ldb, int32_t* __restrict__ info) {
  const double cnt = G[0];
  const int64_t total = (int64_t)npad * (npad + T);
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int i = (int)(e / (npad + T)), j = (int)(e % (npad + T));
    if (j < npad) {
      double v = i == j ? 1.0 : 0.0;
      if (i < nf && j < nf) {
        v = G[(size_t)(1 + i) * ldg + 1 + j] - (G[1 + i] * G[1 + j]) / cnt;
        if (i == j) v += alpha;
      }
      A[(size_t)i * npad + j] = v;
    } else {
      const int t = j - npad;
      double v = 0.0;
      if (i < nf) v = G[(size_t)(1 + i) * ldg + na + t] - G[1 + i] * (G[na + t] / cnt);
      Bm[(size_t)i * ldb + t] = v;
    }
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) *info = 0;
}

// W (nf, T) float64 solution -> intercept b = ybar - (s / cnt) . W (float64 and float32), W as float32 with a zero row up
// to kpad, mean and 1 / scale as float32: everything the predict kernels read.  A block owns 32 targets: s / cnt goes to
// LDS once, eight thread groups take every eighth feature (coalesced over the targets), partial sums joined in group
// order.  (First version: one thread per target walking all features with a division per step - 65 us for T = 32.)
__global__ __launch_bounds__(256) void ridge_finish_kernel(const double* __restrict__ G, int na, int nf, int T,
                                                           const double* __restrict__ Wm, int64_t ldw, const double* __restrict__ mean,
                                                           const double* __restrict__ scale, int n_in, int kpad,
                                                           double* __restrict__ b64, float* __restrict__ b32,
                                                           float* __restrict__ W32, float* __restrict__ mean32,
                                                           float* __restrict__ inv32) {
  __shared__ double sc[kMaxFeat];
  __shared__ double part[8][32];
  const double cnt = G[0];
  const int tid = threadIdx.x;
  for (int f = tid; f < nf; f += 256) sc[f] = G[1 + f] / cnt;
  __syncthreads();
  const int grp = tid >> 5, tt = tid & 31;
  const int t = blockIdx.x * 32 + tt;
  double acc = 0.0;
  if (t < T)
    for (int f = grp; f < nf; f += 8) acc += sc[f] * Wm[(size_t)f * ldw + t];
  part[grp][tt] = acc;
  __syncthreads();
  if (grp == 0 && t < T) {
    double a = part[0][tt];
#pragma unroll
    for (int g = 1; g < 8; ++g) a += part[g][tt];
    const double b = G[na + t] / cnt - a;
    b64[t] = b;
    b32[t] = (float)b;
  }
  const int64_t gid = (int64_t)blockIdx.x * 256 + tid, gsz = (int64_t)gridDim.x * 256;
  for (int64_t e = gid; e < (int64_t)kpad * T; e += gsz) {
    const int f = (int)(e / T), tq = (int)(e % T);
    W32[e] = f < nf ? (float)Wm[(size_t)f * ldw + tq] : 0.0f;
  }
  for (int64_t c = gid; c < n_in; c += gsz) {
    mean32[c] = (float)mean[c];
    inv32[c] = (float)(1.0 / scale[c]);
  }
}

extern "C" size_t hsr_ridge_stats_work_bytes(int32_t n_in) {
  return n_in >= 1 && n_in <= 16 ? (size_t)kStatsBlocks * n_in * 2 * sizeof(double) : 0;
}

extern "C" int hsr_ridge_stats(const float* x_dev, int64_t x_rs, int64_t x_cs, int64_t n, int32_t n_in, double* work_dev,
                               double* stats_dev, double* mean_dev, double* scale_dev, hsr_stream_t stream) {
  HSR_REQUIRE(x_dev && work_dev && stats_dev && mean_dev && scale_dev, HSR_ERR_INVALID, "hsr_ridge_stats: NULL pointer");
  HSR_REQUIRE(n >= 1 && n_in >= 1 && n_in <= 16, HSR_ERR_INVALID, "hsr_ridge_stats: n=%lld n_in=%d", (long long)n, n_in);
  int nblocks = (int)((n + 1023) / 1024);
  if (nblocks > kStatsBlocks) nblocks = kStatsBlocks;
  hipLaunchKernelGGL(ridge_stats_partial_kernel, dim3(nblocks), dim3(256), 0, (hipStream_t)stream, x_dev, x_rs, x_cs, n, n_in,
                     work_dev);
  static_assert(kStatsBlocks <= 64, "one lane per block");
  hipLaunchKernelGGL(ridge_stats_finish_kernel, dim3(1), dim3(64 * n_in), 0, (hipStream_t)stream, x_dev, x_cs, n, n_in, nblocks,
                     work_dev, stats_dev, mean_dev, scale_dev);
  HSR_LAUNCH_CHECK("ridge_stats_kernel");
  return HSR_OK;
}

extern "C" int hsr_ridge_assemble(const double* g_dev, int64_t ldg, int32_t na, int32_t nf, int32_t T, double alpha,
                                  double* a_dev, int32_t npad, double* b_dev, int64_t ldb, int32_t* info_dev,
                                  hsr_stream_t stream) {
  HSR_REQUIRE(g_dev && a_dev && b_dev && info_dev, HSR_ERR_INVALID, "hsr_ridge_assemble: NULL pointer");
  HSR_REQUIRE(nf >= 1 && na >= nf + 1 && npad >= nf && T >= 1 && ldg >= na + T && ldb >= T, HSR_ERR_INVALID,
              "hsr_ridge_assemble: bad shape (na=%d nf=%d npad=%d T=%d)", na, nf, npad, T);
  const int64_t total = (int64_t)npad * (npad + T);
  const int grid = (int)((total + 255) / 256 < 1024 ? (total + 255) / 256 : 1024);
  hipLaunchKernelGGL(ridge_assemble_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, g_dev, ldg, na, nf, T, alpha, a_dev,
                     npad, b_dev, ldb, info_dev);
  HSR_LAUNCH_CHECK("ridge_assemble_kernel");
  return HSR_OK;
}

extern "C" int hsr_ridge_finish(const double* g_dev, int32_t na, int32_t nf, int32_t T, const double* w_dev, int64_t ldw,
                                const double* mean_dev, const double* scale_dev, int32_t n_in, int32_t kpad,
                                double* b64_dev, float* b32_dev, float* w32_dev, float* mean32_dev, float* inv32_dev,
                                hsr_stream_t stream) {
  HSR_REQUIRE(g_dev && w_dev && mean_dev && scale_dev && b64_dev && b32_dev && w32_dev && mean32_dev && inv32_dev,
              HSR_ERR_INVALID, "hsr_ridge_finish: NULL pointer");
  HSR_REQUIRE(nf >= 1 && na >= nf + 1 && T >= 1 && ldw >= T && kpad >= nf && n_in >= 1, HSR_ERR_INVALID,
              "hsr_ridge_finish: bad shape");
  HSR_REQUIRE(nf <= kMaxFeat, HSR_ERR_UNSUPPORTED, "hsr_ridge_finish: nf=%d > %d", nf, kMaxFeat);
  int grid = (T + 31) / 32;                       // a block per 32 targets; the float32 copies ride along grid-strided,
  const int64_t cpy = ((int64_t)kpad * T + 1023) / 1024;   // so there are at least enough blocks for 4 elements per thread
  if (cpy > grid) grid = (int)(cpy < 64 ? cpy : 64);
  hipLaunchKernelGGL(ridge_finish_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, g_dev, na, nf, T, w_dev, ldw, mean_dev,
                     scale_dev, n_in, kpad, b64_dev, b32_dev, w32_dev, mean32_dev, inv32_dev);
  HSR_LAUNCH_CHECK("ridge_finish_kernel");
  return HSR_OK;
}

extern "C" int hsr_polyfeat_predict(const float* x_dev, int64_t x_ps, int64_t x_cs, const float* mean_dev,
                                    const float* inv_scale_dev, int64_t npix, int32_t n_in, int32_t degree,
                                    const float* w_dev, int64_t ldw, const float* bias_dev, int32_t T,
                                    int32_t activation, float* out_dev, int64_t out_stride, hsr_stream_t stream) {
  return hsr_polyfeat_predict_cube(x_dev, x_ps, x_cs, mean_dev, inv_scale_dev, npix, n_in, degree, w_dev, ldw, bias_dev, T,
                                   activation, 0, 0.0f, 0, out_dev, out_stride, stream);
}

extern "C" int hsr_polyfeat_predict_cube(const float* x_dev, int64_t x_ps, int64_t x_cs, const float* mean_dev,
                                         const float* inv_scale_dev, int64_t npix, int32_t n_in, int32_t degree,
                                         const float* w_dev, int64_t ldw, const float* bias_dev, int32_t T,
                                         int32_t activation, int32_t nan_bad_pixels, float nodata, int32_t use_nodata,
                                         float* out_dev, int64_t out_stride, hsr_stream_t stream) {
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
  a.nan_bad = nan_bad_pixels != 0;
  a.use_nodata = use_nodata != 0;
  a.nodata = nodata;
  a.out = out_dev;
  a.out_stride = out_stride;
  if (degree == 3 && try_predict103(a, (hipStream_t)stream)) {
    HSR_LAUNCH_CHECK("predict103_kernel");
    return HSR_OK;
  }
  const size_t lds = ((size_t)kPredPix * (a.kpad + 1) + (size_t)kPredPix * (n_in + 1) + kPredPix) * sizeof(float);
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
