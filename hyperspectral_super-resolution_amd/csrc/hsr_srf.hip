// K1 (+ fused K2): SRF band integration of an (npix, B) float32 cube on gfx950.
//
// Replaces the hot loop of pseudo_s2_srf_integral (reference s2_emit/synth.py:32-43).  The reference
// makes 13 full-cube float64 passes; here the cube is streamed from HBM exactly once.
//
// Data flow per workgroup (512 threads = 8 waves, 2 workgroups resident per CU):
//   1. a group of 64 consecutive pixels (64*B*4 bytes, one linear 16-byte-aligned slab because the
//      cube is pixel-major) goes HBM -> LDS with global_load_lds_dwordx4 (no VGPR round trip,
//      1 KiB per wave instruction, fully coalesced), marked non-temporal: the cube is read once,
//      and without the hint the stream thrashes L2 against the output lines (-12 % on the kernel).
//      ~73 KB in flight per workgroup.
//   2. one linear ds_read_b128 sweep flags pixels holding a non-finite sample.
//   3. lane = pixel, wave = band group: each band is a short dot product over its SRF support read
//      from LDS with a row stride of B words (B odd -> bank-conflict free).  The weight taps of all
//      bands (a few hundred floats, 16-byte aligned segments) are staged into LDS once per
//      workgroup and read as broadcast ds_read_b128 - no scalar-load latency inside the group loop
//      (the first version fetched them with s_load per group and spent ~5 us per group waiting).
//      Flagged pixels take the dense product so that 0*Inf -> NaN poisons exactly the bands the
//      reference poisons (synth.py:41 multiplies all B samples of every band).
//   4. output.  Pixel-major (band-last) output is staged in LDS as the group's contiguous
//      [pixel][band] slab and flushed with 16-byte stores in the next iteration, after that group's
//      DMA has been issued: one contiguous ~3 KB write per group.  (Band-major planes are stored directly: nb scattered
//      256-B segments per group; measured 10 % slower on the whole kernel because of the write
//      pattern, although the planes are only 4 % of the bytes.)  With DEG > 0 the same lane
//      also accumulates the Vandermonde power sums of (x = plane value, y = real S2 value) in
//      float64 registers; they are reduced over the wave by a fixed DPP tree and written to the
//      partial slot of the work unit (no float atomics -> bitwise reproducible).
// Work units.  A tile of G pixel groups owns S = min(G, resident workgroups) partial slots; unit (tile, s)
// is the group sequence s, s+S, s+2S, ... accumulated in that order into slot s.  A single-tile launch
// runs one unit per workgroup (grid = S).  A batch launch (hsr_srf_integrate_moments_batched) walks a
// device table of units over many small tiles - workgroup b takes units b, b+grid, ... - and flushes the
// finished unit's sums while the next group's DMA is in flight; per-tile results are bit-identical to the
// single-tile launch because a tile's slot layout and every summation order depend on its npix only.
// HBM-bound by construction: 4*B bytes in, 4*nb (+4*nb+1) bytes out/in per pixel, ~0.35 kflop.
#include "hsr_common.h"
#include "hsr_solve.h"
#include "hsr_sync_dev.h"

namespace hsr {

struct SrfBands {
  int32_t k0[HSR_MAX_BANDS];
  int32_t klen[HSR_MAX_BANDS];
  int32_t woff[HSR_MAX_BANDS];  // offset (floats, multiple of 4) of the band's taps in the LDS weight area
  int8_t band_of[8][2];         // band handled by (group, slot), -1 = none: balanced over the 8 groups (srf_common)
};

constexpr int kWeightCap = 1024;  // floats of LDS reserved for compact weight taps (4 KiB)

// One (tile, slot) work unit: the public hsr_batch_unit record (include/hsr.h), 64 bytes.
typedef hsr_batch_unit SrfUnit;
static_assert(sizeof(SrfUnit) == 64 && sizeof(hsr_batch_tile) == 64, "batch records are 64 bytes");

// Fused fit (hsr_fused_fit, single-tile launches): the workgroup that completes a group of slots reduces it, the one
// that completes the last group reduces the groups, solves and writes moments + coefficients (fused_fit below).
struct SrfFit {
  double* gpart;      // [64][nb][M] group sums
  int32_t* tickets;   // [65] arrival counters, zero between launches; NULL: no fused fit
  double* moments;    // [nb][M]
  double* coeffs;     // [nb][deg+1]
  long long min_count;
};

struct SrfArgs {
  SrfFit fit;
  SrfUnit one;            // single-tile launch: the tile (slot = blockIdx.x, part_dev = base of slot 0); partials [slot][band][moment]
  const SrfUnit* units;   // batch launch: the unit table (device)
  int32_t nunits;
  int32_t B;
  int32_t ldsB;  // LDS row stride in words (odd)
  int32_t wtaps; // floats used in the LDS weight area (0: weights do not fit, read them from global)
  const float* wn;
  SrfBands bands;
  int32_t nb;
  // element (b, p) of the output at pseudo_dev[b * out_bs + p * out_ps], of the target at real_dev[b * real_bs + p * real_ps]
  int64_t out_bs, out_ps;
  int64_t real_bs, real_ps;
  float min_x, min_y;
  // uint16 tiles (srf_u16_kernel): cube points at uint16 samples, x = float(u) * scale, u == nodata -> NaN
  int32_t u16;
  int32_t u16_fast;  // opt-in fast arithmetic of the uint16 kernels (hsr_srf_options.flags & HSR_SRF_U16_FAST)
  float scale;
  uint32_t nodata;   // > 0xffff: no nodata value
  // K3 of an OLDER tile riding in this launch (APPLY variants, hsr_srf_integrate_moments_apply): matched = poly(x) over
  // apply_npix pixels of pixel-major rows with this launch's row length (out_ps) and band count
  const float* apply_x;
  float* apply_out;
  const double* apply_coeffs;       // (nb, DEG + 1) float64, highest power first
  const uint8_t* apply_mask;        // polynomial only where != 0 (NULL: everywhere)
  int64_t apply_npix;
  int32_t apply_clip;
  // the fit of the PREVIOUS tile in this launch's tail (hsr_apply_job.fit_*): the first nb workgroups to finish their groups
  // take a ticket each and reduce + solve one band of the previous launch's partial slots
  const double* lazy_partials;      // [lazy_slots][nb][3 DEG + 2]; NULL: no tail fit
  int32_t lazy_slots;
  long long lazy_min_count;
  double* lazy_moments;             // (nb, 3 DEG + 2) out
  double* lazy_coeffs;              // (nb, DEG + 1) out
  unsigned int* lazy_counter;       // running ticket counter (one ticket per workgroup and launch)
  unsigned int lazy_base;           // its value before this launch
  // exchange pipelines (hsr_pipeline_create_exchange): the fit crosses to ANOTHER queue between the slot reduction and the solve
  // (all-reduce of the moments over the ranks), so the tail stops at the moments - written through to memory - and counts every
  // finished band into *lazy_ready, which a one-wave gate kernel on the side stream polls; and the K3 pre-phase reads
  // coefficients that a side-stream kernel published by setting *apply_ready to apply_ready_value
  unsigned int* lazy_ready;         // non-NULL: reduce only, publish (no solve)
  // a fit over a GROUP of T tiles (a mosaic held by one GPU, hsr_apply_job.fit_group_*): lazy_moments is entry lazy_group_index of
  // lazy_group_moments [T][nb][M]; the tail of the group's LAST tile also adds the T entries (the tree of hsr_moments_reduce
  // over T slots) and solves: lazy_group_total, lazy_coeffs.  The other tiles' tails write no coefficients.
  int32_t lazy_group_T;             // 0 / 1: every tile has its own fit
  int32_t lazy_group_index;
  const double* lazy_group_moments;
  double* lazy_group_total;
  const unsigned int* apply_ready;  // non-NULL: poll until (int)(*apply_ready - apply_ready_value) >= 0, then read the coefficients through
  unsigned int apply_ready_value;
  unsigned int* sync_error;         // set to a non-zero code if a poll runs into its time limit (HSR_SYNC_TIMEOUT_S)
#ifdef HSR_PHASE_STAMPS
  unsigned long long* stamps;
  unsigned long long* stamps2;   // [grid][4]: REFCLK (100 MHz) at workgroup entry and exit, XCC id, HW_ID
  uint32_t* stamps3;             // [grid][64]: REFCLK ticks of each of the workgroup's first 64 group iterations
#endif
};

#ifdef HSR_PHASE_STAMPS
// Diagnostic build only (tools/k1_lab): per-phase s_memtime stamps, written to a buffer nothing else
// reads.  Never compiled into libhsr_mi355x.so.
unsigned long long* g_stamp_buffer = nullptr;
unsigned long long* g_stamp_buffer2 = nullptr;
uint32_t* g_stamp_buffer3 = nullptr;
__device__ __forceinline__ unsigned long long real_time() {   // constant 100 MHz counter, the same on every XCD
  unsigned long long t;
  asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  return t;
}
__device__ __forceinline__ unsigned long long phase_stamp() {
  unsigned long long t;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  return t;
}
#define HSR_STAMP(var) const unsigned long long var = phase_stamp()
#else
#define HSR_STAMP(var)
#endif

constexpr int kGroups = 8;                          // band groups per workgroup (band = group + 8 * slot)
constexpr int kBandSlots = HSR_MAX_BANDS / kGroups;  // 2 bands per thread
constexpr int kTapChunk = 16;                       // taps per unrolled dot-product chunk
constexpr int kScanBatch = 5;                       // ds_read_b128 in flight per thread in the uint16 sweep
#ifndef HSR_F32_SCAN_BATCH
#define HSR_F32_SCAN_BATCH 5
#endif
constexpr int kScanBatchF32 = HSR_F32_SCAN_BATCH;   // ... in the float32 sweep

// Group geometry: P = 64 pixels per LDS group, 512 threads (8 waves), lane = pixel, wave = band group; 2 workgroups
// per CU: 16 waves (4 per SIMD, <= 128 VGPRs) and ~146 KB of LDS groups per CU.
// (Measured and dropped: P = 8, one-wave workgroups with wave-local barriers, ~14 per CU: 0.2535 ms; P = 128, one
//  1024-thread workgroup per CU: 0.234 ms; P = 32, 256 threads x 4 per CU: 3-8 % slower, and its per-half-wave band
//  groups kept the band parameters in VGPRs; P = 64: 0.2133 ms on the same box.)
// The geometry comes with every call (hsr_srf_options); the library keeps no tuning state.
struct SrfTuning {
  int tile_pixels;    // 64
  int reserved_cus;   // CUs left without a persistent K1 workgroup (side-stream kernels of the previous tile's fit)
  bool u16_ring;      // uint16 cubes: double-buffered kernel where it fits
  bool u16_fast;      // uint16 cubes: fast arithmetic (decode scale folded into the weights, packed fma)
};

static int srf_tuning(const hsr_srf_options* o, SrfTuning* t, const char* who) {
  t->tile_pixels = 64;
  t->reserved_cus = 0;
  t->u16_ring = true;
  t->u16_fast = false;
  if (o == nullptr) return HSR_OK;
  HSR_REQUIRE(o->tile_pixels == 0 || o->tile_pixels == 64, HSR_ERR_INVALID,
              "%s: options.tile_pixels must be 0 (default) or 64, got %d (the 32-pixel geometry of round 1 measured 3-8 %% "
              "slower and was removed)", who, o->tile_pixels);
  HSR_REQUIRE(o->reserved_cus >= 0 && o->reserved_cus <= 128, HSR_ERR_INVALID, "%s: options.reserved_cus=%d outside [0,128]",
              who, o->reserved_cus);
  HSR_REQUIRE((o->flags & ~HSR_SRF_U16_FAST) == 0, HSR_ERR_INVALID, "%s: options.flags has unknown bits (0x%x)", who, o->flags);
  if (o->tile_pixels) t->tile_pixels = o->tile_pixels;
  t->reserved_cus = o->reserved_cus;
  t->u16_ring = o->u16_single_buffer == 0;
  t->u16_fast = (o->flags & HSR_SRF_U16_FAST) != 0;
  return HSR_OK;
}

// Partial slots of a tile of npix pixels: min(pixel groups, resident workgroups).
static int srf_slots(int64_t npix, int P, int reserved_cus) {
  int64_t groups = (npix + P - 1) / P;
  if (groups < 1) groups = 1;
  const int64_t cap = (int64_t)(256 - reserved_cus) * (P == 64 ? 2 : 4);  // CUs x resident workgroups
  // Up to 64 groups: a slot per group.  65 .. 512 groups (tiles up to ~180 x 180, whose stand-alone step is launch-bound
  // anyway): four groups per slot, at least 64 slots - so that a batch of small tiles does not flush and re-read one
  // 1.3 KB partial per 64-pixel group (a 100 x 100 tile: 64 slots of 2-3 groups instead of 157 of one; batched K1+K2 of
  // 256 such tiles 5 % faster on float32, 13-16 % on uint16 tiles).  More than 512 groups: one slot per resident workgroup
  // as before (coarsening there costs a stand-alone 256 x 256 tile 8 %).
  int64_t slots = groups;
  if (groups > 64 && groups <= 512) {
    slots = (groups + 3) / 4;
    if (slots < 64) slots = 64;
  }
  return (int)(slots < cap ? slots : cap);
}

typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// Workgroup barrier that orders LDS accesses only: s_waitcnt lgkmcnt(0) (leaves vmcnt untouched, so
// the plane stores just issued are not waited for) + s_barrier.
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// Fit targets (2 float32 per thread + the pixel's mask byte) travel HBM -> LDS by per-lane LDS-DMA, issued right behind
// the group's own DMA and read back with ordinary ds_reads after the group barrier.  Round 1/2 loaded them with inline-asm
// global_load_dword / _ubyte (invisible to hipcc's waitcnt pass, which otherwise drains vmcnt(0) between ordinary loads
// and LDS-DMA: +20 us on the kernel) and retired them with a hand-written s_waitcnt - correct only as long as the
// compiler placed no copy or spill of the "=v" result between the load and the wait (one such copy was seen: a masked
// tile's moments changed from launch to launch).  An LDS-DMA has no destination register, so there is no value the
// compiler could move: the reads are ordinary ds_reads placed behind the group barrier (whose __syncthreads() carries
// s_waitcnt vmcnt(0)) and an explicit wait.  Lane i's dword lands at stage + 4 i.
// The mask byte travels as the aligned dword that holds it (a byte-sized LDS-DMA packs its 64 bytes instead of using
// dword slots - measured: the first version read 97 % of a 79 % mask as set); staged_u8 picks the byte out again.
__device__ __forceinline__ void stage_f32(const float* p, float* stage_wave_uniform) {
  __builtin_amdgcn_global_load_lds((gptr_t)p, (lptr_t)stage_wave_uniform, 4, 0, 0);
}
__device__ __forceinline__ void stage_u8(const uint8_t* p, uint32_t* stage_wave_uniform) {
  const uintptr_t a = reinterpret_cast<uintptr_t>(p);
  __builtin_amdgcn_global_load_lds((gptr_t)(a & ~(uintptr_t)3), (lptr_t)stage_wave_uniform, 4, 0, 0);   // same dword: never leaves the buffer's page
}
__device__ __forceinline__ uint32_t staged_u8(const uint8_t* p, const uint32_t* stage, int lane) {
  return (stage[lane] >> ((reinterpret_cast<uintptr_t>(p) & 3) * 8)) & 0xffu;
}
constexpr int kTargetStageBytesPerBand = 64 * 4;   // ystage[band][64 lanes]; + 64 dwords for the mask bytes
__host__ __device__ constexpr size_t target_stage_bytes(int nb) { return (size_t)nb * kTargetStageBytesPerBand + 64 * 4; }

// ---- batch launches: the record of a workgroup's NEXT unit travels HBM -> LDS by a 64-byte LDS-DMA issued by 16
// lanes of wave 0 right behind the group's own DMA: no destination register (an inline-asm load into a VGPR that is
// used an iteration later can be copied by the compiler before it has landed), no SGPRs held across the iteration
// (a prefetched record in SGPRs cost 16 of them at the SGPR limit and pushed a dozen VGPRs into scratch).  It is
// read back with wave-uniform ds_reads + v_readfirstlane when the current unit ends.
__device__ __forceinline__ const char* unit_record_addr(const SrfUnit* units, int64_t idx, int nunits, int lane) {
  const int64_t i = idx < nunits ? idx : (int64_t)nunits - 1;      // clamped: always a valid address
  return reinterpret_cast<const char*>(units + i) + (lane & 15) * 4;
}
__device__ __forceinline__ SrfUnit unit_from_lds(const uint32_t* us) {
  uint32_t w[15];
#pragma unroll
  for (int i = 0; i < 15; ++i) w[i] = __builtin_amdgcn_readfirstlane(us[i]);
  SrfUnit u;
  u.cube_dev = reinterpret_cast<const void*>(((uint64_t)w[1] << 32) | w[0]);
  u.real_dev = reinterpret_cast<const float*>(((uint64_t)w[3] << 32) | w[2]);
  u.mask_dev = reinterpret_cast<const uint8_t*>(((uint64_t)w[5] << 32) | w[4]);
  u.pseudo_dev = reinterpret_cast<float*>(((uint64_t)w[7] << 32) | w[6]);
  u.part_dev = reinterpret_cast<double*>(((uint64_t)w[9] << 32) | w[8]);
  u.npix = (int64_t)(((uint64_t)w[11] << 32) | w[10]);
  u.slots = (int32_t)w[12];
  u.slot = (int32_t)w[13];
  u.ngroups = (int32_t)w[14];
  u.reserved = 0;
  return u;
}

// x*0 is NaN exactly when x is NaN or +-Inf; four of them chained cost 4 VALU ops.
__device__ __forceinline__ bool any_nonfinite4(const float4& v) {
  float z = v.x * 0.0f;
  z = fmaf(v.y, 0.0f, z);
  z = fmaf(v.z, 0.0f, z);
  z = fmaf(v.w, 0.0f, z);
  return z != z;
}

// Row length (floats) of the output slab staged in LDS: the image's row when it has no pad columns, the band count otherwise - the
// pad columns are not staged, the flush writes them as zeros (hsr.h: padded rows are owned by the callee).  13 Sentinel-2 bands in
// rows of 16 floats: 768 B less LDS per workgroup, which is what keeps TWO workgroups per CU at B = 285 (r03: 82 368 B with the
// padded slab, one workgroup per CU, K1 0.317 ms instead of 0.238); and an odd row stride is bank-conflict free.
__host__ __device__ __forceinline__ int stage_row(int nb, int ops) { return ops > nb ? nb : ops; }

// Flush the staged [pixel][band] slab of the previous group: contiguous in HBM, 16 bytes per lane.
template <int T>
__device__ __forceinline__ void flush_stage(const float* ostage, float* out, int64_t pix0, int npx, int nb, int ops, int t) {
  const int n4 = (npx * ops) >> 2;  // ops is a multiple of 4
  float4* dst = reinterpret_cast<float4*>(out + pix0 * ops);
  if (ops == nb) {                  // no pad columns: the slab is the rows
    const float4* src = reinterpret_cast<const float4*>(ostage);
    for (int i = t; i < n4; i += T) st_stream(dst + i, src[i]);
    return;
  }
  const int q = ops >> 2;           // float4 per row: 1 .. 4
  for (int i = t; i < n4; i += T) {
    const int px = q == 4 ? i >> 2 : (q == 2 ? i >> 1 : (q == 1 ? i : i / 3));
    const int c = (i - px * q) * 4;
    const float* r = ostage + px * nb + c;
    st_stream(dst + i, make_float4(c < nb ? r[0] : 0.0f, c + 1 < nb ? r[1] : 0.0f, c + 2 < nb ? r[2] : 0.0f, c + 3 < nb ? r[3] : 0.0f));
  }
}

// ---- fixed-order sums of the per-lane power sums over the pixels of a group, without LDS ---------------------------
// Every kernel adds the 64 per-pixel values of a group in the same binary tree over the PIXEL index p:
// pairs (p, p ^ 32), then ^ 16, ^ 8, ^ 4, ^ 2, ^ 1 - so float32 K1, uint16 K1 and the batch kernels give the same bits.
//   float32 kernels: lane = pixel, lane distances 32, 16, 8, 4, 2, 1.
//   uint16 kernels:  lanes 0..31 hold the even pixels, 32..63 the odd ones (see srf_u16_kernel): p ^ 32 is lane ^ 16,
//                    ..., p ^ 2 is lane ^ 1 and p ^ 1 is lane ^ 32: lane distances 16, 8, 4, 2, 1, 32.
// A wave holds N = 2 * M such values (two bands x M moments).  Reduced one by one that is N x 6 x (2 v_mov_dpp +
// v_add_f64) = 400-500 VALU instructions per wave; a batch of small tiles reduces once per 64-pixel group, all 8
// waves at once, and that was 37 % of the batch kernel (0.74 ms against 0.47 ms without any flush; neither fewer
// stores nor more ILP helped: it is VALU issue).  So the tree is evaluated MERGED: after the level with lane distance D
// only half of the lanes of a register carry a needed result (lanes l and l ^ D hold the same sum), so two registers
// a, b are combined into one - lanes with (l & D) == 0 keep a's sums, the others b's:
//     x = { a[l] where (l & D) == 0, b[l ^ D] elsewhere },  y = { a[l ^ D] where (l & D) == 0, b[l] elsewhere },  r = x + y
// which is one v_add_f64 for two trees (same operands as the plain tree, addition is commutative: same bits), and
// the register count halves level by level: 22 -> 11 -> 6 -> 3 -> 2.  x and y cost two instructions per 32-bit
// half: v_permlane32_swap / v_permlane16_swap for D = 32 / 16 (CDNA4), a DPP row rotation whose bank mask does the
// select for D = 8 / 4.  The last levels (D = 2, 1, and 32 for the uint16 lane order) run plain on the 2-3 registers left.
// ~90 instructions instead of ~450.  Which lane ends up with which sum is a compile-time function of the lane bits
// (flush_moments); the sums leave in one store instruction per remaining register.
__device__ __forceinline__ double f64_of(int hi, int lo) { return __hiloint2double(hi, lo); }

template <int D>
__device__ __forceinline__ double merge_pair(double a, double b) {
  const int alo = __double2loint(a), ahi = __double2hiint(a), blo = __double2loint(b), bhi = __double2hiint(b);
  if (D == 32) {        // new a = [a lanes 0..31 | b lanes 0..31], new b = [a lanes 32..63 | b lanes 32..63]
    const auto l = __builtin_amdgcn_permlane32_swap(alo, blo, false, false);
    const auto h = __builtin_amdgcn_permlane32_swap(ahi, bhi, false, false);
    return f64_of(h[0], l[0]) + f64_of(h[1], l[1]);
  } else if (D == 16) {  // new a = rows [a0 b0 a2 b2], new b = rows [a1 b1 a3 b3]
    const auto l = __builtin_amdgcn_permlane16_swap(alo, blo, false, false);
    const auto h = __builtin_amdgcn_permlane16_swap(ahi, bhi, false, false);
    return f64_of(h[0], l[0]) + f64_of(h[1], l[1]);
  } else if (D == 8) {   // row_ror:8; bank mask 0xC = lanes 8..15 of a row, 0x3 = lanes 0..7
    const int xlo = __builtin_amdgcn_update_dpp(alo, blo, 0x128, 0xf, 0xC, false), xhi = __builtin_amdgcn_update_dpp(ahi, bhi, 0x128, 0xf, 0xC, false);
    const int ylo = __builtin_amdgcn_update_dpp(blo, alo, 0x128, 0xf, 0x3, false), yhi = __builtin_amdgcn_update_dpp(bhi, ahi, 0x128, 0xf, 0x3, false);
    return f64_of(xhi, xlo) + f64_of(yhi, ylo);
  } else {               // D == 4: row_ror:4 (lane <- lane - 4) into banks 1, 3; row_ror:12 (lane <- lane + 4) into banks 0, 2
    static_assert(D == 32 || D == 16 || D == 8 || D == 4, "merge levels");
    const int xlo = __builtin_amdgcn_update_dpp(alo, blo, 0x124, 0xf, 0xA, false), xhi = __builtin_amdgcn_update_dpp(ahi, bhi, 0x124, 0xf, 0xA, false);
    const int ylo = __builtin_amdgcn_update_dpp(blo, alo, 0x12C, 0xf, 0x5, false), yhi = __builtin_amdgcn_update_dpp(bhi, ahi, 0x12C, 0xf, 0x5, false);
    return f64_of(xhi, xlo) + f64_of(yhi, ylo);
  }
}

// v[l] + v[l ^ D] in every lane (the levels that run on the few registers left)
template <int D>
__device__ __forceinline__ double plain_level(double v) {
  const int lo = __double2loint(v), hi = __double2hiint(v);
  if (D == 32) {
    const auto l = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    const auto h = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    return f64_of(h[0], l[0]) + f64_of(h[1], l[1]);
  }
  static_assert(D == 32 || D == 2 || D == 1, "plain levels");
  constexpr int ctrl = D == 2 ? 0x4E : 0xB1;   // quad_perm [2,3,0,1] / [1,0,3,2]
  return v + f64_of(__builtin_amdgcn_update_dpp(0, hi, ctrl, 0xf, 0xf, false), __builtin_amdgcn_update_dpp(0, lo, ctrl, 0xf, 0xf, false));
}

template <int D, int NIN, int NMAX>
__device__ __forceinline__ void merge_level(double (&v)[NMAX]) {
  constexpr int NOUT = (NIN + 1) / 2;
#pragma unroll
  for (int i = 0; i < NOUT; ++i) v[i] = merge_pair<D>(v[2 * i], v[2 * i + 1 < NIN ? 2 * i + 1 : NIN - 1]);   // an odd one out merges with itself
}

// Reduce the per-lane power sums of a finished work unit, write them to its partial slot and clear them.
// COHERENT: the stores are agent-scope (write-through, `sc1`): another workgroup - on another XCD, behind another L2 -
// may read the slot in the same launch (fused_fit) without an L2 write-back in between.
template <int M, int P, bool LANE_IS_PIXEL, bool COHERENT>
__device__ __forceinline__ void flush_moments(double (&acc_m)[2][M], const bool (&bval)[2], const int (&bidx)[2],
                                              double* part, int lane) {
  constexpr int N = 2 * M;
  constexpr bool kTop32 = P == 64 && LANE_IS_PIXEL;       // float32, 64-pixel groups: lane distance 32 comes first
  constexpr int N1 = kTop32 ? (N + 1) / 2 : N, N2 = (N1 + 1) / 2, N3 = (N2 + 1) / 2, N4 = (N3 + 1) / 2;
  double v[N];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int m = 0; m < M; ++m) {
      v[j * M + m] = acc_m[j][m];
      acc_m[j][m] = 0.0;
    }
  if (kTop32) merge_level<32, N, N>(v);
  merge_level<16, N1, N>(v);
  merge_level<8, N2, N>(v);
  merge_level<4, N3, N>(v);
#pragma unroll
  for (int q = 0; q < N4; ++q) {
    v[q] = plain_level<1>(plain_level<2>(v[q]));
    if (!LANE_IS_PIXEL) v[q] = plain_level<32>(v[q]);     // uint16 lane order: even / odd pixel halves last
  }
  // which sum sits in this lane of register q: undo the merges, last level first
  const int b2 = (lane >> 2) & 1, b3 = (lane >> 3) & 1, b4 = (lane >> 4) & 1, b5 = (lane >> 5) & 1;
  const bool canonical = (lane & 3) == 0 && (LANE_IS_PIXEL || b5 == 0);      // one writer among the lanes holding a copy
#pragma unroll
  for (int q = 0; q < N4; ++q) {
    int i = 2 * q + b2;
    bool valid = canonical && i < N3;
    i = 2 * i + b3;
    valid = valid && i < N2;
    i = 2 * i + b4;
    valid = valid && i < N1;
    if (kTop32) {
      i = 2 * i + b5;
      valid = valid && i < N;
    }
    const bool second = i >= M;                            // i = j * M + m
    if (valid && (second ? bval[1] : bval[0])) {
      double* dst = part + (size_t)(second ? bidx[1] : bidx[0]) * M + (i - (second ? M : 0));   // slot-major partials: [slot][band][moment]
      if (COHERENT)
        __hip_atomic_store(dst, v[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else
        *dst = v[q];
    }
  }
}

// Tail of a single-tile K1+K2 launch with a.fit set: the slot reduction and the solve without a second launch.
// The summation tree is the one of row_sum (hsr_poly.hip) - lane l of a wave adds slots l, l+64, ... in order, then
// the xor butterfly 32,16,..,1 over the 64 lanes - cut at the lane sums: the 64 "lanes" are 64 GROUPS of slots
// (group = slot mod 64).  The workgroup whose ticket completes a group adds that group's slots (level A, one thread per
// (band, moment) row, coalesced over rows) into gpart; the workgroup whose ticket completes the groups runs the
// butterfly over the 64 group sums (level B, one wave per row), writes the moments, solves the bands (same
// solve_band_t as every other path; its Jacobi matrices in LDS) and re-arms the tickets.  Same tree, same adds ->
// the bits of hsr_moments_reduce_solve on the same partials.
// Visibility between workgroups on different XCDs (each behind its own L2): every value that crosses is stored and
// loaded at agent scope (`sc1`: write-through / read-through), the writers wait for their stores (vmcnt(0)) before the
// barrier that precedes the ticket.  (A release/acquire fence pair instead - buffer_wbl2 / buffer_inv in each of the
// 512 workgroups, with the output image's dirty lines in L2 - cost 40 us per launch.)
__device__ __forceinline__ double ld_agent(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_agent(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void stores_done_barrier() {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
}

template <int DEG>
__device__ __forceinline__ void fused_fit(const SrfFit& f, const double* part, int S, int nb, unsigned char* lds, int t) {
  constexpr int M = moment_count(DEG);
  const int R = nb * M;                      // rows, <= 224
  const int grp = (int)blockIdx.x & 63;
  int* flag = reinterpret_cast<int*>(lds);
  double* mom = reinterpret_cast<double*>(lds + 64);
  double* work = mom + HSR_MAX_BANDS * M;
  stores_done_barrier();                     // every wave's partial stores are through; the LDS staging is dead
  if (t == 0) {
    const int expect = (S - grp + 63) >> 6;  // slots s < S with s mod 64 == grp
    *flag = __hip_atomic_fetch_add(f.tickets + grp, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == expect - 1;
  }
  __syncthreads();
  if (!*flag) return;
  if (t < R) {                               // level A: row t of the slots grp, grp + 64, ... (row_sum's lane loop)
    const double* row = part + t;
    double s = 0.0;
    for (int i0 = grp; i0 < S; i0 += 64 * 8) {
      double v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int i = i0 + 64 * u;
        v[u] = i < S ? ld_agent(row + (size_t)i * R) : 0.0;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) s += v[u];
    }
    st_agent(f.gpart + (size_t)grp * R + t, s);
  }
  stores_done_barrier();
  const int ngroups = S < 64 ? S : 64;
  if (t == 0) *flag = __hip_atomic_fetch_add(f.tickets + 64, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == ngroups - 1;
  __syncthreads();
  if (!*flag) return;
  {                                          // level B: the butterfly over the 64 group sums, one wave per row
    const int lane = t & 63, wave = t >> 6;
    constexpr int RW = (HSR_MAX_BANDS * M + 7) / 8;   // rows per wave; all loads first (read-through: ~1 us each)
    double v[RW];
#pragma unroll
    for (int k = 0; k < RW; ++k) {
      const int r = wave + 8 * k;
      v[k] = (r < R && lane < ngroups) ? ld_agent(f.gpart + (size_t)lane * R + r) : 0.0;
    }
#pragma unroll
    for (int k = 0; k < RW; ++k) {
      const int r = wave + 8 * k;
      const double s = wave_sum(v[k]);
      if (lane == 0 && r < R) {
        mom[r] = s;
        f.moments[r] = s;
      }
    }
  }
  __syncthreads();
  if (t < nb) solve_band_t<DEG, true>(mom + t * M, f.min_count, f.coeffs + (size_t)t * (DEG + 1), work + t * kSolveWork);
  if (t < 65) __hip_atomic_store(f.tickets + t, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // re-arm: every other workgroup has drawn its tickets
}

// The fit of the PREVIOUS tile as tail work of a K1 launch (APPLY variants, no exchange).  Its partial slots were written
// by the previous launch, so - unlike fused_fit above, which reduces the launch's OWN slots and pays memory-side coherence
// round trips for it - nothing has to cross between running workgroups: a workgroup that has finished its groups draws a
// ticket, and tickets 0 .. nb-1 reduce + solve one band each while the slower workgroups are still streaming.  Idle tail
// time instead of a launch of its own, a side stream, two events and CUs kept free for it.  Same tree as
// hsr_moments_reduce_solve ("lane" l adds slots l, l + 64, ... in batches of eight, butterfly over the 64 lane sums, the
// same solve), hence the same bits.
template <int DEG, int T>
__device__ __forceinline__ void lazy_fit(const SrfArgs& a, unsigned char* smem, int t) {
  constexpr int M = moment_count(DEG);
  static_assert(T == 512, "two passes of 32 lane rows");
  int* ticket = reinterpret_cast<int*>(smem);
  __syncthreads();                                        // everybody is done with the tile buffers
  if (t == 0) *ticket = (int)(__hip_atomic_fetch_add(a.lazy_counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - a.lazy_base);
  __syncthreads();
  const int b = *ticket;
  if (b < 0 || b >= a.nb) return;                         // workgroup-uniform
  double (*lsum)[16] = reinterpret_cast<double (*)[16]>(smem + 64);
  double* mom = reinterpret_cast<double*>(smem + 64 + 64 * 16 * 8);
  double* work = mom + 16;
  const int stride = a.nb * M, m = t & 15;
  const double* row = a.lazy_partials + (size_t)b * M + (m < M ? m : 0);
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    const int l = pass * 32 + (t >> 4);
    double s = 0.0;
    for (int i0 = l; i0 < a.lazy_slots; i0 += 64 * 8) {
      double v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int i = i0 + 64 * u;
        v[u] = (i < a.lazy_slots && m < M) ? row[(size_t)i * stride] : 0.0;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) s += v[u];
    }
    lsum[l][m] = s;
  }
  __syncthreads();
  if (t < M) {
    double acc[32];
#pragma unroll
    for (int l = 0; l < 32; ++l) acc[l] = lsum[l][t] + lsum[l + 32][t];
#pragma unroll
    for (int off = 16; off >= 1; off >>= 1)
#pragma unroll
      for (int l = 0; l < off; ++l) acc[l] = acc[l] + acc[l + off];
    mom[t] = acc[0];
    if (a.lazy_ready) st_agent(a.lazy_moments + (size_t)b * M + t, acc[0]);     // read by another queue while this launch still runs
    else a.lazy_moments[(size_t)b * M + t] = acc[0];
  }
  if (a.lazy_ready) {                                     // exchange pipelines: the all-reduce and the solve follow on the side stream
    stores_done_barrier();
    if (t == 0) __hip_atomic_fetch_add(a.lazy_ready, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return;
  }
  __syncthreads();
  if (a.lazy_group_T > 1) {                               // one fit over a group of tiles
    const int T2 = a.lazy_group_T;
    if (a.lazy_group_index != T2 - 1) return;             // not the group's last tile: its moments are in place, nothing to solve yet
    // hsr_moments_reduce over T "slots" (the tiles' moments): lane l's sum is 0.0 + entry l (one entry per lane for T <= 64, the
    // other lanes hold 0.0), then the butterfly over the 64 lane sums - the same adds, hence the bits of the other mosaic forms.
    // The last tile's own entry comes from LDS (this workgroup has just written it), the others from earlier launches.
    double* gm = mom + 16 + kSolveWork;
    if (t < M) {
      double acc[32];
#pragma unroll
      for (int l = 0; l < 32; ++l) {
        const double lo = l < T2 ? 0.0 + (l == T2 - 1 ? mom[t] : a.lazy_group_moments[((size_t)l * a.nb + b) * M + t]) : 0.0;
        const int h = l + 32;
        const double hi = h < T2 ? 0.0 + (h == T2 - 1 ? mom[t] : a.lazy_group_moments[((size_t)h * a.nb + b) * M + t]) : 0.0;
        acc[l] = lo + hi;
      }
#pragma unroll
      for (int off = 16; off >= 1; off >>= 1)
#pragma unroll
        for (int l = 0; l < off; ++l) acc[l] = acc[l] + acc[l + off];
      gm[t] = acc[0];
      a.lazy_group_total[(size_t)b * M + t] = acc[0];
    }
    __syncthreads();
    if (t == 0) solve_band_t<DEG, true>(gm, a.lazy_min_count, a.lazy_coeffs + (size_t)b * (DEG + 1), work);
    return;
  }
  if (t == 0) solve_band_t<DEG, true>(mom, a.lazy_min_count, a.lazy_coeffs + (size_t)b * (DEG + 1), work);
}

// K3 as a pre-phase of a K1 launch (APPLY variants; round 3).  In the pipelined order K3 of tile i-2 only needs coefficients
// that were ready a whole K1 ago, so it does not need a launch of its own: every workgroup applies its slice of the older
// tile before it starts its groups.  What that buys (profiles/r03_strong_scaling.md): a separate K3 is 20.6 us + a launch
// boundary on a 1024 x 1024 tile and a fixed ~8 us of latency on a 128-row block; as a pre-phase it costs its bytes
// (101 MB at the chip's rate = ~15 us; 2-3 us for the block).  Same arithmetic as apply_rows_kernel (float64 Horner
// without FMA contraction, mask select, clip, channels >= nb pass through), hence the same bits.  504 of the 512 threads
// take part: 504 is a multiple of every row length in float4 (1 .. 4), so a thread keeps its channel group and its
// coefficients stay in registers.
// Exchange pipelines: the coefficients were written by a kernel of ANOTHER queue (all-reduce -> solve on the side stream) with no
// event in between, so the workgroup first polls the tile's "coefficients ready" word - set a whole K1 ago in any sane
// schedule - and reads them through to LDS (agent-scope loads: this XCD's L2 may hold the slot's previous set).
template <int N, int T>
__device__ __forceinline__ void apply_prephase(const SrfArgs& a, unsigned char* smem, int t) {
  constexpr int kUse = T / 12 * 12;
  constexpr int U = 4;
  if (a.apply_x == nullptr) return;                            // workgroup-uniform
  const bool gated = a.apply_ready != nullptr;
  double* cl = reinterpret_cast<double*>(smem);                // [nb][N]: the tile buffers are not in use yet
  if (gated) {
    if (t == 0) wait_word_at_least(a.apply_ready, a.apply_ready_value, a.sync_error, 2u);
    __syncthreads();
    if (t < a.nb * N) cl[t] = ld_agent(a.apply_coeffs + t);
    __syncthreads();
  }
  const int q = (int)(a.out_ps >> 2);
  const uint32_t nv = (uint32_t)(a.apply_npix * q);            // host: apply_npix * q < 2^31
  const uint32_t stride = gridDim.x * (uint32_t)kUse;
  const uint32_t i0 = blockIdx.x * (uint32_t)kUse + (uint32_t)(t < kUse ? t : 0);
  const int c0 = (int)(i0 % (uint32_t)q) * 4;
  double c[4][N];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int ch = c0 + j < a.nb ? c0 + j : 0;
#pragma unroll
    for (int k = 0; k < N; ++k) c[j][k] = gated ? cl[ch * N + k] : a.apply_coeffs[ch * N + k];
  }
  if (gated) __syncthreads();                                  // everybody holds its coefficients: the first group's DMA may land on cl
  if (t >= kUse) return;
  const float4* x4 = reinterpret_cast<const float4*>(a.apply_x);
  float4* o4 = reinterpret_cast<float4*>(a.apply_out);
  for (uint32_t ib = i0; ib < nv; ib += stride * U) {
    float4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const uint32_t i = ib + u * stride;
      if (i < nv) v[u] = ld_stream(x4 + i);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const uint32_t i = ib + u * stride;
      if (i >= nv) break;
      const bool m = !a.apply_mask || a.apply_mask[i / (uint32_t)q];
      float r[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (c0 + j < a.nb) {
          float xv = r[j];
          if (m) {                      // np.polyval: y = 0; y = y*x + c, separately rounded
            const double xd = (double)xv;
            double y = 0.0;
#pragma unroll
            for (int k = 0; k < N; ++k) y = __dadd_rn(__dmul_rn(y, xd), c[j][k]);
            xv = (float)y;
          }
          r[j] = a.apply_clip ? (xv < 0.0f ? 0.0f : (xv > 1.0f ? 1.0f : xv)) : xv;
        }
      }
      st_stream(o4 + i, make_float4(r[0], r[1], r[2], r[3]));
    }
  }
}

template <int DEG, bool FAST, bool WLDS, int P, bool OUTV, bool BATCH, bool APPLY = false>
__global__ __launch_bounds__(8 * P, 4) void srf_kernel(const SrfArgs a) {
  constexpr int T = 8 * P;
  constexpr int NW = T / 64;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* tile = reinterpret_cast<float*>(smem);
  const int B = a.B;
  const int ldsB = a.ldsB;
  uint32_t* flags = reinterpret_cast<uint32_t*>(smem + (size_t)P * ldsB * 4);
  uint32_t* ustage = flags + 64;                                  // [16] record of the next unit (batch launches)
  const float* wl = reinterpret_cast<const float*>(flags + 64 + (BATCH ? 16 : 0));  // 16-byte aligned
  float* ostage = const_cast<float*>(wl) + (WLDS ? a.wtaps : 0);  // [P][out_ps] output slab (OUTV only)
  const int ops = (int)a.out_ps;
  const int osr = stage_row(a.nb, ops);                           // row stride of the staged slab (floats)
  static_assert(P == 64, "one pixel per lane: the target stage is indexed by lane");
  float* ystage = ostage + (OUTV ? P * osr : 0);                  // [nb][64] fit targets of the group (DEG > 0)
  uint32_t* mstage = reinterpret_cast<uint32_t*>(ystage + a.nb * 64);   // [64] mask bytes, one dword slot per lane
  float* prev_out = nullptr;
  int64_t prev_pix0 = 0;
  int prev_npx = 0;

#ifdef HSR_PHASE_STAMPS
  const unsigned long long rt_begin = real_time();
#endif
  const int t = threadIdx.x;
  const int lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int pl = t % P;     // pixel of this thread inside the group
  const int grp = wave;     // band group 0..7: wave-uniform, so the band parameters below live in SGPRs (-20 VGPRs)
  const int nchunk = P * B / 4;  // 16-byte chunks of a full group (P is a multiple of 4)

  // (The pre-phase stays AHEAD of the first DMA.  Behind it - so that the first group lands while the workgroup applies its
  // slice - the fused launch took 0.296 ms instead of 0.207: vmcnt counts in order, so every load of the pre-phase waits for
  // the whole 73 KB group that was issued before it, in every workgroup at once.)
  if constexpr (APPLY && DEG > 0) apply_prephase<DEG + 1, T>(a, smem, t);

  // the (at most two) bands of this thread, fixed for the whole launch
  int bk0[kBandSlots], bkl[kBandSlots], bwo[kBandSlots], bidx[kBandSlots];
  bool bval[kBandSlots];
#pragma unroll
  for (int j = 0; j < kBandSlots; ++j) {
    const int b = a.bands.band_of[grp][j];   // balanced assignment, not grp + 8*j
    bval[j] = b >= 0;
    bidx[j] = bval[j] ? b : 0;
    const int bb = bidx[j];
    bk0[j] = a.bands.k0[bb];
    bkl[j] = bval[j] ? a.bands.klen[bb] : 0;
    bwo[j] = a.bands.woff[bb];
  }

  // Once per workgroup: compact weight taps -> LDS, pad columns of the output slab zeroed.  Issued BEHIND the first group's
  // DMA (behind_the_dma below), not in front of it: anything ahead of the first DMA is dead time for the whole workgroup
  // (r03 per-round stamps: round 0 costs ~6 us more than a steady round - 3 % of a 1024 x 1024 launch, 20 % of the
  // 128-row block of an 8-way strong-scaling run).  Both are first read after the first group barrier.
  bool constants_staged = false;
  auto stage_constants = [&]() {
    if (WLDS) {
      float* wlw = const_cast<float*>(wl);
      for (int b = 0; b < a.nb; ++b) {
        const int kl = a.bands.klen[b];  // whole 16-tap chunks inside [0, B) (srf_common)
        for (int i = t; i < kl; i += T) wlw[a.bands.woff[b] + i] = a.wn[(size_t)b * B + a.bands.k0[b] + i];
      }
    }
  };

#ifdef HSR_CONSTANTS_FIRST      // diagnostic variant (tools/dbg/build_variants.sh): round 2's order, for A/B runs
  stage_constants();
  constants_staged = true;
#endif

#ifdef HSR_PHASE_STAMPS
  unsigned long long stamp_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long round_t0 = rt_begin;
  int round_no = 0;
#endif
  constexpr int M = DEG > 0 ? moment_count(DEG) : 1;
  double acc_m[kBandSlots][M];
  if (DEG > 0) {
#pragma unroll
    for (int j = 0; j < kBandSlots; ++j)
#pragma unroll
      for (int m = 0; m < M; ++m) acc_m[j][m] = 0.0;
  }

  // work units of this workgroup: cu = the one being processed; batch launches stage the record of the next one
  // (index nidx) in LDS, see unit_from_lds
  SrfUnit cu;
  if (BATCH) {
    cu = a.units[blockIdx.x];
  } else {
    cu = a.one;
    cu.slot = blockIdx.x;
    cu.part_dev = a.one.part_dev + (size_t)blockIdx.x * a.nb * M;
  }
  int64_t nidx = (int64_t)blockIdx.x + gridDim.x;
  int g = cu.slot;
  bool pend = false;             // a finished unit's sums still sit in acc_m (batch launches)
  double* pend_part = nullptr;

  for (bool more = true; more;) {
    const int64_t pix0 = (int64_t)g * P;
    const int64_t left = cu.npix - pix0;
    const int npx = left < P ? (int)left : P;
    const float* src = reinterpret_cast<const float*>(cu.cube_dev) + pix0 * B;
    const bool pvalid = pl < npx;

    // operands of the fused fit (2 target values + 1 mask byte per thread): staged in LDS by per-lane LDS-DMA right
    // after the group's DMA (stage_f32 / stage_u8 above), read after the group barrier
    float yv[kBandSlots];
    uint32_t mraw = 1u;
    auto load_targets = [&]() {
      if (DEG > 0) {
        const int64_t pc = pvalid ? pix0 + pl : cu.npix - 1;   // clamped: always a valid address, no branch
#pragma unroll
        for (int j = 0; j < kBandSlots; ++j)
          if (bval[j]) stage_f32(cu.real_dev + bidx[j] * a.real_bs + pc * a.real_ps, ystage + bidx[j] * 64);
        if (cu.mask_dev != nullptr && wave == NW - 1) stage_u8(cu.mask_dev + pc, mstage);   // same pixels in every wave
      }
    };
    auto wait_targets = [&]() {      // behind the group barrier: every wave's DMA has landed
      if (DEG > 0) {
        // hipcc's own ordering of ds_reads behind a pending LDS-DMA is not to be relied on (seen in the .s: a ds_read of the
        // stage with no vmcnt wait on the path that skips the slab flush); the barrier's vmcnt(0) covers it, this makes it local
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int j = 0; j < kBandSlots; ++j) yv[j] = bval[j] ? ystage[bidx[j] * 64 + lane] : 0.0f;
        mraw = cu.mask_dev != nullptr ? staged_u8(cu.mask_dev + (pvalid ? pix0 + pl : cu.npix - 1), mstage, lane) : 1u;
      }
    };
    // everything that fills the wait for the DMA: targets, the next unit record, the previous group's output slab,
    // the previous unit's sums
    auto behind_the_dma = [&]() {
      load_targets();
      if (BATCH && wave == 0 && lane < 16)
        __builtin_amdgcn_global_load_lds((gptr_t)unit_record_addr(a.units, nidx, a.nunits, lane), (lptr_t)ustage, 4, 0, 0);
      if (!constants_staged) {       // first group of the workgroup only
        stage_constants();
        constants_staged = true;
      }
      if (OUTV) flush_stage<8 * P>(ostage, prev_out, prev_pix0, prev_npx, a.nb, ops, t);   // no-op before the first group
      if (BATCH && DEG > 0 && pend) {
        flush_moments<M, P, true, !BATCH>(acc_m, bval, bidx, pend_part, lane);
        pend = false;
      }
    };

    HSR_STAMP(st0);
    if (t < P) flags[t] = 0u;

    const bool fast_group = FAST && npx == P;
    // Order matters (measured, and checked in the .s): the DMA is issued FIRST - anything ahead of it is
    // dead time (flushing the previous slab first cost +12 us on the kernel); then the small target loads
    // (inline asm, see above); then the flush of the previous slab, whose ds_reads see the pending LDS-DMA
    // and make hipcc wait vmcnt(0) - i.e. it runs once the group (and the targets) have landed, just ahead
    // of the barrier that waits for the same thing.
    if (fast_group) {
      const char* srcb = reinterpret_cast<const char*>(src);
      for (int c0 = wave * 64; c0 < nchunk; c0 += T) {  // c0 is wave-uniform
        const int c = c0 + lane;
        if (c < nchunk)
          __builtin_amdgcn_global_load_lds((gptr_t)(srcb + (size_t)c * 16), (lptr_t)(smem + (size_t)c0 * 16),
                                           16, 0, kGldsStream);
      }
    }
    behind_the_dma();   // one call site: with two, the batch kernels' copy is not inlined and acc_m lives in scratch
    HSR_STAMP(st1);
    __syncthreads();    // the group has landed (generic loader: flags zeroed before anybody sets one)
    wait_targets();
    HSR_STAMP(st2);
    if (fast_group) {
      // non-finite sweep: kScanBatchF32 independent ds_read_b128 in flight per thread (a serial
      // read->wait->test loop cost 3.7k cycles per group; batched it is LDS-bandwidth bound).
      // Indices past the group are clamped to its last chunk (a harmless re-read, no predication).
      const float4* t4 = reinterpret_cast<const float4*>(smem);
      for (int c0 = t; c0 < nchunk; c0 += T * kScanBatchF32) {
        float4 v[kScanBatchF32];
#pragma unroll
        for (int u = 0; u < kScanBatchF32; ++u) {
          const int c = c0 + u * T;
          v[u] = t4[c < nchunk ? c : nchunk - 1];
        }
        bool bad = false;
#pragma unroll
        for (int u = 0; u < kScanBatchF32; ++u) bad |= any_nonfinite4(v[u]);
        if (bad) {  // rare
#pragma unroll
          for (int u = 0; u < kScanBatchF32; ++u) {
            const int c = c0 + u * T;
            const int e = (c < nchunk ? c : nchunk - 1) * 4;
            if (!finite_f32(v[u].x)) flags[(e + 0) / B] = 1u;
            if (!finite_f32(v[u].y)) flags[(e + 1) / B] = 1u;
            if (!finite_f32(v[u].z)) flags[(e + 2) / B] = 1u;
            if (!finite_f32(v[u].w)) flags[(e + 3) / B] = 1u;
          }
        }
      }
#ifdef HSR_PHASE_STAMPS
      HSR_STAMP(st3);
      if (a.stamps) { stamp_acc[0] += st1 - st0; stamp_acc[1] += st2 - st1; stamp_acc[2] += st3 - st2; }
#endif
    } else {
      // generic loader: any 4-byte alignment, any B, ragged last group.  One pixel row per wave step.
      for (int pp = wave; pp < npx; pp += NW) {
        bool bad = false;
        for (int k = lane; k < B; k += 64) {
          const float v = ld_stream(src + (size_t)pp * B + k);
          bad |= !finite_f32(v);
          tile[pp * ldsB + k] = v;
        }
        if (bad) flags[pp] = 1u;
      }
    }
    __syncthreads();

    HSR_STAMP(st4);
    const bool slow = flags[pl] != 0u;
    const float* v = tile + pl * ldsB;
    float accv[kBandSlots];
#pragma unroll
    for (int j = 0; j < kBandSlots; ++j) {
      float acc = 0.0f;
      const float* vs = v + bk0[j];
      if (WLDS) {
        // [k0, k0+kl) was widened by srf_common to whole 16-tap chunks that stay inside this pixel's
        // row; the added taps carry weight 0 and (for an unflagged pixel) multiply finite samples,
        // so the sum is bit-identical to the exact-support sum.  One chunk = 4 broadcast
        // ds_read_b128 (weights) + 16 ds_read_b32 (samples, stride ldsB words: conflict-free)
        // issued together, then a 16-deep fma chain: no serial remainder loop.
        // (8 taps are read at a time although the supports are padded to 16: 16 + 16 operands in flight cost registers
        // and measured no faster - A/B on one box, 0.2165 vs 0.2162 ms; same taps, same order, same bits)
        constexpr int CH = kTapChunk / 2;
        const float4* w4 = reinterpret_cast<const float4*>(wl + bwo[j]);
        for (int i0 = 0; i0 < bkl[j]; i0 += CH) {
          float4 ww[CH / 4];
          float xv[CH];
#pragma unroll
          for (int u = 0; u < CH / 4; ++u) ww[u] = w4[(i0 >> 2) + u];
#pragma unroll
          for (int u = 0; u < CH; ++u) xv[u] = vs[i0 + u];
#pragma unroll
          for (int u = 0; u < CH / 4; ++u) {
            acc = fmaf(ww[u].x, xv[4 * u + 0], acc);
            acc = fmaf(ww[u].y, xv[4 * u + 1], acc);
            acc = fmaf(ww[u].z, xv[4 * u + 2], acc);
            acc = fmaf(ww[u].w, xv[4 * u + 3], acc);
          }
        }
      } else {
        const float* ws = a.wn + (size_t)bidx[j] * B + bk0[j];
        for (int i = 0; i < bkl[j]; ++i) acc = fmaf(ws[i], vs[i], acc);
      }
      accv[j] = acc;
    }
    if (slow) {  // rare: the dense product reproduces the reference's Inf/NaN classification per band
#pragma unroll
      for (int j = 0; j < kBandSlots; ++j) {
        if (bval[j]) {
          const float* w = a.wn + (size_t)bidx[j] * B;
          float acc = 0.0f;
          for (int k = 0; k < B; ++k) acc = fmaf(w[k], v[k], acc);
          accv[j] = acc;
        }
      }
    }
#pragma unroll
    for (int j = 0; j < kBandSlots; ++j) {
      if (bval[j]) {
        const float acc = accv[j];
        if (OUTV) ostage[pl * osr + bidx[j]] = acc;
        else if (pvalid) st_stream(cu.pseudo_dev + bidx[j] * a.out_bs + (pix0 + pl) * a.out_ps, acc);
        if (DEG > 0) {
          const float y = yv[j];
          const bool ok = pvalid && mraw != 0u && finite_f32(acc) && finite_f32(y) && acc > a.min_x && y > a.min_y;
          if (ok) {
            const double xd = (double)acc, yd = (double)y;
            acc_m[j][0] += 1.0;
            acc_m[j][2 * DEG + 1] += yd;
            double pw = 1.0;
#pragma unroll
            for (int k = 1; k <= 2 * DEG; ++k) {
              pw *= xd;
              acc_m[j][k] += pw;
              if (k <= DEG) acc_m[j][2 * DEG + 1 + k] += pw * yd;
            }
          }
        }
      }
    }
    prev_out = cu.pseudo_dev;
    prev_pix0 = pix0;
    prev_npx = npx;
    // next group of this unit, or the next unit of this workgroup
    g += cu.slots;
    if (g >= cu.ngroups) {
      pend = true;
      pend_part = cu.part_dev;
      if (BATCH && nidx < a.nunits) {
        cu = unit_from_lds(ustage);     // landed before the barrier that published this group
        g = cu.slot;
        nidx += gridDim.x;
      } else {
        more = false;
      }
    }
    HSR_STAMP(st5);
    // The tile and the flags are rewritten by the next iteration: an LDS-only hazard.  A plain
    // __syncthreads() here would also wait (vmcnt(0)) for the plane stores just issued.
    lds_barrier();
#ifdef HSR_PHASE_STAMPS
    {
      HSR_STAMP(st6);
      if (a.stamps) { stamp_acc[3] += st5 - st4; stamp_acc[4] += st6 - st5; stamp_acc[5] += st6 - st0; stamp_acc[6] += 1; }
      if (a.stamps3 && t == 0) {
        const unsigned long long now = real_time();
        if (round_no < 64) a.stamps3[(size_t)blockIdx.x * 64 + round_no] = (uint32_t)(now - round_t0);
        round_t0 = now;
        ++round_no;
      }
    }
#endif
  }
  if (OUTV) flush_stage<8 * P>(ostage, prev_out, prev_pix0, prev_npx, a.nb, ops, t);  // last group (after the barrier)
#ifdef HSR_PHASE_STAMPS
  if (a.stamps && lane == 0)
    for (int k = 0; k < 8; ++k) a.stamps[((size_t)blockIdx.x * NW + wave) * 8 + k] = stamp_acc[k];
#endif
  if (DEG > 0 && pend) flush_moments<M, P, true, !BATCH>(acc_m, bval, bidx, pend_part, lane);
  if constexpr (DEG > 0 && !BATCH)
    if (a.fit.tickets) fused_fit<DEG>(a.fit, a.one.part_dev, a.one.slots, a.nb, smem, t);
  if constexpr (APPLY && DEG > 0)
    if (a.lazy_partials) lazy_fit<DEG, T>(a, smem, t);
#ifdef HSR_PHASE_STAMPS
  if (a.stamps2 && t == 0) {
    uint32_t xcc, hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    unsigned long long* o = a.stamps2 + (size_t)blockIdx.x * 4;
    o[0] = rt_begin; o[1] = real_time(); o[2] = xcc; o[3] = hw;
  }
#endif
}

// ---------------------------------------------------------------------------------------------------
// K1 on uint16 tiles: the training tiles on disk are uint16 reflectance x 10000 with
// nodata 65535 (reference writer tiles_helpers/utils.py:362-374).  Same data flow as srf_kernel, but
// the LDS-DMA moves the 2-byte samples (half the HBM bytes) and the decode x = float(u) * scale
// happens on the way out of LDS, one separately rounded multiply per tap, so that the planes and the
// moments are bit-identical to hsr_tile_decode_u16 followed by the float32 kernel.  A pixel holding a
// nodata sample decodes to NaN there, and 0 * NaN poisons every band: flagged pixels are NaN in all
// bands.  64-pixel groups (P*B*2 = 128*B bytes: always whole 16-byte chunks), 8 waves.
// lane -> pixel: lanes 0..31 take the even pixels of the group, lanes 32..63 the odd ones.  A pixel row is
// 2*B bytes = B/2 dwords (142.5 for B = 285), so with lane = pixel neighbouring rows alternate between two
// bank phases and 10 of 16 lane pairs collide; inside each half-wave the rows now start B dwords apart
// (odd -> conflict-free, as in the float32 kernel).

// Power sums of one (x, y) sample into the per-lane accumulators of a band.
template <int DEG>
__device__ __forceinline__ void moments_accumulate(double (&acc)[DEG > 0 ? moment_count(DEG) : 1], float x, float y) {
  if (DEG == 0) return;
  const double xd = (double)x, yd = (double)y;
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

typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
// nodata sweep of one group held in LDS: a 16-bit lane of (v ^ nodata:nodata) is zero exactly where the sample is nodata
__device__ __forceinline__ void u16_nodata_sweep(const uint4* t4, int nchunk, int nsamples, int B, uint32_t nd2,
                                                 uint32_t nodata, uint32_t* fl, int t) {
  constexpr int T = 512;
  for (int c0 = t; c0 < nchunk; c0 += T * kScanBatch) {
    uint4 v[kScanBatch];
#pragma unroll
    for (int u = 0; u < kScanBatch; ++u) {
      const int c = c0 + u * T;
      v[u] = t4[c < nchunk ? c : nchunk - 1];
    }
    // a 16-bit half of (v ^ nodata:nodata) is zero exactly where the sample is nodata: the packed minimum over all
    // halves is zero iff the batch holds one (xor + v_pk_min_u16 per dword; the carry-trick test it replaces took four)
    u16x2 mn = {(unsigned short)0xffffu, (unsigned short)0xffffu};
#pragma unroll
    for (int u = 0; u < kScanBatch; ++u) {
      const uint32_t w[4] = {v[u].x ^ nd2, v[u].y ^ nd2, v[u].z ^ nd2, v[u].w ^ nd2};
#pragma unroll
      for (int q = 0; q < 4; ++q) mn = __builtin_elementwise_min(mn, __builtin_bit_cast(u16x2, w[q]));
    }
    const bool hit = mn.x == 0 || mn.y == 0;
    if (hit) {  // rare
#pragma unroll
      for (int u = 0; u < kScanBatch; ++u) {
        const int c = c0 + u * T;
        const int e = (c < nchunk ? c : nchunk - 1) * 8;
        const uint32_t w[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          if ((w[q] & 0xffffu) == (nodata & 0xffffu) && e + 2 * q < nsamples) fl[(e + 2 * q) / B] = 1u;
          if ((w[q] >> 16) == (nodata & 0xffffu) && e + 2 * q + 1 < nsamples) fl[(e + 2 * q + 1) / B] = 1u;
        }
      }
    }
  }
}

// One band of one pixel from a uint16 group in LDS.  16 taps = 9 aligned dwords (two samples each; one extra because
// odd sample offsets start in the middle of a dword) realigned per lane with v_alignbyte: 9 ds_read_b32 instead of
// 16 ds_read_u16.  The ninth dword may lie past the row (next pixel / the flag words): its upper half is never used.
__device__ __forceinline__ float u16_band_dot(const uint16_t* tile, int e0, const float4* w4, int klen, float scale) {
  float acc = 0.0f;
  const uint32_t sh = (uint32_t)(e0 & 1) * 2u;
  const uint32_t* d32 = reinterpret_cast<const uint32_t*>(tile) + (e0 >> 1);
  for (int i0 = 0; i0 < klen; i0 += kTapChunk) {
    float4 ww[kTapChunk / 4];
    uint32_t d[kTapChunk / 2 + 1];
#pragma unroll
    for (int u = 0; u < kTapChunk / 4; ++u) ww[u] = w4[(i0 >> 2) + u];
#pragma unroll
    for (int u = 0; u <= kTapChunk / 2; ++u) d[u] = d32[(i0 >> 1) + u];
#pragma unroll
    for (int u = 0; u < kTapChunk / 4; ++u) {
      const uint32_t ea = __builtin_amdgcn_alignbyte(d[2 * u + 1], d[2 * u], sh);
      const uint32_t eb = __builtin_amdgcn_alignbyte(d[2 * u + 2], d[2 * u + 1], sh);
      acc = fmaf(ww[u].x, (float)(ea & 0xffffu) * scale, acc);
      acc = fmaf(ww[u].y, (float)(ea >> 16) * scale, acc);
      acc = fmaf(ww[u].z, (float)(eb & 0xffffu) * scale, acc);
      acc = fmaf(ww[u].w, (float)(eb >> 16) * scale, acc);
    }
  }
  return acc;
}

// Opt-in fast arithmetic for uint16 cubes (hsr_srf_options.flags & HSR_SRF_U16_FAST).  The exact path above spends
// ~3.5 VALU instructions per tap (realign, extract, convert, decode multiply, fma) and is VALU-bound at half the HBM
// bytes of the float32 kernel.  Here the decode scale is folded into the weights once (w' = w * scale, rounded to
// float32 when the taps are staged) and even / odd taps accumulate in the two halves of one v_pk_fma_f32: ~2 per tap.
// Not bit-identical to decode -> float32 K1 any more (one rounding moved, two partial sums): |rel| <= 1e-6 against
// the exact kernel (7e-7 observed), <= 2e-6 against the float64 oracle (tests/test_gpu_parity.py).  Measured: K1 on a
// 1024 x 1024 tile 0.136 vs 0.144 ms - the exact kernel is no longer VALU-bound since the ring, so halving the
// arithmetic buys 6 %, not 2x.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float u16_band_dot_fast(const uint16_t* tile, int e0, const float4* w4, int klen) {
  f32x2 acc = {0.0f, 0.0f};
  const uint32_t sh = (uint32_t)(e0 & 1) * 2u;
  const uint32_t* d32 = reinterpret_cast<const uint32_t*>(tile) + (e0 >> 1);
  for (int i0 = 0; i0 < klen; i0 += kTapChunk) {
    float4 ww[kTapChunk / 4];
    uint32_t d[kTapChunk / 2 + 1];
#pragma unroll
    for (int u = 0; u < kTapChunk / 4; ++u) ww[u] = w4[(i0 >> 2) + u];
#pragma unroll
    for (int u = 0; u <= kTapChunk / 2; ++u) d[u] = d32[(i0 >> 1) + u];
#pragma unroll
    for (int u = 0; u < kTapChunk / 4; ++u) {
      const uint32_t ea = __builtin_amdgcn_alignbyte(d[2 * u + 1], d[2 * u], sh);
      const uint32_t eb = __builtin_amdgcn_alignbyte(d[2 * u + 2], d[2 * u + 1], sh);
      const f32x2 x0 = {(float)(ea & 0xffffu), (float)(ea >> 16)}, x1 = {(float)(eb & 0xffffu), (float)(eb >> 16)};
      const f32x2 w0 = {ww[u].x, ww[u].y}, w1 = {ww[u].z, ww[u].w};
      acc = __builtin_elementwise_fma(w0, x0, acc);
      acc = __builtin_elementwise_fma(w1, x1, acc);
    }
  }
  return acc.x + acc.y;
}

template <int DEG, bool FAST, bool OUTV, bool BATCH>
__global__ __launch_bounds__(512, 4) void srf_u16_kernel(const SrfArgs a) {
  constexpr int P = 64, T = 512;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint16_t* tile = reinterpret_cast<uint16_t*>(smem);
  const int B = a.B;
  uint32_t* flags = reinterpret_cast<uint32_t*>(smem + (size_t)P * B * 2);
  uint32_t* ustage = flags + 64;                                  // [16] record of the next unit (batch launches)
  const float* wl = reinterpret_cast<const float*>(flags + 64 + (BATCH ? 16 : 0));
  const bool wlds = a.wtaps > 0;
  float* ostage = const_cast<float*>(wl) + a.wtaps;
  const int ops = (int)a.out_ps;
  const int osr = stage_row(a.nb, ops);
  float* ystage = ostage + (OUTV ? P * osr : 0);                  // [nb][64] fit targets of the group (DEG > 0)
  uint32_t* mstage = reinterpret_cast<uint32_t*>(ystage + a.nb * 64);
  float* prev_out = nullptr;
  int64_t prev_pix0 = 0;
  int prev_npx = 0;

  const int t = threadIdx.x;
  const int lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int pl = 2 * (lane & 31) + (lane >> 5), grp = wave;
  const float scale = a.scale;
  const uint32_t nd2 = (a.nodata & 0xffffu) * 0x00010001u;
  const bool has_nodata = a.nodata <= 0xffffu;

  int bk0[kBandSlots], bkl[kBandSlots], bwo[kBandSlots], bidx[kBandSlots];
  bool bval[kBandSlots];
#pragma unroll
  for (int j = 0; j < kBandSlots; ++j) {
    const int b = a.bands.band_of[grp][j];
    bval[j] = b >= 0;
    bidx[j] = bval[j] ? b : 0;
    bk0[j] = a.bands.k0[bidx[j]];
    bkl[j] = bval[j] ? a.bands.klen[bidx[j]] : 0;
    bwo[j] = a.bands.woff[bidx[j]];
  }
  if (wlds) {
    float* wlw = const_cast<float*>(wl);
    for (int b = 0; b < a.nb; ++b) {
      const int kl = a.bands.klen[b];
      for (int i = t; i < kl; i += T) wlw[a.bands.woff[b] + i] = a.wn[(size_t)b * B + a.bands.k0[b] + i];
    }
  }

  constexpr int M = DEG > 0 ? moment_count(DEG) : 1;
  double acc_m[kBandSlots][M];
  if (DEG > 0) {
#pragma unroll
    for (int j = 0; j < kBandSlots; ++j)
#pragma unroll
      for (int m = 0; m < M; ++m) acc_m[j][m] = 0.0;
  }

  SrfUnit cu;   // see srf_kernel
  if (BATCH) {
    cu = a.units[blockIdx.x];
  } else {
    cu = a.one;
    cu.slot = blockIdx.x;
    cu.part_dev = a.one.part_dev + (size_t)blockIdx.x * a.nb * M;
  }
  int64_t nidx = (int64_t)blockIdx.x + gridDim.x;
  int g = cu.slot;
  bool pend = false;
  double* pend_part = nullptr;

  for (bool more = true; more;) {
    const int64_t pix0 = (int64_t)g * P;
    const int64_t left = cu.npix - pix0;
    const int npx = left < P ? (int)left : P;
    const uint16_t* src = reinterpret_cast<const uint16_t*>(cu.cube_dev) + pix0 * B;
    const bool pvalid = pl < npx;
    const int nchunk = (npx * B + 7) >> 3;  // 16-byte chunks (8 samples) holding the group

    // targets of the fused fit: per-lane LDS-DMA right after the group's DMA (see srf_kernel)
    float yv[kBandSlots];
    uint32_t mraw = 1u;
    auto load_targets = [&]() {
      if (DEG > 0) {
        const int64_t pc = pvalid ? pix0 + pl : cu.npix - 1;
#pragma unroll
        for (int j = 0; j < kBandSlots; ++j)
          if (bval[j]) stage_f32(cu.real_dev + bidx[j] * a.real_bs + pc * a.real_ps, ystage + bidx[j] * 64);
        if (cu.mask_dev != nullptr && wave == 7) stage_u8(cu.mask_dev + pc, mstage);   // every wave maps lanes to pixels alike
      }
    };
    auto wait_targets = [&]() {      // call behind the group barrier only (see srf_kernel)
      if (DEG > 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int j = 0; j < kBandSlots; ++j) yv[j] = bval[j] ? ystage[bidx[j] * 64 + lane] : 0.0f;
        mraw = cu.mask_dev != nullptr ? staged_u8(cu.mask_dev + (pvalid ? pix0 + pl : cu.npix - 1), mstage, lane) : 1u;
      }
    };
    auto behind_the_dma = [&]() {
      load_targets();
      if (BATCH && wave == 0 && lane < 16)
        __builtin_amdgcn_global_load_lds((gptr_t)unit_record_addr(a.units, nidx, a.nunits, lane), (lptr_t)ustage, 4, 0, 0);
      if (OUTV) flush_stage<T>(ostage, prev_out, prev_pix0, prev_npx, a.nb, ops, t);
      if (BATCH && DEG > 0 && pend) {
        flush_moments<M, P, false, !BATCH>(acc_m, bval, bidx, pend_part, lane);
        pend = false;
      }
    };

    if (t < P) flags[t] = 0u;
    const bool fast_group = FAST && npx == P;
    if (fast_group) {
      const char* srcb = reinterpret_cast<const char*>(src);
      for (int c0 = wave * 64; c0 < nchunk; c0 += T) {
        const int c = c0 + lane;
        if (c < nchunk)
          __builtin_amdgcn_global_load_lds((gptr_t)(srcb + (size_t)c * 16), (lptr_t)(smem + (size_t)c0 * 16),
                                           16, 0, kGldsStream);
      }
    }
    behind_the_dma();   // one call site (see srf_kernel)
    if (!fast_group) {
      // generic loader: any 2-byte alignment, ragged last group; the tail of the last chunk is zeroed
      const int n = npx * B;
      for (int i = t; i < nchunk * 8; i += T) tile[i] = i < n ? ld_stream(src + i) : (uint16_t)0;
    }
    __syncthreads();     // vmcnt(0) + barrier: the group and every wave's staged targets have landed
    wait_targets();
    if (has_nodata) {
      u16_nodata_sweep(reinterpret_cast<const uint4*>(smem), nchunk, npx * B, B, nd2, a.nodata, flags, t);
      __syncthreads();
    }

    const bool bad = flags[pl] != 0u;
#pragma unroll
    for (int j = 0; j < kBandSlots; ++j) {
      float acc = 0.0f;
      if (wlds) {
        acc = u16_band_dot(tile, pl * B + bk0[j], reinterpret_cast<const float4*>(wl + bwo[j]), bkl[j], scale);
      } else {
        const uint16_t* vs = tile + pl * B + bk0[j];
        const float* ws = a.wn + (size_t)bidx[j] * B + bk0[j];
        for (int i = 0; i < bkl[j]; ++i) acc = fmaf(ws[i], (float)vs[i] * scale, acc);
      }
      if (bad) acc = __uint_as_float(0x7fc00000u);
      if (bval[j]) {
        if (OUTV) ostage[pl * osr + bidx[j]] = acc;
        else if (pvalid) st_stream(cu.pseudo_dev + bidx[j] * a.out_bs + (pix0 + pl) * a.out_ps, acc);
        if (DEG > 0) {
          const float y = yv[j];
          const bool ok = pvalid && mraw != 0u && finite_f32(acc) && finite_f32(y) && acc > a.min_x && y > a.min_y;
          if (ok) moments_accumulate<DEG>(acc_m[j], acc, y);
        }
      }
    }
    prev_out = cu.pseudo_dev;
    prev_pix0 = pix0;
    prev_npx = npx;
    g += cu.slots;
    if (g >= cu.ngroups) {
      pend = true;
      pend_part = cu.part_dev;
      if (BATCH && nidx < a.nunits) {
        cu = unit_from_lds(ustage);
        g = cu.slot;
        nidx += gridDim.x;
      } else {
        more = false;
      }
    }
    lds_barrier();
  }
  if (OUTV) flush_stage<T>(ostage, prev_out, prev_pix0, prev_npx, a.nb, ops, t);
  if (DEG > 0 && pend) flush_moments<M, P, false, !BATCH>(acc_m, bval, bidx, pend_part, lane);
  if constexpr (DEG > 0 && !BATCH)
    if (a.fit.tickets) fused_fit<DEG>(a.fit, a.one.part_dev, a.one.slots, a.nb, smem, t);
}

// LDS-DMA issued from inline asm: the compiler's waitcnt pass then does not know a DMA is pending and does
// not force vmcnt(0) in front of every ds_read; the wait is the explicit vmcnt(0) at the top of each iteration.
__device__ __forceinline__ void glds16_nt_asm(const void* gaddr, uint32_t lds_base) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off nt" ::"v"(gaddr), "s"(lds_base) : "memory");
}

__device__ __forceinline__ void glds4_asm(const void* gaddr, uint32_t lds_base) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %0, off" ::"v"(gaddr), "s"(lds_base) : "memory");
}

// Double-buffered form of srf_u16_kernel for 16-byte aligned cubes (the normal case).  A uint16 group is
// half the bytes of a float32 one, so with one group per workgroup only ~73 KB per CU were in flight and the
// kernel was latency-bound (0.169 ms).  Here every workgroup owns two group buffers (2 x 36.5 KB, the LDS
// footprint of the float32 kernel): the DMA of group k+1 and its fit targets are issued right after the
// barrier that publishes group k and land while group k is swept and reduced.  Two barriers per group; the
// closing barrier of the single-buffer kernel is not needed because nothing of group k is overwritten before
// the next top barrier.  Same arithmetic, same summation trees -> same bits as srf_u16_kernel.
// Batch launches: the group after the last one of a unit is the first group of the workgroup's next unit, so three
// unit records are alive: cu (being reduced), nu (being prefetched), nn (its record is on its way).
// Where an iteration goes (r02, -DHSR_PHASE_STAMPS, tools/k1_stamps 1024 1024 64 1, deg 3): wait for the own DMA 3 %,
// top barrier 13 %, issue of the next group's DMA + flush 32 %, sweep + barrier 22 %, dot products + moments 31 %
// (the wave with the 48-tap band 2850 cycles, the others ~1770).  The DMA has landed when the waves come back for it; the
// time goes into the ISSUE of 4-5 global_load_lds per wave, which blocks at HBM pace because the CU's request queue is
// full (hsr_probe_read mode 2: the same 36 KiB groups with nothing else read 6.9-7.0 TB/s), and a blocked wave computes
// nothing.  Measured and dropped: the issue spread in five pieces over the iteration (the blocked time moves with it,
// 0.162 vs 0.159 ms on one box); one producer wave issuing everything and flushing the rows, 7 consumer waves with a
// consumer-only LDS counter instead of the second barrier (a single wave issues 1 KB per ~155 cycles: 0.178 ms; two
// producer waves: 0.174 ms - the consumers, whose target loads wait in the same full queue and whose widest band sets
// the pace, are the critical path then); deferring flags and moments by one iteration to drop the second barrier
// (43-58 spilled VGPRs in the loop, and scratch waits in the same queue: 0.258 ms); the fit targets as one LDS-DMA of the
// group's 64 target rows (3 instructions, 24 cache lines) instead of 16 scattered wave loads (512 lines) - within 1 % of
// this kernel in three alternating fresh processes each; each wave issuing its share of the DMA at a different point of
// the iteration (wave mod 5) - 3 % slower.  (All in the settled power state: 0.124-0.132 ms exact, 0.120-0.132 ms fast.)
// Round 4: WAVE PRIORITIES.  With half the bytes per group the two workgroups of a CU no longer hide each other's compute behind
// the DMA: an iteration is a chain of issue -> sweep -> dot products, and the SIMD's arbiter (oldest wave first) lets a wave of
// the other workgroup that is merely sweeping or spinning hold up the wave that is on the critical path.  s_setprio per phase -
// 3 while a wave issues the next group's DMA and flushes the previous rows (HBM is fed first), 1 during the dot products and
// moments, 0 for the sweep and the waits - measured on one box against the library without it (bench.py --cube u16, 100 steps,
// fresh processes, two rounds): K1+K2 alone 0.1250 / 0.1272 -> 0.1137 / 0.1152 ms, the fused launch 0.1422 / 0.1419 ->
// 0.1301 / 0.1332 ms (6.0 - 6.2 TB/s of everything the launch moves), fast arithmetic 0.1441 / 0.1352 -> 0.1363 / 0.1266.
// Any non-zero level for the dot products gives most of it (1, 2, 3, or graded by the wave's tap count: within the +-2 % of a
// fresh process); the same levels in the float32 kernel, which waits for HBM and nothing else, change nothing (0.2098 vs
// 0.2100 ms).  Also tried on the way and dropped: ONE barrier per group (every thread sweeps the chunks its own DMA wrote in
// front of the top barrier, three flag buffers, two staged slabs) - 5 % slower than two barriers, with and without priorities:
// the second barrier keeps the eight waves' DMA issue together; band -> wave assignment by SIMD instead of by wave - no effect.
constexpr int kU16PrioIssue = 3, kU16PrioDots = 1;
template <int DEG, bool OUTV, bool BATCH, bool FASTU, bool APPLY = false>
__global__ __launch_bounds__(512, 4) void srf_u16_ring_kernel(const SrfArgs a) {
  constexpr int P = 64, T = 512;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int B = a.B;
  const int tile_bytes = P * B * 2;   // 128*B: whole 16-byte chunks
  uint32_t* flags = reinterpret_cast<uint32_t*>(smem + 2 * (size_t)tile_bytes);   // [2][64]
  uint32_t* ustage = flags + 128;                                 // [2][16] records of the next unit (batch launches)
  const float* wl = reinterpret_cast<const float*>(flags + 128 + (BATCH ? 32 : 0));
  const bool wlds = a.wtaps > 0;
  float* ostage = const_cast<float*>(wl) + a.wtaps;
  const int ops = (int)a.out_ps;
  const int osr = stage_row(a.nb, ops);

  const int t = threadIdx.x;
  const int lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int pl = 2 * (lane & 31) + (lane >> 5), grp = wave;   // see srf_u16_kernel
  const float scale = a.scale;
  const uint32_t nd2 = (a.nodata & 0xffffu) * 0x00010001u;
  const bool has_nodata = a.nodata <= 0xffffu;
  const int nchunk_full = tile_bytes >> 4;

  if constexpr (APPLY && DEG > 0) apply_prephase<DEG + 1, T>(a, smem, t);   // K3 of an older tile (fused pipeline, as in srf_kernel)

  int bk0[kBandSlots], bkl[kBandSlots], bwo[kBandSlots], bidx[kBandSlots];
  bool bval[kBandSlots];
#pragma unroll
  for (int j = 0; j < kBandSlots; ++j) {
    const int b = a.bands.band_of[grp][j];
    bval[j] = b >= 0;
    bidx[j] = bval[j] ? b : 0;
    bk0[j] = a.bands.k0[bidx[j]];
    bkl[j] = bval[j] ? a.bands.klen[bidx[j]] : 0;
    bwo[j] = a.bands.woff[bidx[j]];
  }
  auto stage_constants = [&]() {      // called right BEHIND the first group's prefetch (see srf_kernel)
    if (wlds) {
      float* wlw = const_cast<float*>(wl);
      for (int b = 0; b < a.nb; ++b) {
        const int kl = a.bands.klen[b];
        for (int i = t; i < kl; i += T) {
          const float wv = a.wn[(size_t)b * B + a.bands.k0[b] + i];
          wlw[a.bands.woff[b] + i] = FASTU ? wv * scale : wv;     // fast arithmetic: decode scale folded into the taps
        }
      }
    }
  };
  constexpr int M = DEG > 0 ? moment_count(DEG) : 1;
  double acc_m[kBandSlots][M];
  if (DEG > 0) {
#pragma unroll
    for (int j = 0; j < kBandSlots; ++j)
#pragma unroll
      for (int m = 0; m < M; ++m) acc_m[j][m] = 0.0;
  }
  const uint32_t lds0 = (uint32_t)(uintptr_t)(lptr_t)(smem);

  // prefetch of one group: LDS-DMA of the samples (full groups only; a ragged last group is filled by hand when
  // it is consumed) + the fit targets of this thread's pixel, all invisible to the compiler's waitcnt pass
  float yn[kBandSlots] = {0.0f, 0.0f};
  uint32_t mn = 1u;
  auto prefetch = [&](const SrfUnit& u, int gg, int buf) {
    const int64_t pix0 = (int64_t)gg * P;
    const int64_t left = u.npix - pix0;
    if (left >= P) {
      const char* srcb = reinterpret_cast<const char*>(reinterpret_cast<const uint16_t*>(u.cube_dev) + pix0 * B);
      for (int c0 = wave * 64; c0 < nchunk_full; c0 += T) {
        const int c = c0 + lane;
        if (c < nchunk_full)
          glds16_nt_asm(srcb + (size_t)c * 16, __builtin_amdgcn_readfirstlane(lds0 + (uint32_t)buf * tile_bytes + (uint32_t)c0 * 16));
      }
    }
    if (DEG > 0) {
      // The targets are ORDINARY loads here: yn / mn live across the loop back-edge, and an inline-asm load into a
      // loop-carried register can be copied by the compiler (phi of the prologue's and the loop's prefetch) before the
      // data has landed - seen as moments that changed from launch to launch in the batch kernel.  Ordinary loads
      // cost nothing in this kernel: its LDS-DMA is inline asm, so the compiler has no DMA to drain for (the
      // float32 kernel, whose DMA is the builtin, must keep asm loads, but consumes them in the same iteration).
      const int64_t pc = pl < left ? pix0 + pl : u.npix - 1;
#pragma unroll
      for (int j = 0; j < kBandSlots; ++j) yn[j] = u.real_dev[bidx[j] * a.real_bs + pc * a.real_ps];
      mn = u.mask_dev != nullptr ? (uint32_t)u.mask_dev[pc] : 1u;
    }
  };

  SrfUnit cu;
  if (BATCH) {
    cu = a.units[blockIdx.x];
  } else {
    cu = a.one;
    cu.slot = blockIdx.x;
    cu.part_dev = a.one.part_dev + (size_t)blockIdx.x * a.nb * M;
  }
  int g = cu.slot;
  prefetch(cu, g, 0);
  stage_constants();
  int64_t nidx = (int64_t)blockIdx.x + gridDim.x;   // index of the unit after cu
  // the record of unit nidx is in ustage[cur] at the top barrier of every iteration: an LDS-DMA like the samples,
  // issued one iteration earlier into the other half (srf_kernel explains why not through registers)
  auto fetch_record = [&](int64_t idx, int buf) {
    if (BATCH && wave == 0 && lane < 16)
      glds4_asm(unit_record_addr(a.units, idx, a.nunits, lane),
                __builtin_amdgcn_readfirstlane(lds0 + (uint32_t)((const unsigned char*)(ustage + 16 * buf) - smem)));
  };
  fetch_record(nidx, 0);
  int cur = 0;
  float* prev_out = nullptr;
  int64_t prev_pix0 = 0;
  int prev_npx = 0;
  bool pend = false;
  double* pend_part = nullptr;

#ifdef HSR_PHASE_STAMPS
  unsigned long long stamp_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
  for (bool more = true; more; cur ^= 1) {
    HSR_STAMP(st0);
    const int64_t pix0 = (int64_t)g * P;
    const int64_t left = cu.npix - pix0;
    const int npx = left < P ? (int)left : P;
    const bool pvalid = pl < npx;
    const int nchunk = (npx * B + 7) >> 3;
    uint16_t* tile = reinterpret_cast<uint16_t*>(smem + (size_t)cur * tile_bytes);
    uint32_t* fl = flags + 64 * cur;

    if (t < P) fl[t] = 0u;
    // everything this wave issued one iteration ago has landed: group k, its targets, the flush of group k-2
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    HSR_STAMP(st1);
    float yv[kBandSlots];
    static_assert(kBandSlots == 2, "two target registers");
    yv[0] = yn[0];
    yv[1] = yn[1];
    const uint32_t mraw = mn;
    if (npx < P) {  // ragged last group: plain copy, tail of the last chunk zeroed
      const uint16_t* src = reinterpret_cast<const uint16_t*>(cu.cube_dev) + pix0 * B;
      const int n = npx * B;
      for (int i = t; i < nchunk * 8; i += T) tile[i] = i < n ? src[i] : (uint16_t)0;
    }
    __syncthreads();   // group k (every wave's share of the DMA) and the staged planes of group k-1 are visible
    HSR_STAMP(st2);

    __builtin_amdgcn_s_setprio(kU16PrioIssue);
    // group k+1: the next group of this unit, or the first group of the workgroup's next unit.
    // Buffer cur^1 was last read before the barrier above.
    const int g2 = g + cu.slots;
    const bool same_unit = g2 < cu.ngroups;
    const bool next_unit = BATCH && !same_unit && nidx < a.nunits;
    {
      // ONE call site for the prefetch: its inline-asm loads write loop-carried registers (yn, mn), and two sites under
      // different conditions would be merged by register copies the compiler may place before the data has landed
      SrfUnit nx = cu;
      int ng = g2;
      if (next_unit) {
        nx = unit_from_lds(ustage + 16 * cur);
        ng = nx.slot;
      }
      if (same_unit || next_unit) prefetch(nx, ng, cur ^ 1);
    }
    fetch_record(next_unit ? nidx + gridDim.x : nidx, cur ^ 1);
    if (OUTV) flush_stage<T>(ostage, prev_out, prev_pix0, prev_npx, a.nb, ops, t);
    if (BATCH && DEG > 0 && pend) {
      flush_moments<M, P, false, !BATCH>(acc_m, bval, bidx, pend_part, lane);
      pend = false;
    }

    HSR_STAMP(st3);
    __builtin_amdgcn_s_setprio(0);
    if (has_nodata) u16_nodata_sweep(reinterpret_cast<const uint4*>(tile), nchunk, npx * B, B, nd2, a.nodata, fl, t);
    // flags of group k complete; also orders the flush reads of the staged slab before the writes below
    lds_barrier();
    HSR_STAMP(st4);

    const bool bad = fl[pl] != 0u;
    __builtin_amdgcn_s_setprio(kU16PrioDots);
#pragma unroll
    for (int j = 0; j < kBandSlots; ++j) {
      float acc = 0.0f;
      if (wlds) {
        acc = FASTU ? u16_band_dot_fast(tile, pl * B + bk0[j], reinterpret_cast<const float4*>(wl + bwo[j]), bkl[j])
                    : u16_band_dot(tile, pl * B + bk0[j], reinterpret_cast<const float4*>(wl + bwo[j]), bkl[j], scale);
      } else {
        const uint16_t* vs = tile + pl * B + bk0[j];
        const float* ws = a.wn + (size_t)bidx[j] * B + bk0[j];
        for (int i = 0; i < bkl[j]; ++i) acc = fmaf(ws[i], (float)vs[i] * scale, acc);
      }
      if (bad) acc = __uint_as_float(0x7fc00000u);
      if (bval[j]) {
        if (OUTV) ostage[pl * osr + bidx[j]] = acc;
        else if (pvalid) st_stream(cu.pseudo_dev + bidx[j] * a.out_bs + (pix0 + pl) * a.out_ps, acc);
        if (DEG > 0) {
          const float y = yv[j];
          const bool ok = pvalid && mraw != 0u && finite_f32(acc) && finite_f32(y) && acc > a.min_x && y > a.min_y;
          if (ok) moments_accumulate<DEG>(acc_m[j], acc, y);
        }
      }
    }
    __builtin_amdgcn_s_setprio(0);
#ifdef HSR_PHASE_STAMPS
    {
      HSR_STAMP(st5);
      // 0: own DMA share + stores landed, 1: barrier (slowest wave's share), 2: issue DMA k+1 + flush k-1 (+ batch flush),
      // 3: sweep + barrier, 4: dots + moments, 5: whole iteration
      if (a.stamps) { stamp_acc[0] += st1 - st0; stamp_acc[1] += st2 - st1; stamp_acc[2] += st3 - st2; stamp_acc[3] += st4 - st3;
                      stamp_acc[4] += st5 - st4; stamp_acc[5] += st5 - st0; stamp_acc[6] += 1; }
    }
#endif
    prev_out = cu.pseudo_dev;
    prev_pix0 = pix0;
    prev_npx = npx;
    if (same_unit) {
      g = g2;
    } else {
      pend = true;
      pend_part = cu.part_dev;
      if (next_unit) {
        cu = unit_from_lds(ustage + 16 * cur);   // decoded again rather than held in 15 SGPRs across the dot products
        g = cu.slot;
        nidx += gridDim.x;
      } else {
        more = false;
      }
    }
  }
#ifdef HSR_PHASE_STAMPS
  if (a.stamps && lane == 0)
    for (int k = 0; k < 8; ++k) a.stamps[((size_t)blockIdx.x * 8 + wave) * 8 + k] = stamp_acc[k];
#endif
  if (OUTV) {
    __syncthreads();
    flush_stage<T>(ostage, prev_out, prev_pix0, prev_npx, a.nb, ops, t);
  }
  if (DEG > 0 && pend) flush_moments<M, P, false, !BATCH>(acc_m, bval, bidx, pend_part, lane);
  if constexpr (DEG > 0 && !BATCH)
    if (a.fit.tickets) fused_fit<DEG>(a.fit, a.one.part_dev, a.one.slots, a.nb, smem, t);
  if constexpr (APPLY && DEG > 0)
    if (a.lazy_partials) lazy_fit<DEG, T>(a, smem, t);   // the previous tile's fit as tail work (fused pipeline)
}

template <typename K>
static void ensure_dynamic_lds(K kern, size_t lds, size_t* configured) {
  if (lds > *configured) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipGetLastError();
    *configured = lds;
  }
}

template <int DEG, bool OUTV, bool BATCH, bool FASTU, bool APPLY = false>
static int launch_srf_u16_ring(const SrfArgs& a, int grid, hipStream_t stream) {
  const size_t lds = (size_t)2 * 64 * a.B * 2 + 128 * sizeof(uint32_t) + (size_t)a.wtaps * 4 + (OUTV ? (size_t)64 * stage_row(a.nb, (int)a.out_ps) * 4 : 0);
  auto kern = srf_u16_ring_kernel<DEG, OUTV, BATCH, FASTU, APPLY>;
#ifdef HSR_PHASE_STAMPS
  const_cast<SrfArgs&>(a).stamps = g_stamp_buffer;
#endif
  static thread_local size_t configured = 0;
  ensure_dynamic_lds(kern, lds, &configured);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, stream, a);
  HSR_LAUNCH_CHECK("srf_u16_ring_kernel");
  return HSR_OK;
}

template <int DEG, bool FAST, bool OUTV, bool BATCH>
static int launch_srf_u16(const SrfArgs& a, int grid, hipStream_t stream) {
  const size_t lds = (size_t)64 * a.B * 2 + (64 + (BATCH ? 16 : 0)) * sizeof(uint32_t) + (size_t)a.wtaps * 4 +
                     (OUTV ? (size_t)64 * stage_row(a.nb, (int)a.out_ps) * 4 : 0) + (DEG > 0 ? target_stage_bytes(a.nb) : 0);
  auto kern = srf_u16_kernel<DEG, FAST, OUTV, BATCH>;
  static thread_local size_t configured = 0;
  ensure_dynamic_lds(kern, lds, &configured);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, stream, a);
  HSR_LAUNCH_CHECK("srf_u16_kernel");
  return HSR_OK;
}

static bool out_rows_vectorised(const SrfArgs& a, const float* out) {
  return a.out_bs == 1 && (a.out_ps & 3) == 0 && a.out_ps <= HSR_MAX_BANDS && (((uintptr_t)out) & 15) == 0;
}

// two group buffers must fit twice per CU next to the weights and the output slab: B <= ~300 spectral samples
static bool u16_ring_fits(const SrfArgs& a, bool outv) {
  const size_t ring_lds = (size_t)2 * 64 * a.B * 2 + 512 + (size_t)a.wtaps * 4 + (outv ? (size_t)64 * stage_row(a.nb, (int)a.out_ps) * 4 : 0);
  return ring_lds <= 80 * 1024;
}

template <int DEG>
static int dispatch_u16_deg(const SrfArgs& a, bool fast, bool ring, int grid, hipStream_t s) {
  const bool outv = out_rows_vectorised(a, a.one.pseudo_dev);
  if (a.apply_x != nullptr || a.lazy_partials != nullptr) {   // the fused pipeline's launch: only the ring kernel carries it
    if constexpr (DEG > 0) {
      if (!(fast && ring && outv && a.wtaps > 0 && u16_ring_fits(a, true) && 2 * 64 * a.B * 2 >= 12 * 1024)) {
        set_error("hsr_srf_integrate_moments_u16_apply: this launch cannot carry an apply job (needs the ring kernel: 16-byte aligned "
                  "cube, weights in LDS, pixel-major rows, 48 <= B <= ~300)");
        return HSR_ERR_UNSUPPORTED;
      }
      return a.u16_fast ? launch_srf_u16_ring<DEG, true, false, true, true>(a, grid, s)
                        : launch_srf_u16_ring<DEG, true, false, false, true>(a, grid, s);
    }
  }
  if (fast && ring && u16_ring_fits(a, outv)) {
    if (outv && a.u16_fast && a.wtaps > 0) return launch_srf_u16_ring<DEG, true, false, true>(a, grid, s);
    return outv ? launch_srf_u16_ring<DEG, true, false, false>(a, grid, s) : launch_srf_u16_ring<DEG, false, false, false>(a, grid, s);
  }
  if (outv) return fast ? launch_srf_u16<DEG, true, true, false>(a, grid, s) : launch_srf_u16<DEG, false, true, false>(a, grid, s);
  return fast ? launch_srf_u16<DEG, true, false, false>(a, grid, s) : launch_srf_u16<DEG, false, false, false>(a, grid, s);
}

static int dispatch_u16(const SrfArgs& a, int deg, bool fast, bool ring, int grid, hipStream_t s) {
  switch (deg) {
    case 0: return dispatch_u16_deg<0>(a, fast, ring, grid, s);
    case 1: return dispatch_u16_deg<1>(a, fast, ring, grid, s);
    case 2: return dispatch_u16_deg<2>(a, fast, ring, grid, s);
    case 3: return dispatch_u16_deg<3>(a, fast, ring, grid, s);
    case 4: return dispatch_u16_deg<4>(a, fast, ring, grid, s);
  }
  set_error("hsr_srf_integrate_moments_u16: deg=%d outside [1,%d]", deg, HSR_MAX_DEG);
  return HSR_ERR_UNSUPPORTED;
}

template <int DEG, bool FAST, bool WLDS, int P, bool OUTV, bool BATCH, bool APPLY = false>
static int launch_srf(const SrfArgs& a, int grid, hipStream_t stream) {
  const size_t lds = (size_t)P * a.ldsB * 4 + (64 + (BATCH ? 16 : 0)) * sizeof(uint32_t) + (WLDS ? (size_t)a.wtaps * 4 : 0) +
                     (OUTV ? (size_t)P * stage_row(a.nb, (int)a.out_ps) * 4 : 0) + (DEG > 0 ? target_stage_bytes(a.nb) : 0);
  auto kern = srf_kernel<DEG, FAST, WLDS, P, OUTV, BATCH, APPLY>;
  static thread_local size_t configured = 0;
  ensure_dynamic_lds(kern, lds, &configured);
#ifdef HSR_PHASE_STAMPS
  const_cast<SrfArgs&>(a).stamps = g_stamp_buffer;
  const_cast<SrfArgs&>(a).stamps2 = g_stamp_buffer2;
  const_cast<SrfArgs&>(a).stamps3 = g_stamp_buffer3;
#endif
  hipLaunchKernelGGL(kern, dim3(grid), dim3(8 * P), lds, stream, a);
  HSR_LAUNCH_CHECK("srf_kernel");
  return HSR_OK;
}

template <int DEG, int P>
static int dispatch_fast(const SrfArgs& a, bool fast, int grid, hipStream_t s) {
  // pixel-major output with a 16-byte friendly row: stage the slab in LDS and flush it vectorised
  const bool outv = out_rows_vectorised(a, a.one.pseudo_dev);
  if (a.wtaps == 0) return launch_srf<DEG, false, false, 64, false, false>(a, grid, s);  // rare fallback: one generic kernel
  if (a.apply_x != nullptr || a.lazy_partials != nullptr) {   // srf_common has checked that this launch can carry it (outv, weights in LDS, DEG > 0)
    if constexpr (DEG > 0)
      return fast ? launch_srf<DEG, true, true, P, true, false, true>(a, grid, s) : launch_srf<DEG, false, true, P, true, false, true>(a, grid, s);
  }
  if (outv) return fast ? launch_srf<DEG, true, true, P, true, false>(a, grid, s) : launch_srf<DEG, false, true, P, true, false>(a, grid, s);
  return fast ? launch_srf<DEG, true, true, P, false, false>(a, grid, s) : launch_srf<DEG, false, true, P, false, false>(a, grid, s);
}

template <int P>
static int dispatch_deg(const SrfArgs& a, int deg, bool fast, int grid, hipStream_t s) {
  switch (deg) {
    case 0: return dispatch_fast<0, P>(a, fast, grid, s);
    case 1: return dispatch_fast<1, P>(a, fast, grid, s);
    case 2: return dispatch_fast<2, P>(a, fast, grid, s);
    case 3: return dispatch_fast<3, P>(a, fast, grid, s);
    case 4: return dispatch_fast<4, P>(a, fast, grid, s);
  }
  set_error("hsr_srf_integrate_moments: deg=%d outside [1,%d]", deg, HSR_MAX_DEG);
  return HSR_ERR_UNSUPPORTED;
}

// batch launches: pixel-major rows staged in LDS, weights in LDS, 64-pixel groups
template <int DEG>
static int dispatch_batch_deg(const SrfArgs& a, bool fast, bool ring, int grid, hipStream_t s) {
  if (a.u16) {
    if (fast && ring && u16_ring_fits(a, true))
      return a.u16_fast ? launch_srf_u16_ring<DEG, true, true, true>(a, grid, s) : launch_srf_u16_ring<DEG, true, true, false>(a, grid, s);
    return fast ? launch_srf_u16<DEG, true, true, true>(a, grid, s) : launch_srf_u16<DEG, false, true, true>(a, grid, s);
  }
  return fast ? launch_srf<DEG, true, true, 64, true, true>(a, grid, s) : launch_srf<DEG, false, true, 64, true, true>(a, grid, s);
}

static int dispatch_batch(const SrfArgs& a, int deg, bool fast, bool ring, int grid, hipStream_t s) {
  switch (deg) {
    case 0: return dispatch_batch_deg<0>(a, fast, ring, grid, s);
    case 1: return dispatch_batch_deg<1>(a, fast, ring, grid, s);
    case 2: return dispatch_batch_deg<2>(a, fast, ring, grid, s);
    case 3: return dispatch_batch_deg<3>(a, fast, ring, grid, s);
    case 4: return dispatch_batch_deg<4>(a, fast, ring, grid, s);
  }
  set_error("hsr_srf_integrate_moments_batched: deg=%d outside [0,%d]", deg, HSR_MAX_DEG);
  return HSR_ERR_UNSUPPORTED;
}

// Band table of a launch: validated supports, LDS weight segments, band -> (group, slot) assignment.
static int srf_prepare_bands(SrfArgs& a, const int32_t* k0, const int32_t* klen) {
  HSR_REQUIRE(a.wn && k0 && klen, HSR_ERR_INVALID, "hsr_srf_integrate: NULL pointer");
  HSR_REQUIRE(a.B >= 1 && a.B <= HSR_MAX_SPECTRAL, HSR_ERR_UNSUPPORTED,
              "hsr_srf_integrate: B=%d outside [1,%d]", a.B, HSR_MAX_SPECTRAL);
  HSR_REQUIRE(a.nb >= 1 && a.nb <= HSR_MAX_BANDS, HSR_ERR_UNSUPPORTED,
              "hsr_srf_integrate: nb=%d outside [1,%d]", a.nb, HSR_MAX_BANDS);
  for (int b = 0; b < a.nb; ++b) {
    HSR_REQUIRE(k0[b] >= 0 && klen[b] >= 0 && k0[b] + klen[b] <= a.B, HSR_ERR_INVALID,
                "hsr_srf_integrate: support of band %d = [%d,%d) outside [0,%d)", b, k0[b], k0[b] + klen[b], a.B);
    a.bands.k0[b] = k0[b];
    a.bands.klen[b] = klen[b];
    a.bands.woff[b] = 0;
  }
  for (int b = a.nb; b < HSR_MAX_BANDS; ++b) a.bands.k0[b] = a.bands.klen[b] = a.bands.woff[b] = 0;
  // LDS weight segments: widen every support to whole 16-tap chunks that stay inside [0, B) (the
  // added taps have weight 0 in the dense rows of wn); if that is impossible (B < 16 * chunks) or the
  // taps do not fit the reserved LDS, the kernel reads the weights from global memory instead.
  {
    int32_t sk0[HSR_MAX_BANDS], skl[HSR_MAX_BANDS], off[HSR_MAX_BANDS], total = 0;
    bool fits = true;
    for (int b = 0; b < a.nb && fits; ++b) {
      const int nc = (klen[b] + kTapChunk - 1) / kTapChunk;
      int s0 = k0[b];
      if (s0 + kTapChunk * nc > a.B) s0 = a.B - kTapChunk * nc;
      if (s0 < 0) fits = false;
      sk0[b] = s0;
      skl[b] = kTapChunk * nc;
      off[b] = total;
      total += kTapChunk * nc;
    }
    if (fits && total <= kWeightCap) {
      for (int b = 0; b < a.nb; ++b) {
        a.bands.k0[b] = sk0[b];
        a.bands.klen[b] = skl[b];
        a.bands.woff[b] = off[b];
      }
      a.wtaps = total > 0 ? total : kTapChunk;
    } else {
      a.wtaps = 0;
    }
  }
  // Band -> (group, slot): the slowest of the 8 groups sets the length of the dot-product phase, so deal the
  // bands out longest first to the least loaded group (2 slots each).  Any assignment gives identical bits.
  {
    int order[HSR_MAX_BANDS], load[8] = {0, 0, 0, 0, 0, 0, 0, 0}, used[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int g = 0; g < 8; ++g) a.bands.band_of[g][0] = a.bands.band_of[g][1] = -1;
    for (int b = 0; b < a.nb; ++b) order[b] = b;
    for (int i = 1; i < a.nb; ++i)   // insertion sort by descending tap count (stable)
      for (int k = i; k > 0 && a.bands.klen[order[k]] > a.bands.klen[order[k - 1]]; --k) {
        const int t = order[k]; order[k] = order[k - 1]; order[k - 1] = t;
      }
    for (int i = 0; i < a.nb; ++i) {
      int best = -1;
      for (int g = 0; g < 8; ++g)
        if (used[g] < 2 && (best < 0 || load[g] < load[best])) best = g;
      a.bands.band_of[best][used[best]++] = (int8_t)order[i];
      load[best] += a.bands.klen[order[i]] + 1;
    }
  }
  a.ldsB = (a.B & 1) ? a.B : a.B + 1;
  return HSR_OK;
}

static int srf_common(SrfArgs& a, const int32_t* k0, const int32_t* klen, int32_t deg, const hsr_srf_options* opts,
                      hipStream_t stream) {
  SrfTuning tn;
  int rc = srf_tuning(opts, &tn, "hsr_srf_integrate");
  if (rc != HSR_OK) return rc;
  HSR_REQUIRE(a.one.cube_dev && a.one.pseudo_dev, HSR_ERR_INVALID, "hsr_srf_integrate: NULL pointer");
  HSR_REQUIRE(a.one.npix >= 0, HSR_ERR_INVALID, "hsr_srf_integrate: npix < 0");
  rc = srf_prepare_bands(a, k0, klen);
  if (rc != HSR_OK) return rc;
  HSR_REQUIRE((a.out_ps == 1 && a.out_bs >= a.one.npix) || (a.out_bs == 1 && a.out_ps >= a.nb), HSR_ERR_INVALID,
              "hsr_srf_integrate: output strides (%lld, %lld) are neither band-major nor pixel-major",
              (long long)a.out_bs, (long long)a.out_ps);
  HSR_REQUIRE(((uintptr_t)a.one.cube_dev & (a.u16 ? 1 : 3)) == 0, HSR_ERR_INVALID, "hsr_srf_integrate: cube not %d-byte aligned",
              a.u16 ? 2 : 4);
  if (a.one.npix == 0) return HSR_OK;
  HSR_REQUIRE((a.apply_x == nullptr && a.lazy_partials == nullptr) || a.wtaps > 0, HSR_ERR_UNSUPPORTED,
              "hsr_srf_integrate_moments_apply: the weight table does not fit LDS, this launch cannot carry an apply job");
  a.u16_fast = a.u16 && tn.u16_fast;
  int P = a.u16 ? 64 : tn.tile_pixels;
  if (a.wtaps == 0) P = 64;  // the generic fallback kernel exists for 64-pixel groups only
  a.one.ngroups = (int32_t)((a.one.npix + P - 1) / P);
  a.one.slots = srf_slots(a.one.npix, P, tn.reserved_cus);
  a.one.slot = 0;
  a.nunits = a.one.slots;
  const bool aligned = (((uintptr_t)a.one.cube_dev) & 15) == 0;
  if (a.u16) return dispatch_u16(a, deg, aligned, tn.u16_ring, a.one.slots, stream);
  const bool fast = (a.B & 1) && aligned;
  return dispatch_deg<64>(a, deg, fast, a.one.slots, stream);
}

}  // namespace hsr

extern "C" int hsr_partial_slots(int64_t npix, const hsr_srf_options* opts) {
  hsr::SrfTuning tn;
  if (hsr::srf_tuning(opts, &tn, "hsr_partial_slots") != HSR_OK) return -1;
  return hsr::srf_slots(npix, tn.tile_pixels, tn.reserved_cus);
}

// Can a K1 launch of this geometry carry an apply job / a tail fit (hsr_srf_integrate_moments[_u16]_apply)?  The conditions of
// dispatch_fast / dispatch_u16_deg and srf_common in one place, without a launch, so that hsr_pipeline_create_fused /
// _exchange fail at creation - where the caller can still fall back to the two-slot pipeline - instead of at the first
// carrying launch, with tiles in flight.  (What it cannot see is the cube pointer: uint16 cubes must be 16-byte aligned.)
extern "C" int hsr_srf_fused_launch_supported(int32_t cube_dtype, int32_t B, int32_t nb, const int32_t* k0, const int32_t* klen,
                                              int64_t out_ps, int32_t deg, const hsr_srf_options* opts) {
  const char* who = "hsr_srf_fused_launch_supported";
  HSR_REQUIRE(cube_dtype == 0 || cube_dtype == 2, HSR_ERR_INVALID, "%s: cube_dtype %d (0 float32, 2 uint16)", who, cube_dtype);
  HSR_REQUIRE(deg >= 1 && deg <= HSR_MAX_DEG, HSR_ERR_UNSUPPORTED, "%s: deg=%d outside [1,%d]", who, deg, HSR_MAX_DEG);
  HSR_REQUIRE((out_ps & 3) == 0 && out_ps >= nb && out_ps <= HSR_MAX_BANDS, HSR_ERR_UNSUPPORTED,
              "%s: pixel-major rows of 4 / 8 / 12 / 16 floats needed, got %lld", who, (long long)out_ps);
  hsr::SrfTuning tn;
  int rc = hsr::srf_tuning(opts, &tn, who);
  if (rc != HSR_OK) return rc;
  hsr::SrfArgs a{};
  static const float dummy = 0.0f;
  a.wn = &dummy;                   // srf_prepare_bands only tests it
  a.B = B;
  a.nb = nb;
  a.out_bs = 1;
  a.out_ps = out_ps;
  rc = hsr::srf_prepare_bands(a, k0, klen);
  if (rc != HSR_OK) return rc;
  HSR_REQUIRE(a.wtaps > 0, HSR_ERR_UNSUPPORTED, "%s: the weight taps of the %d bands do not fit the %d floats of LDS reserved for them",
              who, nb, hsr::kWeightCap);
  if (cube_dtype == 2) {
    HSR_REQUIRE(tn.u16_ring && hsr::u16_ring_fits(a, true) && 2 * 64 * B * 2 >= 12 * 1024, HSR_ERR_UNSUPPORTED,
                "%s: uint16 tiles ride only in the double-buffered kernel (u16_single_buffer = 0, 48 <= B, two %d-byte group buffers + "
                "%d taps + the %lld-float rows within 80 KB of LDS)", who, 64 * B * 2, a.wtaps, (long long)out_ps);
  } else {
    const size_t lds = (size_t)64 * a.ldsB * 4 + 64 * sizeof(uint32_t) + (size_t)a.wtaps * 4 + (size_t)64 * hsr::stage_row(nb, (int)out_ps) * 4 + hsr::target_stage_bytes(nb);
    HSR_REQUIRE(lds <= 160 * 1024, HSR_ERR_UNSUPPORTED, "%s: a 64-pixel group of B=%d samples needs %zu bytes of LDS (160 KB per workgroup)", who, B, lds);
  }
  return HSR_OK;
}

extern "C" int hsr_srf_integrate(const float* cube_dev, int64_t npix, int32_t B, const float* wn_dev,
                                 const int32_t* k0, const int32_t* klen, int32_t nb, float* out_dev,
                                 int64_t out_bs, int64_t out_ps, const hsr_srf_options* opts, hsr_stream_t stream) {
  hsr::SrfArgs a{};
  a.one.cube_dev = cube_dev;
  a.one.npix = npix;
  a.one.pseudo_dev = out_dev;
  a.B = B;
  a.wn = wn_dev;
  a.nb = nb;
  a.out_bs = out_bs;
  a.out_ps = out_ps;
  return hsr::srf_common(a, k0, klen, 0, opts, (hipStream_t)stream);
}

// K1+K2 entry shared by the float32 / uint16 and the plain / fused-fit forms.
static int srf_moments_entry(const char* who, const void* cube_dev, bool u16, float scale, int32_t nodata, int64_t npix,
                             int32_t B, const float* wn_dev, const int32_t* k0, const int32_t* klen, int32_t nb,
                             float* out_dev, int64_t out_bs, int64_t out_ps, const float* real_dev, int64_t real_bs,
                             int64_t real_ps, const uint8_t* mask_dev, float min_x, float min_y, int32_t deg,
                             double* partials_dev, int32_t* slots_out, const hsr_fused_fit* fit,
                             const hsr_srf_options* opts, hsr_stream_t stream, const hsr_apply_job* job = nullptr) {
  HSR_REQUIRE(deg >= 1 && deg <= HSR_MAX_DEG, HSR_ERR_UNSUPPORTED, "%s: deg=%d outside [1,%d]", who, deg, HSR_MAX_DEG);
  HSR_REQUIRE(real_dev && partials_dev, HSR_ERR_INVALID, "%s: NULL pointer", who);
  HSR_REQUIRE((real_ps == 1 && real_bs >= npix) || (real_bs == 1 && real_ps >= nb), HSR_ERR_INVALID,
              "%s: real strides (%lld, %lld) are neither band-major nor pixel-major", who, (long long)real_bs,
              (long long)real_ps);
  HSR_REQUIRE(npix > 0, HSR_ERR_INVALID, "%s: npix must be > 0", who);
  HSR_REQUIRE(!u16 || nodata <= 0xffff, HSR_ERR_INVALID, "%s: nodata=%d is not a uint16 value (negative = none)", who, nodata);
  hsr::SrfArgs a{};
  if (fit) {
    HSR_REQUIRE(fit->group_partials_dev && fit->tickets_dev && fit->moments_dev && fit->coeffs_dev, HSR_ERR_INVALID,
                "%s: NULL pointer in hsr_fused_fit", who);
    HSR_REQUIRE(fit->min_count >= 0, HSR_ERR_INVALID, "%s: min_count < 0", who);
    a.fit.gpart = fit->group_partials_dev;
    a.fit.tickets = fit->tickets_dev;
    a.fit.moments = fit->moments_dev;
    a.fit.coeffs = fit->coeffs_dev;
    a.fit.min_count = (long long)fit->min_count;
  }
  a.one.cube_dev = cube_dev;
  a.one.npix = npix;
  a.one.pseudo_dev = out_dev;
  a.one.real_dev = real_dev;
  a.one.mask_dev = mask_dev;
  a.one.part_dev = partials_dev;
  if (u16) {
    a.u16 = 1;
    a.scale = scale;
    a.nodata = nodata < 0 ? 0x10000u : (uint32_t)nodata;
  }
  a.B = B;
  a.wn = wn_dev;
  a.nb = nb;
  a.out_bs = out_bs;
  a.out_ps = out_ps;
  a.real_bs = real_bs;
  a.real_ps = real_ps;
  a.min_x = min_x;
  a.min_y = min_y;
  if (job) {
    HSR_REQUIRE((job->x_dev == nullptr && job->fit_partials_dev) || (job->x_dev && job->out_dev && job->coeffs_dev && job->npix >= 1),
                HSR_ERR_INVALID, "%s: NULL pointer or npix < 1 in hsr_apply_job", who);
    HSR_REQUIRE(out_bs == 1 && (out_ps & 3) == 0 && out_ps <= HSR_MAX_BANDS && job->npix * (out_ps >> 2) < ((int64_t)1 << 31) &&
                    ((((uintptr_t)job->x_dev) | ((uintptr_t)job->out_dev) | ((uintptr_t)out_dev)) & 15) == 0,
                HSR_ERR_UNSUPPORTED, "%s: a launch carries an apply job only for 16-byte aligned pixel-major rows of 4, 8, 12 or 16 floats", who);
    a.apply_x = job->x_dev;
    a.apply_out = job->out_dev;
    a.apply_coeffs = job->coeffs_dev;
    a.apply_mask = job->mask_dev;
    a.apply_npix = job->npix;
    a.apply_clip = job->clip;
    if (job->fit_partials_dev) {
      HSR_REQUIRE(job->fit_moments_dev && job->fit_coeffs_dev && job->fit_counter_dev && job->fit_slots >= 1 && job->fit_min_count >= 0,
                  HSR_ERR_INVALID, "%s: incomplete tail fit in hsr_apply_job", who);
      // one ticket per workgroup, one band per ticket: a launch of fewer workgroups than bands would leave bands unfitted
      HSR_REQUIRE(hsr_partial_slots(npix, opts) >= nb, HSR_ERR_UNSUPPORTED,
                  "%s: a launch of %d workgroups cannot carry the tail fit of %d bands (hsr_partial_slots(npix) >= nb needed); run "
                  "hsr_moments_reduce_solve for that tile instead", who, hsr_partial_slots(npix, opts), nb);
      a.lazy_partials = job->fit_partials_dev;
      a.lazy_slots = job->fit_slots;
      a.lazy_min_count = (long long)job->fit_min_count;
      a.lazy_moments = job->fit_moments_dev;
      a.lazy_coeffs = job->fit_coeffs_dev;
      a.lazy_counter = job->fit_counter_dev;
      a.lazy_base = job->fit_ticket_base;
      a.lazy_ready = job->fit_ready_dev;
      if (job->fit_group_tiles > 1) {
        HSR_REQUIRE(job->fit_group_tiles <= 64 && job->fit_group_index >= 0 && job->fit_group_index < job->fit_group_tiles &&
                        job->fit_group_moments_dev && job->fit_group_total_dev && job->fit_ready_dev == nullptr, HSR_ERR_INVALID,
                    "%s: group fit of %d tiles (at most 64), index %d, or NULL group buffers, or combined with fit_ready_dev", who,
                    job->fit_group_tiles, job->fit_group_index);
        a.lazy_group_T = job->fit_group_tiles;
        a.lazy_group_index = job->fit_group_index;
        a.lazy_group_moments = job->fit_group_moments_dev;
        a.lazy_group_total = job->fit_group_total_dev;
      }
    }
    if (job->x_dev && job->coeffs_ready_dev) {
      a.apply_ready = job->coeffs_ready_dev;
      a.apply_ready_value = job->coeffs_ready_value;
    }
    a.sync_error = job->sync_error_dev;
  }
  int rc = hsr::srf_common(a, k0, klen, deg, opts, (hipStream_t)stream);
  if (rc == HSR_OK && slots_out) *slots_out = a.one.slots;
  return rc;
}

extern "C" int hsr_srf_integrate_moments(const float* cube_dev, int64_t npix, int32_t B, const float* wn_dev,
                                         const int32_t* k0, const int32_t* klen, int32_t nb, float* out_dev,
                                         int64_t out_bs, int64_t out_ps, const float* real_dev, int64_t real_bs,
                                         int64_t real_ps, const uint8_t* mask_dev, float min_x, float min_y, int32_t deg,
                                         double* partials_dev, int32_t* slots_out, const hsr_srf_options* opts,
                                         hsr_stream_t stream) {
  return srf_moments_entry("hsr_srf_integrate_moments", cube_dev, false, 0.0f, -1, npix, B, wn_dev, k0, klen, nb, out_dev,
                           out_bs, out_ps, real_dev, real_bs, real_ps, mask_dev, min_x, min_y, deg, partials_dev, slots_out,
                           nullptr, opts, stream);
}

extern "C" int hsr_srf_integrate_moments_apply(const float* cube_dev, int64_t npix, int32_t B, const float* wn_dev,
                                               const int32_t* k0, const int32_t* klen, int32_t nb, float* out_dev,
                                               int64_t out_bs, int64_t out_ps, const float* real_dev, int64_t real_bs,
                                               int64_t real_ps, const uint8_t* mask_dev, float min_x, float min_y, int32_t deg,
                                               double* partials_dev, int32_t* slots_out, const hsr_srf_options* opts,
                                               const hsr_apply_job* job, hsr_stream_t stream) {
  return srf_moments_entry("hsr_srf_integrate_moments_apply", cube_dev, false, 0.0f, -1, npix, B, wn_dev, k0, klen, nb, out_dev,
                           out_bs, out_ps, real_dev, real_bs, real_ps, mask_dev, min_x, min_y, deg, partials_dev, slots_out,
                           nullptr, opts, stream, job);
}

extern "C" int hsr_srf_integrate_fit(const float* cube_dev, int64_t npix, int32_t B, const float* wn_dev,
                                     const int32_t* k0, const int32_t* klen, int32_t nb, float* out_dev, int64_t out_bs,
                                     int64_t out_ps, const float* real_dev, int64_t real_bs, int64_t real_ps,
                                     const uint8_t* mask_dev, float min_x, float min_y, int32_t deg, double* partials_dev,
                                     int32_t* slots_out, const hsr_fused_fit* fit, const hsr_srf_options* opts,
                                     hsr_stream_t stream) {
  HSR_REQUIRE(fit, HSR_ERR_INVALID, "hsr_srf_integrate_fit: NULL hsr_fused_fit");
  return srf_moments_entry("hsr_srf_integrate_fit", cube_dev, false, 0.0f, -1, npix, B, wn_dev, k0, klen, nb, out_dev,
                           out_bs, out_ps, real_dev, real_bs, real_ps, mask_dev, min_x, min_y, deg, partials_dev, slots_out,
                           fit, opts, stream);
}

extern "C" int hsr_srf_integrate_u16(const uint16_t* cube_dev, int64_t npix, int32_t B, float scale, int32_t nodata,
                                     const float* wn_dev, const int32_t* k0, const int32_t* klen, int32_t nb,
                                     float* out_dev, int64_t out_bs, int64_t out_ps, const hsr_srf_options* opts,
                                     hsr_stream_t stream) {
  HSR_REQUIRE(nodata <= 0xffff, HSR_ERR_INVALID, "hsr_srf_integrate_u16: nodata=%d is not a uint16 value (negative = none)", nodata);
  hsr::SrfArgs a{};
  a.one.cube_dev = cube_dev;
  a.one.npix = npix;
  a.one.pseudo_dev = out_dev;
  a.u16 = 1;
  a.scale = scale;
  a.nodata = nodata < 0 ? 0x10000u : (uint32_t)nodata;
  a.B = B;
  a.wn = wn_dev;
  a.nb = nb;
  a.out_bs = out_bs;
  a.out_ps = out_ps;
  return hsr::srf_common(a, k0, klen, 0, opts, (hipStream_t)stream);
}

extern "C" int hsr_srf_integrate_moments_u16(const uint16_t* cube_dev, int64_t npix, int32_t B, float scale,
                                             int32_t nodata, const float* wn_dev, const int32_t* k0,
                                             const int32_t* klen, int32_t nb, float* out_dev, int64_t out_bs,
                                             int64_t out_ps, const float* real_dev, int64_t real_bs, int64_t real_ps,
                                             const uint8_t* mask_dev, float min_x, float min_y, int32_t deg,
                                             double* partials_dev, int32_t* slots_out, const hsr_srf_options* opts,
                                             hsr_stream_t stream) {
  return srf_moments_entry("hsr_srf_integrate_moments_u16", cube_dev, true, scale, nodata, npix, B, wn_dev, k0, klen, nb,
                           out_dev, out_bs, out_ps, real_dev, real_bs, real_ps, mask_dev, min_x, min_y, deg, partials_dev,
                           slots_out, nullptr, opts, stream);
}

extern "C" int hsr_srf_integrate_moments_u16_apply(const uint16_t* cube_dev, int64_t npix, int32_t B, float scale,
                                                   int32_t nodata, const float* wn_dev, const int32_t* k0,
                                                   const int32_t* klen, int32_t nb, float* out_dev, int64_t out_bs,
                                                   int64_t out_ps, const float* real_dev, int64_t real_bs, int64_t real_ps,
                                                   const uint8_t* mask_dev, float min_x, float min_y, int32_t deg,
                                                   double* partials_dev, int32_t* slots_out, const hsr_srf_options* opts,
                                                   const hsr_apply_job* job, hsr_stream_t stream) {
  return srf_moments_entry("hsr_srf_integrate_moments_u16_apply", cube_dev, true, scale, nodata, npix, B, wn_dev, k0, klen, nb,
                           out_dev, out_bs, out_ps, real_dev, real_bs, real_ps, mask_dev, min_x, min_y, deg, partials_dev,
                           slots_out, nullptr, opts, stream, job);
}

extern "C" int hsr_srf_integrate_fit_u16(const uint16_t* cube_dev, int64_t npix, int32_t B, float scale, int32_t nodata,
                                         const float* wn_dev, const int32_t* k0, const int32_t* klen, int32_t nb,
                                         float* out_dev, int64_t out_bs, int64_t out_ps, const float* real_dev,
                                         int64_t real_bs, int64_t real_ps, const uint8_t* mask_dev, float min_x,
                                         float min_y, int32_t deg, double* partials_dev, int32_t* slots_out,
                                         const hsr_fused_fit* fit, const hsr_srf_options* opts, hsr_stream_t stream) {
  HSR_REQUIRE(fit, HSR_ERR_INVALID, "hsr_srf_integrate_fit_u16: NULL hsr_fused_fit");
  return srf_moments_entry("hsr_srf_integrate_fit_u16", cube_dev, true, scale, nodata, npix, B, wn_dev, k0, klen, nb,
                           out_dev, out_bs, out_ps, real_dev, real_bs, real_ps, mask_dev, min_x, min_y, deg, partials_dev,
                           slots_out, fit, opts, stream);
}

// ---- batched small tiles ---------------------------------------------------------------------------------------
extern "C" size_t hsr_batch_partials_bytes(int64_t nunits, int32_t nb, int32_t deg) {
  if (nunits < 1 || nb < 1 || nb > HSR_MAX_BANDS || deg < 0 || deg > HSR_MAX_DEG) return 0;
  return (size_t)nunits * nb * (deg > 0 ? hsr::moment_count(deg) : 1) * sizeof(double);
}

extern "C" int hsr_batch_plan(hsr_batch_tile* tiles, int32_t ntiles, int32_t nb, int32_t deg, double* partials_dev,
                              const hsr_srf_options* opts, hsr_batch_unit* units_out, int64_t units_capacity,
                              hsr_batch_info* info) {
  HSR_REQUIRE(tiles && info && ntiles >= 1, HSR_ERR_INVALID, "hsr_batch_plan: NULL pointer or no tiles");
  HSR_REQUIRE(nb >= 1 && nb <= HSR_MAX_BANDS && deg >= 0 && deg <= HSR_MAX_DEG, HSR_ERR_UNSUPPORTED,
              "hsr_batch_plan: nb=%d deg=%d", nb, deg);
  hsr::SrfTuning tn;
  int rc = hsr::srf_tuning(opts, &tn, "hsr_batch_plan");
  if (rc != HSR_OK) return rc;
  const int M = deg > 0 ? hsr::moment_count(deg) : 1;
  int64_t slot0 = 0, pixels = 0, max_npix = 0;
  int aligned = 1;
  for (int32_t t = 0; t < ntiles; ++t) {
    hsr_batch_tile& tl = tiles[t];
    HSR_REQUIRE(tl.npix > 0 && tl.npix < ((int64_t)1 << 37), HSR_ERR_INVALID, "hsr_batch_plan: tile %d has npix=%lld", t, (long long)tl.npix);
    HSR_REQUIRE(tl.cube_dev && tl.pseudo_dev, HSR_ERR_INVALID, "hsr_batch_plan: tile %d has a NULL cube or output", t);
    HSR_REQUIRE(deg == 0 || tl.real_dev, HSR_ERR_INVALID, "hsr_batch_plan: tile %d has no real-S2 target (deg > 0)", t);
    HSR_REQUIRE((((uintptr_t)tl.pseudo_dev) & 15) == 0 && (tl.matched_dev == nullptr || (((uintptr_t)tl.matched_dev) & 15) == 0),
                HSR_ERR_INVALID, "hsr_batch_plan: tile %d: image rows must be 16-byte aligned", t);
    tl.ngroups = (int32_t)((tl.npix + 63) / 64);
    tl.slots = hsr::srf_slots(tl.npix, 64, tn.reserved_cus);     // batches always use 64-pixel groups
    tl.slot0 = slot0;
    slot0 += tl.slots;
    pixels += tl.npix;
    if (tl.npix > max_npix) max_npix = tl.npix;
    if (((uintptr_t)tl.cube_dev) & 15) aligned = 0;
  }
  HSR_REQUIRE(slot0 < ((int64_t)1 << 31), HSR_ERR_UNSUPPORTED, "hsr_batch_plan: %lld work units exceed 2^31", (long long)slot0);
  info->nunits = slot0;
  info->total_pixels = pixels;
  info->max_npix = max_npix;
  info->ntiles = ntiles;
  info->aligned16 = aligned;
  if (units_out == nullptr) return HSR_OK;
  HSR_REQUIRE(units_capacity >= slot0, HSR_ERR_INVALID, "hsr_batch_plan: unit table holds %lld records, %lld needed",
              (long long)units_capacity, (long long)slot0);
  HSR_REQUIRE(deg == 0 || partials_dev, HSR_ERR_INVALID, "hsr_batch_plan: partials_dev is NULL (deg > 0)");
  // Units in descending order of their group count (stable; ties keep tile order), dealt round-robin to the
  // workgroups by the kernel (workgroup b runs units b, b + grid, ...): long units first, short ones fill the tail.
  int32_t maxg = 0;
  for (int32_t t = 0; t < ntiles; ++t) {
    const int32_t per = (tiles[t].ngroups + tiles[t].slots - 1) / tiles[t].slots;
    if (per > maxg) maxg = per;
  }
  int64_t n = 0;
  // a tile's units have ceil or floor(ngroups / slots) groups: two passes per distinct length would be exact; lengths
  // are small integers, so bucket by length from the longest down
  for (int32_t len = maxg; len >= 1; --len) {
    for (int32_t t = 0; t < ntiles; ++t) {
      const hsr_batch_tile& tl = tiles[t];
      for (int32_t s = 0; s < tl.slots; ++s) {
        const int32_t groups = (tl.ngroups - s + tl.slots - 1) / tl.slots;
        if (groups != len) continue;
        hsr_batch_unit& u = units_out[n++];
        u.cube_dev = tl.cube_dev;
        u.real_dev = tl.real_dev;
        u.mask_dev = tl.mask_dev;
        u.pseudo_dev = tl.pseudo_dev;
        u.part_dev = partials_dev ? partials_dev + ((size_t)tl.slot0 + s) * nb * M : nullptr;
        u.npix = tl.npix;
        u.slots = tl.slots;
        u.slot = s;
        u.ngroups = tl.ngroups;
        u.reserved = 0;
      }
    }
  }
  return HSR_OK;
}

extern "C" int hsr_srf_integrate_moments_batched(const hsr_batch_unit* units_dev, const hsr_batch_info* info,
                                                 int32_t cube_dtype, float scale, int32_t nodata, int32_t B,
                                                 const float* wn_dev, const int32_t* k0, const int32_t* klen, int32_t nb,
                                                 int32_t out_row, int32_t real_row, float min_x, float min_y,
                                                 int32_t deg, const hsr_srf_options* opts, hsr_stream_t stream) {
  HSR_REQUIRE(units_dev && info, HSR_ERR_INVALID, "hsr_srf_integrate_moments_batched: NULL pointer");
  HSR_REQUIRE(info->nunits >= 1 && info->nunits < ((int64_t)1 << 31), HSR_ERR_INVALID,
              "hsr_srf_integrate_moments_batched: nunits=%lld", (long long)info->nunits);
  HSR_REQUIRE(cube_dtype == 0 || cube_dtype == 2, HSR_ERR_INVALID, "hsr_srf_integrate_moments_batched: cube_dtype must be 0 (float32) or 2 (uint16)");
  HSR_REQUIRE(deg >= 0 && deg <= HSR_MAX_DEG, HSR_ERR_UNSUPPORTED, "hsr_srf_integrate_moments_batched: deg=%d outside [0,%d]", deg, HSR_MAX_DEG);
  HSR_REQUIRE(nodata <= 0xffff, HSR_ERR_INVALID, "hsr_srf_integrate_moments_batched: nodata=%d is not a uint16 value (negative = none)", nodata);
  HSR_REQUIRE(out_row >= nb && (out_row & 3) == 0 && out_row <= HSR_MAX_BANDS, HSR_ERR_UNSUPPORTED,
              "hsr_srf_integrate_moments_batched: out_row=%d must be a multiple of 4 in [nb,%d]", out_row, HSR_MAX_BANDS);
  HSR_REQUIRE(deg == 0 || real_row >= nb, HSR_ERR_INVALID, "hsr_srf_integrate_moments_batched: real_row=%d < nb", real_row);
  hsr::SrfTuning tn;
  int rc = hsr::srf_tuning(opts, &tn, "hsr_srf_integrate_moments_batched");
  if (rc != HSR_OK) return rc;
  hsr::SrfArgs a{};
  a.units = units_dev;
  a.nunits = (int32_t)info->nunits;
  a.u16 = cube_dtype == 2;
  a.u16_fast = a.u16 && tn.u16_fast;
  a.scale = scale;
  a.nodata = nodata < 0 ? 0x10000u : (uint32_t)nodata;
  a.B = B;
  a.wn = wn_dev;
  a.nb = nb;
  a.out_bs = 1;
  a.out_ps = out_row;
  a.real_bs = 1;
  a.real_ps = real_row;
  a.min_x = min_x;
  a.min_y = min_y;
  rc = hsr::srf_prepare_bands(a, k0, klen);
  if (rc != HSR_OK) return rc;
  HSR_REQUIRE(a.wtaps > 0, HSR_ERR_UNSUPPORTED,
              "hsr_srf_integrate_moments_batched: the SRF supports do not fit the LDS weight area (B=%d)", B);
  const int64_t cap = (int64_t)(256 - tn.reserved_cus) * 2;
  const int grid = (int)(info->nunits < cap ? info->nunits : cap);
  const bool fast = info->aligned16 && (a.u16 || (B & 1));
  return hsr::dispatch_batch(a, deg, fast, tn.u16_ring, grid, (hipStream_t)stream);
}
