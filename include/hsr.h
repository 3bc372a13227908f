/* hsr.h - C ABI of libhsr_mi355x.so: the EMIT -> Sentinel-2 spectral-fusion hot path on MI355X (gfx950).
 *
 * The reference (martasumyk/hyperspectral_super-resolution) is pure Python/NumPy and has no FFI
 * layer; its boundary for this path is the plain function API of `s2_emit` (s2_emit/__init__.py:1-24,
 * s2_emit/poly_regression.py:16-84).  Each entry point below names the reference code it replaces.
 * The Python package `s2_emit` of this repository binds these symbols with ctypes (see
 * INTEGRATION.md for the stub a maintainer of the reference would add).
 *
 * Conventions
 *   - plain C: pointers + sizes, no C++/torch types.  `*_dev` pointers are device (HBM) addresses
 *     owned by the caller; the library never allocates or frees on a launch path, never
 *     synchronises the device, and is safe to capture in a hipGraph.
 *   - every call is stream ordered on `stream` (a hipStream_t passed as void*; NULL = default).
 *   - return value: HSR_OK or an error code; hsr_last_error() gives the thread-local message.
 *   - images use (band_stride, pixel_stride) addressing, see below; both layouts are accepted everywhere.
 *   - moments of band b for degree d: M = 3d+2 doubles, [S_0..S_2d | T_0..T_d] with
 *     S_k = sum x^k, T_j = sum x^j y over the valid pixels (S_0 = count).
 */
#ifndef HSR_H_
#define HSR_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HSR_ABI_VERSION 5

#define HSR_OK 0
#define HSR_ERR_INVALID 1      /* bad argument (shape, alignment, NULL)            */
#define HSR_ERR_UNSUPPORTED 2  /* outside what the kernels are built for           */
#define HSR_ERR_HIP 3          /* HIP runtime / launch failure                     */

#define HSR_MAX_BANDS 16       /* output bands per call (S2 has 13)                */
#define HSR_MAX_DEG 4          /* polynomial degree of a FIT (reference uses 1, 2, 4) */
#define HSR_MAX_APPLY_DEG 8    /* polynomial degree K3 can evaluate                  */
#define HSR_MAX_SPECTRAL 560   /* input bands B per pixel (EMIT: 285)              */
#define HSR_TILE_PIXELS 64     /* pixels staged per LDS tile                       */
#define HSR_MAX_PARTIALS 4096  /* upper bound of per-launch partial-sum slots      */

/* Image-like tensors (pseudo-S2, real S2, matched output) are addressed with two strides, in elements:
 *   element (band b, pixel p) lives at base[b * band_stride + p * pixel_stride]
 *   band-major planes (C, N):          band_stride = plane stride (>= N), pixel_stride = 1
 *   pixel-major / band-last (N, C):    band_stride = 1, pixel_stride = row stride (>= C)
 * Pixel-major is the layout of the cube itself and of the reference's (H, W, C) images, and the fast
 * one on the fused path: K1 then writes one contiguous slab per tile instead of C scattered segments. */

typedef void* hsr_stream_t;

/* Per-call tuning of the K1 launches (NULL = defaults).  There is no process-wide tuning state: two plans in one
 * process cannot interfere, and a call's partial-slot layout is a function of (npix, *opts) only. */
typedef struct hsr_srf_options {
  int32_t tile_pixels;        /* pixels per LDS group of K1: 0 = default, or 64 (512-thread workgroups, 2 per CU) - the
                                 only geometry left; the field stays for ABI stability */
  int32_t reserved_cus;       /* CUs left without a persistent K1 workgroup (default 0, at most 128): lets the small
                                 kernels of the previous tile's fit (slot reduction, RCCL exchange, solve) run on another
                                 stream while K1 streams the next tile. */
  int32_t u16_single_buffer;  /* uint16 cubes: 0 = double-buffered kernel where it fits (default), 1 = single buffer */
  int32_t flags;              /* HSR_SRF_* bits, 0 = default */
} hsr_srf_options;
/* uint16 cubes only: fast arithmetic - the decode scale is folded into the SRF weights (one rounding moved) and even /
 * odd taps accumulate separately (v_pk_fma_f32).  The planes are then no longer bit-identical to hsr_tile_decode_u16
 * followed by the float32 kernel: relative difference <= 1e-6 (7e-7 observed), <= 2e-6 against the float64 reference. */
#define HSR_SRF_U16_FAST 1

/* ---- library ------------------------------------------------------------------------------ */
int hsr_abi_version(void);
const char* hsr_last_error(void);
/* Number of moments per band for a degree: 3*deg + 2. */
int hsr_moment_count(int32_t deg);
/* Partial-sum slots a K1 launch over a tile of `npix` pixels uses.  With G = ceil(npix / 64) pixel groups: G for
 * G <= 64; max(64, ceil(G / 4)) for 65 <= G <= 512 (small tiles: coarser units keep a batch's partial traffic down);
 * min(G, resident workgroups) above.  A function of npix and *opts (tile_pixels, reserved_cus) only - never of the
 * device state - so the summation tree, and with it every bit of the fitted coefficients, is reproducible.  Group g
 * of the tile accumulates into slot g % slots, in increasing g.  (hsr_poly_moments / _f64 use the default geometry: min(ceil(npix/64), 512).) */
int hsr_partial_slots(int64_t npix, const hsr_srf_options* opts);
/* Bytes of the partials workspace for (nb, deg): nb * (3deg+2) * HSR_MAX_PARTIALS doubles. */
size_t hsr_partials_bytes(int32_t nb, int32_t deg);

/* ---- K1: SRF band integration ------------------------------------------------------------
 * Replaces the hot loop of pseudo_s2_srf_integral (s2_emit/synth.py:32-43): for every pixel and
 * every supported band b,  planes[b][p] = sum_k cube[p][k] * wn[b][k]  with IEEE semantics of the
 * dense product (a non-finite sample poisons every band whose weight there is zero, synth.py:41).
 *   cube_dev   (npix, B) float32, pixel-major / band-last, the in-memory layout of
 *              load_emit_envi_rfl (s2_emit/emit_io.py:13); 4-byte aligned (16-byte = fast path)
 *   wn_dev     (nb, B) float32 dense normalised trapezoid weights (host builds them in float64
 *              from np.interp exactly as synth.py:33-35,42-43 and rounds once)
 *   k0, klen   host arrays [nb]: support [k0, k0+klen) of each row of wn (zeros outside)
 *   out_dev    float32 out, element (b, p) at out_dev[b * out_bs + p * out_ps].  A pixel-major output whose rows
 *              are a multiple of 4 floats (<= 16) and 16-byte aligned is owned by the callee as whole rows: the
 *              pad columns nb..row-1 are written too (with zeros).
 */
int hsr_srf_integrate(const float* cube_dev, int64_t npix, int32_t B,
                      const float* wn_dev, const int32_t* k0, const int32_t* klen, int32_t nb,
                      float* out_dev, int64_t out_bs, int64_t out_ps, const hsr_srf_options* opts,
                      hsr_stream_t stream);

/* ---- K1+K2 fused: SRF integration and Vandermonde moments in one pass over the cube --------
 * K1 as above and, in the same pass, the per-band power sums that np.polyfit's normal equations
 * need (s2_emit/poly_regression.py:59-60; all-valid-pixel flavour of
 * calibrate_pseudo_to_real_linear, Pairs_EMIT_S2_demo-2.ipynb cell 72 raw lines 4484-4510):
 * pixel p counts for band b iff mask[p] (if given) && finite(x) && finite(y) && x > min_x && y > min_y,
 * with x = out[b][p] (float32) and y = real_dev[b * real_bs + p * real_ps]; sums are float64.
 *   partials_dev  workspace of hsr_partials_bytes(nb, deg); layout [slot][nb][3deg+2] (a slot is whole cache lines)
 *   returns the slot count used in *slots_out (= hsr_partial_slots(npix, opts)).
 */
int hsr_srf_integrate_moments(const float* cube_dev, int64_t npix, int32_t B,
                              const float* wn_dev, const int32_t* k0, const int32_t* klen, int32_t nb,
                              float* out_dev, int64_t out_bs, int64_t out_ps,
                              const float* real_dev, int64_t real_bs, int64_t real_ps, const uint8_t* mask_dev,
                              float min_x, float min_y, int32_t deg,
                              double* partials_dev, int32_t* slots_out, const hsr_srf_options* opts,
                              hsr_stream_t stream);

/* ---- K1+K2 with K3 of an OLDER tile as a pre-phase of the same launch (ABI 4) -----------------------------
 * hsr_srf_integrate_moments, and before a workgroup starts its groups it applies its slice of `job`: matched = polynomial(x)
 * over job->npix pixels of pixel-major rows with THIS launch's row length (out_ps) and band count, degree = deg
 * (hsr_poly_apply semantics and bits: float64 Horner, mask select, clip, channels >= nb pass through).  For the pipelined
 * order, where K3 of tile i-2 only needs coefficients that have been ready for a whole K1: one launch per tile on the
 * caller's stream.  Float32 cubes, 16-byte aligned rows of 4 / 8 / 12 / 16 floats; HSR_ERR_UNSUPPORTED otherwise.
 * job == NULL: plain hsr_srf_integrate_moments. */
typedef struct hsr_apply_job {
  const float* x_dev;            /* pseudo image of the older tile (npix, out_ps)                                      */
  float* out_dev;                /* its matched image                                                                  */
  const double* coeffs_dev;      /* (nb, deg+1)                                                                        */
  const uint8_t* mask_dev;       /* polynomial only where != 0; NULL: everywhere                                       */
  int64_t npix;
  int32_t clip;
  int32_t fit_slots;             /* tail fit (optional, fit_partials_dev != NULL): the first nb workgroups to finish their groups   */
  const double* fit_partials_dev;/* reduce + solve one band each of the PREVIOUS launch's partial slots [fit_slots][nb][3deg+2]   */
  double* fit_moments_dev;       /* -> (nb, 3deg+2) and (nb, deg+1), bit-identical to hsr_moments_reduce_solve; no launch of its  */
  double* fit_coeffs_dev;        /* own, no side stream.  x_dev may be NULL when only the tail fit rides.                           */
  int64_t fit_min_count;
  unsigned int* fit_counter_dev; /* running ticket counter (device, zero before the first launch that uses it)                      */
  unsigned int fit_ticket_base;  /* = number of workgroups of all earlier launches that used the counter (mod 2^32).  The launch   */
                                 /* must have >= nb workgroups (hsr_partial_slots(npix, opts) >= nb), else HSR_ERR_UNSUPPORTED         */
  int32_t reserved;
  /* ABI 5 - exchange pipelines (hsr_pipeline_create_exchange): the fit crosses to another queue of the same GPU between the slot   */
  /* reduction and the solve.  All three may be NULL (= ABI 4 behaviour).                                                           */
  unsigned int* fit_ready_dev;   /* non-NULL: the tail fit stops at the moments (written through to memory, no solve) and adds 1 to   */
                                 /* this word per finished band: nb additions = the tile's moments are complete                     */
  const unsigned int* coeffs_ready_dev; /* non-NULL: coeffs_dev is written by a kernel of another queue; the pre-phase polls this   */
  unsigned int coeffs_ready_value;      /* word until it has reached this value (wrap-safe >=), then reads the coefficients through */
  unsigned int reserved2;
  unsigned int* sync_error_dev;  /* optional: receives a non-zero code if a poll runs into its wall-clock limit (20 s)              */
  /* ABI 5 - ONE fit over a group of T tiles (a mosaic held by one GPU: BASELINE configs[3]/[4] per-GPU work; the reference's tiles,  */
  /* tiles_helpers/utils.py:223-305, fitted together).  fit_group_tiles >= 2: fit_moments_dev must be entry fit_group_index of       */
  /* fit_group_moments_dev [T][nb][3deg+2]; the tail writes that entry and - for the group's LAST tile (index T-1) only - adds the T   */
  /* entries in the fixed order of hsr_moments_reduce over T slots into fit_group_total_dev (nb, 3deg+2) and solves into              */
  /* fit_coeffs_dev (nb, deg+1).  For the other tiles no coefficients are written.  T <= 64; not together with fit_ready_dev.         */
  int32_t fit_group_tiles;
  int32_t fit_group_index;
  const double* fit_group_moments_dev;
  double* fit_group_total_dev;
} hsr_apply_job;
int hsr_srf_integrate_moments_apply(const float* cube_dev, int64_t npix, int32_t B,
                                    const float* wn_dev, const int32_t* k0, const int32_t* klen, int32_t nb,
                                    float* out_dev, int64_t out_bs, int64_t out_ps,
                                    const float* real_dev, int64_t real_bs, int64_t real_ps, const uint8_t* mask_dev,
                                    float min_x, float min_y, int32_t deg,
                                    double* partials_dev, int32_t* slots_out, const hsr_srf_options* opts,
                                    const hsr_apply_job* job, hsr_stream_t stream);
/* The same for a cube in the uint16 tile format (hsr_srf_integrate_moments_u16's arguments + the job).  Only the ring kernel
 * carries a job: HSR_ERR_UNSUPPORTED unless the cube is 16-byte aligned, 48 <= B <= ~300, the weights fit LDS and the rows
 * are pixel-major. */
/* HSR_OK if a K1 launch of this geometry can carry an apply job / tail fit (the conditions of the two entry points below except the
 * cube pointer's alignment), HSR_ERR_UNSUPPORTED with the reason otherwise.  Host only; hsr_pipeline_create_fused / _exchange call it. */
int hsr_srf_fused_launch_supported(int32_t cube_dtype, int32_t B, int32_t nb, const int32_t* k0, const int32_t* klen,
                                   int64_t out_ps, int32_t deg, const hsr_srf_options* opts);
int hsr_srf_integrate_moments_u16_apply(const uint16_t* cube_dev, int64_t npix, int32_t B, float scale, int32_t nodata,
                                        const float* wn_dev, const int32_t* k0, const int32_t* klen, int32_t nb,
                                        float* out_dev, int64_t out_bs, int64_t out_ps,
                                        const float* real_dev, int64_t real_bs, int64_t real_ps, const uint8_t* mask_dev,
                                        float min_x, float min_y, int32_t deg,
                                        double* partials_dev, int32_t* slots_out, const hsr_srf_options* opts,
                                        const hsr_apply_job* job, hsr_stream_t stream);

/* ---- K1+K2+fit in ONE launch (single tile) ----------------------------------------------------
 * hsr_srf_integrate_moments followed by hsr_moments_reduce_solve without the second launch: every workgroup draws a
 * ticket when its slot of partials is written; the workgroup that completes a group of slots (slot mod 64) adds the
 * group, the one that completes the last group adds the groups, writes the moments, solves the bands and re-arms the
 * tickets.  The summation tree is the one of hsr_moments_reduce (lane l adds slots l, l+64, ...; butterfly over the
 * lanes), so moments and coefficients carry the same bits as the two-launch form.
 *   group_partials_dev  64 * nb * (3deg+2) doubles of scratch
 *   tickets_dev         65 int32, ZERO before the first launch; the kernel leaves them zero
 *   moments_dev         (nb, 3deg+2) float64 out;   coeffs_dev (nb, deg+1) float64 out, highest power first
 *   min_count           as hsr_poly_solve */
typedef struct hsr_fused_fit {
  double* group_partials_dev;
  int32_t* tickets_dev;
  double* moments_dev;
  double* coeffs_dev;
  int64_t min_count;
} hsr_fused_fit;
#define HSR_FIT_GROUPS 64
#define HSR_FIT_TICKETS 65

int hsr_srf_integrate_fit(const float* cube_dev, int64_t npix, int32_t B,
                          const float* wn_dev, const int32_t* k0, const int32_t* klen, int32_t nb,
                          float* out_dev, int64_t out_bs, int64_t out_ps,
                          const float* real_dev, int64_t real_bs, int64_t real_ps,
                          const uint8_t* mask_dev, float min_x, float min_y, int32_t deg,
                          double* partials_dev, int32_t* slots_out, const hsr_fused_fit* fit,
                          const hsr_srf_options* opts, hsr_stream_t stream);


/* ---- K2: moments of already materialised planes -----------------------------------------------
 * Same sums as above for x/y planes that exist already (after a percentile stretch, or for the
 * OT-sampled columns of fit_ot_poly_rgb, poly_regression.py:59-60).  If lohi_x_dev / lohi_y_dev
 * (nb x 2 doubles: lo, hi) is given, the value is first stretched as color.py:33 does:
 * float32(clip((v - lo) / (hi - lo + 1e-12), 0, 1)) and validity is tested on the raw value.
 */
int hsr_poly_moments(const float* x_dev, int64_t x_bs, int64_t x_ps, const float* y_dev, int64_t y_bs, int64_t y_ps,
                     const uint8_t* mask_dev, int64_t npix, int32_t nb, int32_t deg,
                     float min_x, float min_y, const double* lohi_x_dev, const double* lohi_y_dev,
                     double* partials_dev, int32_t* slots_out, hsr_stream_t stream);

/* Same sums for float64 sample columns (the OT-sampled X / barycentric Ybar columns of
 * fit_ot_poly_rgb, poly_regression.py:46-60, which the reference keeps in float64). */
int hsr_poly_moments_f64(const double* x_dev, int64_t x_stride, const double* y_dev, int64_t y_stride,
                         int64_t npix, int32_t nb, int32_t deg,
                         double* partials_dev, int32_t* slots_out, hsr_stream_t stream);

/* Fixed-order reduction of the partial slots -> moments_dev (nb, 3deg+2) doubles. */
int hsr_moments_reduce(const double* partials_dev, int32_t slots, int32_t nb, int32_t deg,
                       double* moments_dev, hsr_stream_t stream);

/* ---- polynomial solve ------------------------------------------------------------------------
 * np.polyfit semantics (column-scaled least squares, rcond = count*eps, minimum-norm on rank loss)
 * from the moments: scaled normal equations, symmetric Jacobi eigen-solve, float64.
 * count < min_count -> identity polynomial (coeffs[-2] = 1), the reference's fallback
 * (poly_regression.py:38-41 with 200; notebook cell 72 with 50).
 * coeffs (nb, deg+1) doubles, highest power first (np.polyfit order).
 */
int hsr_poly_solve(const double* moments_dev, int32_t nb, int32_t deg, int64_t min_count,
                   double* coeffs_dev, hsr_stream_t stream);
/* hsr_moments_reduce + hsr_poly_solve in one launch (single-GPU fast path: no exchange between them);
 * bit-identical to the two separate calls. */
int hsr_moments_reduce_solve(const double* partials_dev, int32_t slots, int32_t nb, int32_t deg,
                             int64_t min_count, double* moments_dev, double* coeffs_dev,
                             hsr_stream_t stream);
/* Host twin (same code compiled for the CPU) for callers that hold the moments on the host. */
int hsr_poly_solve_host(const double* moments, int32_t nb, int32_t deg, int64_t min_count,
                        double* coeffs);

/* ---- K3: polynomial apply ----------------------------------------------------------------------
 * apply_poly_rgb (s2_emit/poly_regression.py:65-84): out = float32(x); where mask (or everywhere
 * if mask_dev is NULL) out = float32(polyval_f64(coeffs[c], x)); then clip to [0,1] if `clip`
 * (NaN stays NaN).  Optional stretch first (lohi_dev: nb x 2 doubles, color.py:33).
 * coeffs_dev == NULL: no polynomial (stretch and/or clip only = apply_shared_percentile_stretch).
 * deg in [0, HSR_MAX_APPLY_DEG].
 */
int hsr_poly_apply(const float* x_dev, int64_t x_bs, int64_t x_ps, const uint8_t* mask_dev,
                   const double* coeffs_dev, int32_t nb, int32_t deg, int64_t npix,
                   const double* lohi_dev, int32_t clip,
                   float* out_dev, int64_t out_bs, int64_t out_ps, hsr_stream_t stream);

/* ---- a4: exact masked percentiles (np.percentile, linear interpolation) ------------------------
 * color.py:31-32: per channel the (pmin, pmax) percentiles of the masked values, exact order
 * statistics + NumPy's lerp, in float64.  3-pass radix select on the float32 keys.
 *   work_dev: workspace of hsr_percentile_work_bytes(nb) bytes;  lohi_dev: (nb, 2) doubles out.
 */
size_t hsr_percentile_work_bytes(int32_t nb);
int hsr_percentile_limits(const float* x_dev, int64_t x_bs, int64_t x_ps,
                          const uint8_t* mask_dev, int64_t npix, int32_t nb,
                          double pmin, double pmax, void* work_dev, double* lohi_dev,
                          hsr_stream_t stream);

/* The same select, pass by pass (1..3), for multi-GPU global limits: every rank runs hist(pass) on its own
 * samples, the ranks all-reduce(sum) the uint32 histogram region of that pass (hsr_percentile_hist_region gives
 * its byte offset and length inside the workspace), then every rank runs scan(pass): the order statistics
 * are then exact over the union of all ranks' masked samples (total count < 2^31).  begin() zeroes the
 * workspace; after scan(3) lohi_dev holds the limits.  hsr_percentile_limits == begin + 3 x (hist, scan). */
int hsr_percentile_begin(void* work_dev, int32_t nb, hsr_stream_t stream);
int hsr_percentile_hist_region(int32_t pass, int32_t nb, int64_t* offset_bytes, int64_t* count_u32);
int hsr_percentile_hist(int32_t pass, const float* x_dev, int64_t x_bs, int64_t x_ps, const uint8_t* mask_dev,
                        int64_t npix, int32_t nb, void* work_dev, hsr_stream_t stream);
int hsr_percentile_scan(int32_t pass, int32_t nb, double pmin, double pmax, void* work_dev, double* lohi_dev,
                        hsr_stream_t stream);

/* ---- uint16 tiles -------------------------------------------------------------------------
 * The training tile pairs are stored as uint16 reflectance x emit_scale (10000) with 65535 as nodata
 * (reference writer: tiles_helpers/utils.py:309-318 defaults, :362-374 arithmetic).
 *
 * hsr_tile_encode_u16 replaces utils.py:362-374 on n float32 samples (any layout, elementwise):
 *   valid = isfinite(x) && !(has_src_nodata && x == src_nodata);
 *   out   = valid ? clip(int32(rint(x * scale)), 0, nodata_u16 - 1) : nodata_u16          [bit-exact]
 * hsr_tile_decode_u16 is the consumers' convention (Pairs_EMIT_S2_demo-2.ipynb cell 65, `out *= float(scale)`
 * on float32): out = u == nodata ? NaN : float32(u) * scale; nodata < 0: no nodata value.
 * hsr_srf_integrate[_moments]_u16: K1 (+K2) directly on an (npix, B) uint16 cube - the LDS-DMA moves the
 * 2-byte samples (half the HBM bytes of the float32 cube) and the decode happens on the way out of LDS.
 * Planes and partial moments are bit-identical to hsr_tile_decode_u16 followed by the float32 entry points;
 * a pixel with a nodata sample is NaN in every band (0 * NaN), exactly as the two-step path.  Other
 * arguments as hsr_srf_integrate / hsr_srf_integrate_moments. */
int hsr_tile_encode_u16(const float* x_dev, int64_t n, float scale, int32_t has_src_nodata, float src_nodata,
                        int32_t nodata_u16, uint16_t* out_dev, hsr_stream_t stream);
int hsr_tile_decode_u16(const uint16_t* u_dev, int64_t n, float scale, int32_t nodata, float* out_dev,
                        hsr_stream_t stream);
int hsr_srf_integrate_u16(const uint16_t* cube_dev, int64_t npix, int32_t B, float scale, int32_t nodata,
                          const float* wn_dev, const int32_t* k0, const int32_t* klen, int32_t nb, float* out_dev,
                          int64_t out_bs, int64_t out_ps, const hsr_srf_options* opts, hsr_stream_t stream);
int hsr_srf_integrate_moments_u16(const uint16_t* cube_dev, int64_t npix, int32_t B, float scale, int32_t nodata,
                                  const float* wn_dev, const int32_t* k0, const int32_t* klen, int32_t nb,
                                  float* out_dev, int64_t out_bs, int64_t out_ps, const float* real_dev,
                                  int64_t real_bs, int64_t real_ps, const uint8_t* mask_dev, float min_x,
                                  float min_y, int32_t deg, double* partials_dev, int32_t* slots_out,
                                  const hsr_srf_options* opts, hsr_stream_t stream);
int hsr_srf_integrate_fit_u16(const uint16_t* cube_dev, int64_t npix, int32_t B, float scale, int32_t nodata,
                              const float* wn_dev, const int32_t* k0, const int32_t* klen, int32_t nb,
                              float* out_dev, int64_t out_bs, int64_t out_ps,
                              const float* real_dev, int64_t real_bs, int64_t real_ps,
                              const uint8_t* mask_dev, float min_x, float min_y, int32_t deg,
                              double* partials_dev, int32_t* slots_out, const hsr_fused_fit* fit,
                              const hsr_srf_options* opts, hsr_stream_t stream);


/* ---- batched small tiles (the reference's actual problem size) -----------------------------------------------
 * The authors run this pipeline on 100 x 100 EMIT <-> 600 x 600 S2 tile pairs, one independent fit per tile
 * (tiles_helpers/utils.py:223-305 find_valid_paired_tiles(emit_tile_size=100, scale=6);
 * legacy_notebooks/Spectral_matching.ipynb raw lines 293-294 "EMIT: (285, 100, 100)").  A 100 x 100 x 285 tile is
 * 11.4 MB = 2 us of HBM time: launched one by one the step is launch-bound.  These entry points run T tiles of
 * arbitrary npix_t in ONE launch per stage with per-tile partial slots, per-tile solve and per-tile coefficients
 * ("independent coefficients per tile" = coeff_sync "local").  Results are bit-identical to T single-tile calls:
 * a tile's slots, the group -> slot map and every summation order depend on that tile's npix (and *opts) only.
 *
 *   1. fill hsr_batch_tile[T] on the host (pointers, npix);  hsr_batch_plan() -> per-tile slot ranges and the work
 *      units (one per (tile, slot)); call it with units_out = NULL first to size the buffers (info->nunits);
 *   2. copy tiles and units to the device (plain bytes; they stay valid as long as the pointers do);
 *   3. per batch: hsr_srf_integrate_moments_batched -> hsr_moments_reduce_solve_batched -> hsr_poly_apply_batched.
 * Images of a batch are pixel-major rows (band_stride 1): pseudo / matched rows of `out_row` floats, real-S2 rows
 * of `real_row` floats. */
typedef struct hsr_batch_tile {       /* 64 bytes */
  const void* cube_dev;               /* (npix, B) float32 - or uint16 tile samples - pixel-major                */
  const float* real_dev;              /* real-S2 target rows of the tile (NULL if deg == 0)                      */
  const uint8_t* mask_dev;            /* npix bytes, 1 = use; or NULL                                            */
  float* pseudo_dev;                  /* K1 output rows                                                          */
  float* matched_dev;                 /* K3 output rows (may equal pseudo_dev: in place)                         */
  int64_t npix;                       /* > 0                                                                     */
  int64_t slot0;                      /* [plan] first partial slot of the tile in the batch workspace            */
  int32_t slots;                      /* [plan] = hsr_partial_slots(npix, opts)                                  */
  int32_t ngroups;                    /* [plan] 64-pixel groups of the tile                                      */
} hsr_batch_tile;
typedef struct hsr_batch_unit {       /* 64 bytes, written by hsr_batch_plan, read by the K1 kernels only          */
  const void* cube_dev;
  const float* real_dev;
  const uint8_t* mask_dev;
  float* pseudo_dev;
  double* part_dev;                   /* partials of (tile, slot): nb * M doubles, (band b, moment m) at [b*M+m]   */
  int64_t npix;
  int32_t slots, slot, ngroups, reserved;
} hsr_batch_unit;
typedef struct hsr_batch_info {
  int64_t nunits;                     /* work units = total partial slots of the batch                           */
  int64_t total_pixels;
  int64_t max_npix;
  int32_t ntiles;
  int32_t aligned16;                  /* every cube base is 16-byte aligned (LDS-DMA path for whole groups)      */
} hsr_batch_info;
/* Bytes of the partials workspace of a batch: nunits * nb * (3deg+2) doubles. */
size_t hsr_batch_partials_bytes(int64_t nunits, int32_t nb, int32_t deg);
/* Host-side, no GPU call.  Fills slot0 / slots / ngroups of every tile and *info; if units_out != NULL (capacity
 * units_capacity >= info->nunits) also the unit table, whose part_dev pointers point into partials_dev. */
int hsr_batch_plan(hsr_batch_tile* tiles, int32_t ntiles, int32_t nb, int32_t deg, double* partials_dev,
                   const hsr_srf_options* opts, hsr_batch_unit* units_out, int64_t units_capacity,
                   hsr_batch_info* info);
/* K1 (+K2 if deg >= 1) over the whole batch in one launch.  cube_dtype 0: float32, 2: uint16 tiles (scale, nodata
 * as hsr_srf_integrate_u16). */
int hsr_srf_integrate_moments_batched(const hsr_batch_unit* units_dev, const hsr_batch_info* info, int32_t cube_dtype,
                                      float scale, int32_t nodata, int32_t B, const float* wn_dev, const int32_t* k0,
                                      const int32_t* klen, int32_t nb, int32_t out_row, int32_t real_row, float min_x,
                                      float min_y, int32_t deg, const hsr_srf_options* opts, hsr_stream_t stream);
/* Per tile: fixed-order slot reduction + np.polyfit solve (same trees as hsr_moments_reduce_solve).
 * moments_dev (T, nb, 3deg+2), coeffs_dev (T, nb, deg+1) doubles. */
int hsr_moments_reduce_solve_batched(const hsr_batch_tile* tiles_dev, int32_t ntiles, const double* partials_dev,
                                     int32_t nb, int32_t deg, int64_t min_count, double* moments_dev,
                                     double* coeffs_dev, hsr_stream_t stream);
/* K3 over the whole batch: matched = apply(pseudo) with the tile's own coefficients (hsr_poly_apply semantics).
 * use_mask is a set of flags: 1 = polynomial only where the tile's mask is set; 2 = coeffs_dev holds ONE (nb, deg+1) set
 * that every tile uses (a mosaic with a global fit) instead of (T, nb, deg+1). */
int hsr_poly_apply_batched(const hsr_batch_tile* tiles_dev, int32_t ntiles, int64_t max_npix, const double* coeffs_dev,
                           int32_t nb, int32_t deg, int32_t row, int32_t use_mask, int32_t clip, hsr_stream_t stream);

/* ---- ENVI interleaves -> pixel-major -------------------------------------------------------
 * load_emit_envi_rfl (s2_emit/emit_io.py:7-16) hands the path an (H, W, B) array whatever the file's interleave; files
 * written by gdalwarp are BIL or BSQ.  The raw file bytes go to the GPU as they are and are transposed there:
 * interleave 1 = BIL (lines, bands, samples), 2 = BSQ (bands, lines, samples) -> out (lines, samples, bands).
 * dtypes: 0 float32, 2 uint16, 3 int16 (in only); out_dtype 0 or 2 (uint16 only from uint16).  HBM-bound, 2 x bytes. */
int hsr_interleave_to_bip(const void* in_dev, int32_t in_dtype, int32_t interleave, int64_t lines, int64_t samples,
                          int64_t bands, void* out_dev, int32_t out_dtype, hsr_stream_t stream);

/* ---- entropic OT targets ------------------------------------------------------------------
 * Replaces, for uniform marginals, the POT calls of s2_emit/poly_regression.py:49-56 and color.py:97-104:
 *   M = ot.dist(X, Y, "sqeuclidean"); P = ot.sinkhorn(a, b, M, reg, numItermax=, stopThr=);
 *   Ybar = (P @ Y) / (P.sum(axis=1, keepdims=True) + 1e-32)
 * x_dev (n,3), y_dev (m,3), ybar_dev (n,3): float64, C order.  Follows POT's sinkhorn_knopp schedule (breakdown
 * test every iteration -> previous (u, v); error ||v*(K^T u) - b||_2 every 10th iteration).  The whole solve is
 * enqueued without host synchronisation; iteration state lives in the workspace and is copied to info_dev
 * (optional, 24 bytes: int32 break_iter, conv_iter [0x7fffffff = never], checks, pad; float64 last error).
 * PARITY UNPINNED: POT is absent offline; validated against the oracle's restatement and OT invariants.
 * Workspace: hsr_ot_work_bytes(n, m) bytes (the n x m float64 kernel matrix dominates), 256-byte aligned. */
int64_t hsr_ot_work_bytes(int64_t n, int64_t m);
/* The same solve in stages: begin (kernel matrix, u = 1/n, v = 1/m, state reset), iterate [first_iter, first_iter +
 * count) - may be called repeatedly, copying the 24-byte state to info_dev so that a caller can stop enqueueing
 * once break_iter / conv_iter is set (one host synchronisation per look) - and finish (barycentric projection with
 * the (u, v) the loop ended on; iterations_done = total iterations enqueued).
 * hsr_ot_sinkhorn_barycentric == begin + iterate(0, numItermax) + finish, without any synchronisation. */
int hsr_ot_begin(const double* x_dev, int64_t n, const double* y_dev, int64_t m, double reg, void* work_dev,
                 hsr_stream_t stream);
int hsr_ot_iterate(int64_t n, int64_t m, int32_t first_iter, int32_t count, double stop_thr, void* work_dev,
                   int32_t* info_dev, hsr_stream_t stream);
int hsr_ot_finish(const double* y_dev, int64_t n, int64_t m, int32_t iterations_done, void* work_dev, double* ybar_dev,
                  int32_t* info_dev, hsr_stream_t stream);
int hsr_ot_sinkhorn_barycentric(const double* x_dev, int64_t n, const double* y_dev, int64_t m, double reg,
                                int32_t num_iter_max, double stop_thr, void* work_dev, double* ybar_dev,
                                int32_t* info_dev, hsr_stream_t stream);

/* Validity mask of the pipeline (poly_regression.py:106,118): mask[p] = all bands of x finite
 * && x[pos_band][p] > 0 (pos_band < 0: skip) && all bands of y finite (y_dev may be NULL),
 * optionally AND-ed with mask_in_dev. */
int hsr_valid_mask(const float* x_dev, int64_t x_bs, int64_t x_ps, int32_t nbx, int32_t pos_band,
                   const float* y_dev, int64_t y_bs, int64_t y_ps, int32_t nby,
                   const uint8_t* mask_in_dev, int64_t npix, uint8_t* mask_out_dev,
                   hsr_stream_t stream);

/* ---- K4 (variant a9): multivariate polynomial-ridge fusion ---------------------------------------
 * legacy_notebooks/Spectral_matching.ipynb: Pipeline(StandardScaler, PolynomialFeatures(3, no bias),
 * Ridge(alpha)) on logit(EMIT) (raw lines 475-490), applied by predict_cube_logit (raw lines 192-213).
 * Monomial order = sklearn's (degree-major, combinations_with_replacement); n_in <= 16, degree <= 3. */
int hsr_polyfeat_count(int32_t n_in, int32_t degree);                 /* 285 for (10, 3); -1 if unsupported */
int hsr_polyfeat_table(int32_t n_in, int32_t degree, uint8_t* idx_out /* [count][3], index n_in = constant 1 */);
/* Uploads the monomial table for (n_in, degree); NOT a launch-path call (allocates). */
int hsr_polyfeat_prepare(int32_t n_in, int32_t degree);
/* First ncols columns of the rows of P (n, ldp) float64 = [1 | monomials of (x - mean)/scale | 0 pad];
 * x element (row r, band c) at x_dev[r * x_rs + c * x_cs]; mean/scale: device doubles [n_in]. */
int hsr_polyfeat_expand_f64(const float* x_dev, int64_t x_rs, int64_t x_cs, const double* mean_dev,
                            const double* scale_dev, int64_t n, int32_t n_in, int32_t degree,
                            double* p_dev, int64_t ldp, int32_t ncols, hsr_stream_t stream);
/* C (na x nb, ldc) = A^T B over the n rows of A (n, lda) and B (n, ldb), float64 on
 * v_mfma_f64_16x16x4_f64; na, nb multiples of 16; row chunks reduced in a fixed order.
 * work_dev: hsr_gram_work_bytes(na, nb, n) bytes. */
size_t hsr_gram_work_bytes(int32_t na, int32_t nb, int64_t n);
int hsr_gram_f64(const double* a_dev, int64_t lda, int32_t na, const double* b_dev, int64_t ldb, int32_t nb,
                 int64_t n, double* work_dev, double* c_dev, int64_t ldc, hsr_stream_t stream);
/* Cholesky solve of the ridge system (Phi_c^T Phi_c + alpha I) W = Phi_c^T Y_c of the fit
 * (legacy_notebooks/Spectral_matching.ipynb raw lines 475-490: Ridge(alpha=1), solver 'cholesky' semantics):
 * a_dev (n, lda) float64 symmetric positive definite - its lower triangle is overwritten by the factor L -,
 * b_dev (n, ldb) holds nrhs right-hand sides in its columns and is overwritten by the solution.  n must be a
 * multiple of 32 in [32, 512]: pad with an identity block and zero right-hand-side rows.  *info_dev = 0, or the
 * 1-based index of the first non-positive pivot (LAPACK potrf convention); asynchronous like everything else. */
size_t hsr_chol_work_bytes(int32_t n);   /* workspace: the inverses of the 32 x 32 diagonal blocks of L */
int hsr_chol_solve_f64(double* a_dev, int64_t lda, int32_t n, double* b_dev, int64_t ldb, int32_t nrhs,
                       double* work_dev, int32_t* info_dev, hsr_stream_t stream);

/* The small steps of the fit around Gram and Cholesky (sklearn Pipeline(StandardScaler, PolynomialFeatures, Ridge),
 * Spectral_matching.ipynb raw lines 475-490), each one launch instead of a dozen tensor operations:
 * hsr_ridge_stats: stats_dev[1 + 2 n_in] = [n, mean.., M2..] of the n rows of x (float64, shifted-data sums, fixed order),
 *   mean_dev / scale_dev [n_in] = StandardScaler's mean_ and scale_ (zero variance -> 1); work: hsr_ridge_stats_work_bytes.
 * hsr_ridge_assemble: from G (na, >= na + T) = [1 | Phi]^T [1 | Phi | Y] (hsr_gram_f64) the centred ridge system
 *   A (npad, npad) = Phi_c^T Phi_c + alpha I (identity block past nf), B (npad, ldb) = Phi_c^T (Y - ybar) (zero rows past nf);
 *   *info_dev = 0 for hsr_chol_solve_f64.
 * hsr_ridge_finish: W (nf, ldw) float64 solution -> intercept (float64 and float32), W as float32 (kpad, T) with zero rows
 *   past nf, mean and 1 / scale as float32 - the operands of hsr_polyfeat_predict. */
size_t hsr_ridge_stats_work_bytes(int32_t n_in);
int hsr_ridge_stats(const float* x_dev, int64_t x_rs, int64_t x_cs, int64_t n, int32_t n_in, double* work_dev,
                    double* stats_dev, double* mean_dev, double* scale_dev, hsr_stream_t stream);
int hsr_ridge_assemble(const double* g_dev, int64_t ldg, int32_t na, int32_t nf, int32_t T, double alpha,
                       double* a_dev, int32_t npad, double* b_dev, int64_t ldb, int32_t* info_dev, hsr_stream_t stream);
int hsr_ridge_finish(const double* g_dev, int32_t na, int32_t nf, int32_t T, const double* w_dev, int64_t ldw,
                     const double* mean_dev, const double* scale_dev, int32_t n_in, int32_t kpad,
                     double* b64_dev, float* b32_dev, float* w32_dev, float* mean32_dev, float* inv32_dev,
                     hsr_stream_t stream);

/* out[t * out_stride + p] = act(sum_f W[f][t] * phi_f((x_p - mean) * inv_scale) + bias[t]) with the features
 * expanded on chip (v_mfma_f32_32x32x2_f32); W (count rounded up to even rows, ldw) float32 with zero rows past
 * count; activation 1 = sigmoid(clip(z, -50, 50)) (notebook raw lines 178-181), 0 = identity. */
int hsr_polyfeat_predict(const float* x_dev, int64_t x_ps, int64_t x_cs, const float* mean_dev,
                         const float* inv_scale_dev, int64_t npix, int32_t n_in, int32_t degree,
                         const float* w_dev, int64_t ldw, const float* bias_dev, int32_t T, int32_t activation,
                         float* out_dev, int64_t out_stride, hsr_stream_t stream);
/* The same with predict_cube_logit's rule for unusable pixels fused into the epilogue (Spectral_matching.ipynb raw
 * lines 197-203: `bad = ~isfinite(X).all(0) | isclose(X, nodata).any(0); out[:, bad] = nan`): nan_bad_pixels != 0 -> a
 * pixel with a non-finite input, or (use_nodata) an input within 1e-8 + 1e-5 |nodata| of nodata, is NaN in every target. */
int hsr_polyfeat_predict_cube(const float* x_dev, int64_t x_ps, int64_t x_cs, const float* mean_dev,
                              const float* inv_scale_dev, int64_t npix, int32_t n_in, int32_t degree,
                              const float* w_dev, int64_t ldw, const float* bias_dev, int32_t T,
                              int32_t activation, int32_t nan_bad_pixels, float nodata, int32_t use_nodata,
                              float* out_dev, int64_t out_stride, hsr_stream_t stream);

/* ---- f1: grid-aligned resamplers between the phases -----------------------------------------------
 * downsample_s2_to_grid ('average') and reproject_stack_to_grid ('bilinear') of the notebook
 * (Pairs_EMIT_S2_demo-2.ipynb cell 73, raw lines 4538-4599) for exactly aligned integer-factor grids:
 * an f x f block mean (double accumulate, float32 store, then `* scale` in float32) and a pixel-centre
 * aligned separable bilinear upsampling with edge clamp.  GDAL parity unpinned (rasterio absent).
 * in_dtype: 0 float32, 1 uint8, 2 uint16.  Fine grid is (Hc*f, Wc*f); images use (band, pixel) strides. */
int hsr_block_mean(const void* in_dev, int32_t in_dtype, int64_t in_bs, int64_t in_ps, int32_t nb,
                   int32_t Hc, int32_t Wc, int32_t factor, float scale,
                   float* out_dev, int64_t out_bs, int64_t out_ps, hsr_stream_t stream);
int hsr_bilinear_upsample(const float* in_dev, int64_t in_bs, int64_t in_ps, int32_t nb, int32_t Hc, int32_t Wc,
                          int32_t factor, float* out_dev, int64_t out_bs, int64_t out_ps, hsr_stream_t stream);
/* The upsampler as a producer for the next two steps of the reference's driver (poly_regression.py:159 ff.): besides the fine
 * image - band-last rows of 4 floats, nb <= 4, out_dev 16-byte aligned - it writes the "all bands finite" mask of every fine
 * pixel (what hsr_valid_mask gives for this image with pos_band = -1) and adds the pass-1 histogram of the masked values
 * to the percentile workspace (after hsr_percentile_begin; continue with hsr_percentile_scan(1), _hist(2), ...), i.e. the
 * fine image is not read back for either.  Bit-identical to hsr_bilinear_upsample + hsr_valid_mask + hsr_percentile_hist(1). */
int hsr_bilinear_upsample_mask_hist(const float* in_dev, int64_t in_bs, int64_t in_ps, int32_t nb, int32_t Hc, int32_t Wc,
                                    int32_t factor, float* out_dev, uint8_t* mask_out_dev, void* percentile_work_dev,
                                    hsr_stream_t stream);

/* ---- step executor (ABI 4): the hot path of one tile as PREPARED launches, and the one-tile-deep pipeline -------------
 * A plan stores every argument of K1+K2 (hsr_srf_integrate_moments[_u16]), of the slot reduction + solve
 * (hsr_moments_reduce[_solve], hsr_poly_solve) and of K3 (hsr_poly_apply) for fixed output / workspace buffers; running a
 * step is ONE call that takes only the pointers that change from tile to tile (cube, target, mask: same shapes and strides
 * as described).  Same launches as the individual entry points, same bits - what it removes is the host-side marshalling
 * (a 128 x 1024 row block of an 8-way strong-scaling run is 36 us of GPU work; three ctypes calls with 15-25 arguments,
 * two stream contexts and four event operations from Python are ~60 us).  Plans are host objects: create / destroy are not
 * launch-path calls; run calls are stream ordered and capturable like everything else.  A plan may be used by one host
 * thread at a time. */
typedef struct hsr_step_desc {
  int32_t cube_dtype;            /* 0: float32 cube, 2: uint16 tile (scale / nodata below)                          */
  int32_t B;
  int64_t npix;
  float scale;                   /* uint16 decode factor (1e-4)                                                     */
  int32_t nodata;                /* uint16 nodata code, -1: none                                                    */
  const float* wn_dev;           /* (nb, B) weights                                                                 */
  const int32_t* k0;             /* host tables, copied by hsr_step_plan_create                                     */
  const int32_t* klen;
  int32_t nb, deg;
  float* pseudo_dev;             /* K1 output image, strides out_bs / out_ps                                        */
  int64_t out_bs, out_ps;
  int64_t real_bs, real_ps;      /* strides of the target image passed to the run calls                             */
  float min_x, min_y;
  double* partials_dev;          /* hsr_partials_bytes(nb, deg)                                                     */
  double* moments_dev;           /* (nb, 3deg+2)                                                                    */
  double* coeffs_dev;            /* (nb, deg+1)                                                                     */
  int64_t min_count;
  float* matched_dev;            /* K3 output image                                                                 */
  int64_t matched_bs, matched_ps;
  int32_t apply_mask;            /* K3 applies the polynomial only where mask != 0                                  */
  int32_t clip;
  hsr_srf_options opts;
} hsr_step_desc;
typedef struct hsr_step_plan hsr_step_plan;
typedef struct hsr_pipeline hsr_pipeline;

int hsr_step_plan_create(const hsr_step_desc* desc, hsr_step_plan** plan_out);
void hsr_step_plan_destroy(hsr_step_plan* plan);
int hsr_step_plan_slots(const hsr_step_plan* plan);      /* partial slots of the last K1 launch of this plan          */
/* K1+K2 -> slot reduction + solve -> K3 on one stream (== SpectralFusion.step without an exchange) */
int hsr_step_run(hsr_step_plan* plan, const void* cube_dev, const float* real_dev, const uint8_t* mask_dev,
                 hsr_stream_t stream);
/* the pieces, for callers that put a collective between reduce and solve */
int hsr_step_run_k1(hsr_step_plan* plan, const void* cube_dev, const float* real_dev, const uint8_t* mask_dev,
                    hsr_stream_t stream);
int hsr_step_run_reduce(hsr_step_plan* plan, hsr_stream_t stream);       /* partials -> moments_dev                   */
int hsr_step_run_solve(hsr_step_plan* plan, hsr_stream_t stream);        /* moments_dev -> coeffs_dev                */
int hsr_step_run_apply(hsr_step_plan* plan, const uint8_t* mask_dev, hsr_stream_t stream);
/* One-tile-deep pipeline over two plans (two buffer sets): submit(i) enqueues K1(i) and K3(i-1) on the caller's stream and
 * the fit of tile i on `side_stream`, which waits for K1(i) through an event recorded behind K3(i-1):
 *     caller's stream :  K1(0)  K1(1)  K3(0)  K1(2)  K3(1) ...        side stream :  fit(0)  fit(1) ...  (fit(i) under K1(i+1))
 * exchange = 0: the fit is hsr_moments_reduce_solve, enqueued by submit.  exchange = 1: when submit returns the side stream
 * already waits for K1(i); the caller enqueues reduce -> collective -> solve on it (hsr_step_run_reduce / _solve) and then
 * calls hsr_pipeline_fit_done.  *finished_slot: 0 / 1 = the slot whose tile was finished (its matched image is enqueued),
 * -1 = none.  side_stream must be a real stream (not NULL); a high-priority one gets its own hardware queue. */
int hsr_pipeline_create(hsr_step_plan* slot0, hsr_step_plan* slot1, hsr_stream_t side_stream, int32_t exchange,
                        hsr_pipeline** pipeline_out);
/* Fused form over THREE plans: K3 of tile i-2 rides in the launch of K1 of tile i (hsr_srf_integrate_moments[_u16]_apply) - one kernel
 * per tile on the caller's stream.  submit(i) finishes tile i-2 (prev_mask_dev = ITS mask); hsr_pipeline_flush finishes the
 * OLDEST unfinished tile per call (call it until *finished_slot == -1).  HSR_ERR_UNSUPPORTED unless all plans describe the same
 * geometry and cube type (float32 or uint16) with 16-byte aligned pixel-major rows of 4 / 8 / 12 / 16 floats. */
int hsr_pipeline_create_fused(hsr_step_plan* slot0, hsr_step_plan* slot1, hsr_step_plan* slot2, hsr_stream_t side_stream,
                              int32_t exchange, hsr_pipeline** pipeline_out);
void hsr_pipeline_destroy(hsr_pipeline* pipeline);
int hsr_pipeline_submit(hsr_pipeline* pipeline, const void* cube_dev, const float* real_dev, const uint8_t* mask_dev,
                        const uint8_t* prev_mask_dev, hsr_stream_t main_stream, int32_t* finished_slot,
                        void* k1_begin_event, void* k1_end_event);   /* optional hipEvent_t pair recorded around K1 (or NULL) */
int hsr_pipeline_fit_done(hsr_pipeline* pipeline);
int hsr_pipeline_flush(hsr_pipeline* pipeline, const uint8_t* mask_dev, hsr_stream_t main_stream, int32_t* finished_slot);
int64_t hsr_pipeline_count(const hsr_pipeline* pipeline);

/* ---- C1 from C: the exchange of the fit over the GPUs of one node (ABI 5) -----------------------------------------
 * One process per GPU; what crosses is the sum the reference forms per channel over the pixels it fits
 * (s2_emit/poly_regression.py:59-60) when those pixels are spread over several ranks' tiles (the reference's own unit
 * of work: tiles_helpers/utils.py:223-305): nb x (3deg+2) moment doubles per step.  RCCL's C API, bound at
 * run time (dlopen of librccl.so.1 - inside a PyTorch process the copy torch has already mapped); without RCCL these
 * return HSR_ERR_UNSUPPORTED and everything else works.
 *   hsr_comm_unique_id   rank 0 fills HSR_COMM_ID_BYTES bytes (ncclGetUniqueId) and hands them to the other ranks by any
 *                        means (a file, a socket, torch.distributed's store)
 *   hsr_comm_init        ncclCommInitRank on the CURRENT device; collective over all ranks
 *   hsr_allreduce_f64    in-place sum of `count` doubles over the ranks, stream ordered; every rank receives the same bits
 *   hsr_reduce_f64       in-place sum to `root` (other ranks' buffers are left as they were)
 *   hsr_allreduce_u32    in-place sum of uint32 (the histograms of the distributed percentile select, hsr_percentile_hist_region)
 *   hsr_bcast            `bytes` bytes from root's buffer to everybody's
 * A communicator is used by one host thread at a time; collectives must be issued in the same order on every rank.
 * In a process that also uses PyTorch, import torch BEFORE the first hsr_comm_* call: the library binds the librccl.so.1 that is
 * already mapped (torch ships its own); the other order leaves two RCCL / rocm_smi copies in the process, which ends in a
 * double free inside rocm_smi's static destructors at exit. */
#define HSR_COMM_ID_BYTES 128
typedef struct hsr_comm hsr_comm;
int hsr_comm_available(void);                      /* 1 if RCCL could be bound                                  */
int hsr_comm_version(void);                        /* ncclGetVersion code (e.g. 22606), -1 without RCCL          */
int hsr_comm_unique_id(void* id_out);
int hsr_comm_init(int32_t rank, int32_t nranks, const void* unique_id, hsr_comm** comm_out);
int hsr_comm_destroy(hsr_comm* comm);
int hsr_comm_rank(const hsr_comm* comm);
int hsr_comm_ranks(const hsr_comm* comm);
int hsr_allreduce_f64(hsr_comm* comm, double* buf_dev, int64_t count, hsr_stream_t stream);
int hsr_reduce_f64(hsr_comm* comm, double* buf_dev, int64_t count, int32_t root, hsr_stream_t stream);
int hsr_allreduce_u32(hsr_comm* comm, uint32_t* buf_dev, int64_t count, hsr_stream_t stream);
int hsr_bcast(hsr_comm* comm, void* buf_dev, int64_t bytes, int32_t root, hsr_stream_t stream);

/* ---- the fused pipeline WITH an exchange (ABI 5) -------------------------------------------------------------------
 * Four plans (four buffer sets); the caller's stream still carries ONE kernel per tile and no event or stream wait:
 *     caller's stream :  [K3(i-3) | K1+K2(i) | slot reduction of tile i-1 in the tail]       (hsr_srf_integrate_moments[_u16]_apply)
 *     side stream     :  gate(i-1) -> all-reduce of the moments of tile i-1 -> solve + publish      (under K1(i+1))
 * The tail of launch i reduces the partial slots of tile i-1 (the first nb workgroups to finish, one band each - the tree of
 * hsr_moments_reduce), writes the moments through to memory and counts the bands into a device word; `gate` is a one-wave
 * kernel on the side stream that polls that word; the collective follows it in stream order; the solve publishes the
 * coefficients by setting the tile's "ready" word, which the K3 pre-phase of launch i+2 polls (it has been set for a
 * whole K1 by then) before it reads them.  Tiles whose launch cannot carry the reduction (fewer workgroups than bands,
 * the last tiles at a drain) get it as a launch of their own; results are bit-identical to hsr_step_run with the same
 * collective between reduce and solve.  hsr_pipeline_submit(i) finishes tile i-3; hsr_pipeline_flush the oldest
 * unfinished tile per call.  The plans must have opts.reserved_cus >= 8 (HSR_ERR_INVALID otherwise): a K1 launch whose pre-phase
 * polls holds its CUs while it waits, and what it waits for - the collective's kernel, the solve - is only dispatched next to
 * it where an XCD has a free CU.  One GPU per rank: two ranks sharing a GPU starve each other's K1 the same way (big tiles).
 *   mode HSR_SYNC_ALLREDUCE : hsr_allreduce_f64 of the moments, every rank solves (identical bits everywhere)
 *   mode HSR_SYNC_BROADCAST : hsr_reduce_f64 to `root`, solve, hsr_bcast of the coefficients (the north star's wording)
 *   host_sum != NULL (comm == NULL): any other transport - the moments travel to a pinned host buffer, host_sum(user, values,
 *     count) must replace them by their sum over the ranks (called on a runtime thread in stream order; it must not call
 *     HIP), and travel back; e.g. an MPI or gloo all-reduce.  Return 0 on success. */
#define HSR_SYNC_ALLREDUCE 1
#define HSR_SYNC_BROADCAST 2
typedef int (*hsr_host_sum_fn)(void* user, double* values, int32_t count);
typedef struct hsr_exchange {
  hsr_comm* comm;
  int32_t mode;
  int32_t root;
  hsr_host_sum_fn host_sum;
  void* host_user;
  /* Rehearsal on ONE GPU only (0 in production): a one-rank RCCL all-reduce launches no kernel, so nothing stands where the
   * collective of an N-rank run would run next to K1.  rehearsal_us > 0 enqueues, behind the collective call, a stand-in kernel of
   * rehearsal_blocks workgroups (256 threads, 48 KB of LDS each - the footprint of a small RCCL kernel) that stays resident for
   * that many microseconds; it computes nothing and touches no data. */
  int32_t rehearsal_us;
  int32_t rehearsal_blocks;
} hsr_exchange;
int hsr_pipeline_create_exchange(hsr_step_plan* const* slots4, hsr_stream_t side_stream, const hsr_exchange* exchange,
                                 hsr_pipeline** pipeline_out);
/* ---- the fused pipeline with ONE fit per group of T consecutive tiles (ABI 5) -------------------------------------------
 * A mosaic held by one GPU (BASELINE.json configs[3]/[4] per-GPU work: the reference's tiles, tiles_helpers/utils.py:223-305, fitted
 * together - s2_emit/poly_regression.py:59-60 over the pixels of all of them).  group_tiles + 2 plans; launch n on the caller's stream =
 *     [K3 of tile n-T-1 | K1+K2 of tile n | slot reduction of tile n-1 in the tail (+ the group's sum and solve behind a group's last tile)]
 * and nothing else: no side stream work, no events, no CUs kept free.  hsr_pipeline_submit(n) finishes tile n-T-1 with ITS group's
 * polynomial; hsr_pipeline_flush finishes the oldest unfinished tile per call and needs whole groups (n % T == 0).
 * group_moments_dev [2][T][nb][3deg+2], group_total_dev [2][nb][3deg+2], group_coeffs_dev [2][nb][deg+1] doubles (caller-owned; index 0:
 * even groups, 1: odd groups): per-tile moments, their fixed-order sum (the tree of hsr_moments_reduce over T slots - the bits of
 * hsr_moments_reduce_solve on the T per-tile moment sets) and the group's polynomial.  2 <= group_tiles <= 64. */
int hsr_pipeline_create_group(hsr_step_plan* const* slots, int32_t nslots, int32_t group_tiles, double* group_moments_dev,
                              double* group_total_dev, double* group_coeffs_dev, hsr_stream_t side_stream,
                              hsr_pipeline** pipeline_out);
/* Synchronises `main_stream` and the side stream and reports what the device-side polls and the host callback recorded:
 * *sync_error_out = 0, or 1 (the gate of a tile's moments) / 2 (the coefficients of a K3 pre-phase) ran into the 20 s limit, + 16 = host_sum failed. */
int hsr_pipeline_status(hsr_pipeline* pipeline, hsr_stream_t main_stream, uint32_t* sync_error_out);

/* ---- diagnostics -------------------------------------------------------------------------------
 * Pure streaming read of `bytes` bytes (a multiple of 16, 16-byte aligned base): the measured HBM read ceiling of
 * the box, reported by bench.py beside the vendor peak.  mode 0: K1's own load shape - non-temporal LDS-DMA
 * (global_load_lds_dwordx4 nt), 72 KiB slabs, 512 persistent workgroups, nothing computed (a ceiling for K1);
 * mode 1: plain 16-byte global_load per lane folded into sink_dev[64] (what an ordinary streaming kernel reads);
 * mode 2: as 0 with 36 KiB slabs - the uint16 kernel's group - still 2 workgroups per CU (73 KB per CU in flight);
 * mode 3: 36 KiB slabs, 4 workgroups per CU (146 KB per CU in flight: what a deeper ring would reach). */
int hsr_probe_read(const void* buf_dev, int64_t bytes, int32_t mode, float* sink_dev, hsr_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* HSR_H_ */
