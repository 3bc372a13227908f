// Step executor: the hot path of one tile (K1+K2 -> slot reduction + solve -> K3) as PREPARED launches, and the
// one-tile-deep pipeline of SpectralFusion.submit() (fit of tile i on a side stream under K1 of tile i+1) with its HIP
// events and stream waits issued from C.
//
// Why (profiles/r03_strong_scaling.md): a rank of an 8-way strong-scaling run processes a 128 x 1024 row block per step -
// 36 us of GPU work in the pipelined order - while the Python side of submit() (argument marshalling of three ctypes
// calls with 15-25 arguments each, two torch stream contexts, four event operations) takes ~60 us: host bound.  A plan
// stores every argument once; running a step is one call with the three pointers that change from tile to tile.
// The kernels are the ones behind hsr_srf_integrate_moments / hsr_moments_reduce[_solve] / hsr_poly_solve /
// hsr_poly_apply: same launches, same bits.
#include <new>
#include <vector>

#include "hsr_common.h"
#include "hsr_solve.h"
#include "hsr_sync_dev.h"

struct hsr_step_plan {
  hsr_step_desc d;
  int32_t k0[HSR_MAX_BANDS], klen[HSR_MAX_BANDS];
  int32_t slots;
  hipEvent_t ev_k1, ev_fit;       // pipeline: K1 of this slot enqueued / fit of this slot done
  bool pending;                   // pipeline: K1 + fit enqueued, K3 not yet
  bool fitted;                    // fused pipelines: the slot reduction (+ solve, without an exchange) of this slot's tile has been enqueued
  bool exchanged;                 // four slots: gate -> collective -> solve of this slot's tile has been enqueued on the side stream
  unsigned int seq;               // four slots: sequence number of the tile in this slot (the value of its "ready" word)
};

enum { kTwoSlot = 0, kFused = 1, kExchange = 2, kGroup = 3 };
struct hsr_pipeline {
  std::vector<hsr_step_plan*> slot;
  int kind;                       // kTwoSlot: K3(i-1) as its own launch behind K1(i);  kFused (3 slots): K3(i-2) inside K1(i)'s launch;
                                  // kExchange (4 slots): K3(i-3) inside K1(i)'s launch, exchange issued from C (hsr_pipeline_create_exchange);
                                  // kGroup (T + 2 slots): one fit per group of T tiles, K3(i-T-1) inside K1(i)'s launch (hsr_pipeline_create_group)
  int nslots;
  int group_T;                    // kGroup: tiles per fit
  double* group_moments;          // kGroup: [2][T][nb][M] per-tile moments, by group parity
  double* group_total;            //         [2][nb][M]
  double* group_coeffs;           //         [2][nb][deg+1]
  hipStream_t side;
  int64_t n;                      // tiles submitted
  int exchange;                   // two slots, 1: the caller runs the fit (reduce -> collective -> solve) itself between
                                  //    hsr_pipeline_submit and hsr_pipeline_fit_done
  unsigned int* counter;          // fused: ticket counter of the tail fits (device), and what it holds
  unsigned int tickets;
  // ---- four slots: the exchange is issued from here ----
  hsr_exchange x;
  unsigned int* sync;             // device words: [0] tickets (= counter), [1] bands whose moments are published, [2] error code,
                                  //               [4 + k] "coefficients ready" word of slot k (holds the sequence number of its tile)
  unsigned int published;         // value sync[1] reaches once every reduction enqueued so far has run (nb per tile)
  int deferred;                   // slot whose solve + publish is not enqueued yet: it rides in the launch that gates the NEXT tile's
  int deferred_solve;             // collective (one side-stream launch per step instead of two), or goes out alone at a drain; -1: none
  double* host_moments[4];        // pinned staging of the host_sum transport, one per slot
  struct host_job { hsr_pipeline* pl; double* values; int32_t count; } host_jobs[4];
  volatile int host_error;        // host_sum returned non-zero
};

namespace {

// ---- kernels of the exchange pipeline (four slots) ----------------------------------------------------------------
// gate: one wave on the side stream that polls a device word until it has reached `target` - the stream-ordered work behind
// it (the collective, the solve) then starts without any event on the caller's stream.  1 wave, no LDS, a handful of
// registers: it fits on a CU the persistent K1 leaves free (hsr_srf_options.reserved_cus).
__global__ __launch_bounds__(64) void gate_kernel(const unsigned int* word, unsigned int target, unsigned int* err, unsigned int code) {
  if (threadIdx.x == 0) hsr::wait_word_at_least(word, target, err, code);
}

// adds `n` to a word at agent scope: tiles whose slot reduction ran as a launch of its own publish their moments this way
__global__ __launch_bounds__(64) void publish_add_kernel(unsigned int* word, unsigned int n) {
  if (threadIdx.x == 0) __hip_atomic_fetch_add(word, n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Rehearsal stand-in for the collective's kernel on a one-GPU box (hsr_exchange.rehearsal_us): resident for `us` microseconds with
// the footprint of a small RCCL kernel, computes nothing.
__global__ __launch_bounds__(256) void rehearsal_collective_kernel(int us, unsigned int* sink) {
  __shared__ unsigned int lds[12288];          // 48 KB
  lds[threadIdx.x] = threadIdx.x;
  __syncthreads();
  const unsigned long long t0 = hsr::sync_realtime();
  while (hsr::sync_realtime() - t0 < (unsigned long long)us * 100ull) __builtin_amdgcn_s_sleep(8);
  if (lds[(threadIdx.x * 7) & 255] == 0xffffffffu) *sink = 1u;      // never true: keeps the LDS allocation
}

// np.polyfit from the (all-reduced) moments, one thread per band - solve_kernel of hsr_poly.hip, hence its bits - and then the
// publication: coefficients written through to memory, stores waited for, the slot's "ready" word set to the tile's sequence
// number.  do_solve = 0: publish only (the coefficients came by broadcast).
// gate_word != NULL: the kernel then goes on as the NEXT tile's gate (one launch per step on the side stream instead of two).
__global__ __launch_bounds__(64) void solve_publish_kernel(const double* __restrict__ moments, int nb, int deg, long long min_count,
                                                           double* coeffs, int do_solve, unsigned int* ready, unsigned int value,
                                                           const unsigned int* gate_word, unsigned int gate_target, unsigned int* err) {
  const int b = threadIdx.x;
  if (do_solve && b < nb) {
    double c[HSR_MAX_DEG + 1];
    hsr::solve_band(moments + (size_t)b * hsr::moment_count(deg), deg, min_count, c);
    for (int j = 0; j <= deg; ++j)
      __hip_atomic_store(coeffs + (size_t)b * (deg + 1) + j, c[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    hsr::st_agent_u32(ready, value);
    if (gate_word) hsr::wait_word_at_least(gate_word, gate_target, err, 1u);
  }
}

int run_k1(hsr_step_plan* p, const void* cube, const float* real, const uint8_t* mask, hipStream_t s,
           const hsr_apply_job* job = nullptr) {
  const hsr_step_desc& d = p->d;
  if (job && d.cube_dtype == 2)
    return hsr_srf_integrate_moments_u16_apply(static_cast<const uint16_t*>(cube), d.npix, d.B, d.scale, d.nodata, d.wn_dev, p->k0,
                                               p->klen, d.nb, d.pseudo_dev, d.out_bs, d.out_ps, real, d.real_bs, d.real_ps, mask,
                                               d.min_x, d.min_y, d.deg, d.partials_dev, &p->slots, &d.opts, job, s);
  if (job)
    return hsr_srf_integrate_moments_apply(static_cast<const float*>(cube), d.npix, d.B, d.wn_dev, p->k0, p->klen, d.nb,
                                           d.pseudo_dev, d.out_bs, d.out_ps, real, d.real_bs, d.real_ps, mask, d.min_x, d.min_y,
                                           d.deg, d.partials_dev, &p->slots, &d.opts, job, s);
  if (d.cube_dtype == 2)
    return hsr_srf_integrate_moments_u16(static_cast<const uint16_t*>(cube), d.npix, d.B, d.scale, d.nodata, d.wn_dev, p->k0,
                                         p->klen, d.nb, d.pseudo_dev, d.out_bs, d.out_ps, real, d.real_bs, d.real_ps, mask,
                                         d.min_x, d.min_y, d.deg, d.partials_dev, &p->slots, &d.opts, s);
  return hsr_srf_integrate_moments(static_cast<const float*>(cube), d.npix, d.B, d.wn_dev, p->k0, p->klen, d.nb, d.pseudo_dev,
                                   d.out_bs, d.out_ps, real, d.real_bs, d.real_ps, mask, d.min_x, d.min_y, d.deg,
                                   d.partials_dev, &p->slots, &d.opts, s);
}

int run_apply(hsr_step_plan* p, const uint8_t* mask, hipStream_t s) {
  const hsr_step_desc& d = p->d;
  return hsr_poly_apply(d.pseudo_dev, d.out_bs, d.out_ps, d.apply_mask ? mask : nullptr, d.coeffs_dev, d.nb, d.deg, d.npix,
                        nullptr, d.clip, d.matched_dev, d.matched_bs, d.matched_ps, s);
}

}  // namespace

extern "C" int hsr_step_plan_create(const hsr_step_desc* desc, hsr_step_plan** out) {
  HSR_REQUIRE(desc && out, HSR_ERR_INVALID, "hsr_step_plan_create: NULL argument");
  HSR_REQUIRE(desc->nb >= 1 && desc->nb <= HSR_MAX_BANDS && desc->k0 && desc->klen, HSR_ERR_INVALID,
              "hsr_step_plan_create: nb=%d outside [1,%d] or NULL band tables", desc->nb, HSR_MAX_BANDS);
  HSR_REQUIRE(desc->deg >= 1 && desc->deg <= HSR_MAX_DEG, HSR_ERR_INVALID, "hsr_step_plan_create: deg=%d outside [1,%d]",
              desc->deg, HSR_MAX_DEG);
  HSR_REQUIRE(desc->cube_dtype == 0 || desc->cube_dtype == 2, HSR_ERR_INVALID, "hsr_step_plan_create: cube_dtype %d (0 float32, 2 uint16)",
              desc->cube_dtype);
  HSR_REQUIRE(desc->npix >= 1 && desc->wn_dev && desc->pseudo_dev && desc->matched_dev && desc->partials_dev && desc->moments_dev &&
                  desc->coeffs_dev, HSR_ERR_INVALID, "hsr_step_plan_create: NULL device pointer or npix < 1");
  hsr_step_plan* p = new (std::nothrow) hsr_step_plan();
  HSR_REQUIRE(p, HSR_ERR_INVALID, "hsr_step_plan_create: out of host memory");
  p->d = *desc;
  for (int b = 0; b < desc->nb; ++b) {
    p->k0[b] = desc->k0[b];
    p->klen[b] = desc->klen[b];
  }
  p->d.k0 = p->k0;          // the plan owns its copy of the host tables
  p->d.klen = p->klen;
  p->slots = 0;
  p->pending = false;
  p->ev_k1 = p->ev_fit = nullptr;
  if (hipEventCreateWithFlags(&p->ev_k1, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&p->ev_fit, hipEventDisableTiming) != hipSuccess) {
    (void)hipGetLastError();
    if (p->ev_k1) (void)hipEventDestroy(p->ev_k1);
    delete p;
    hsr::set_error("hsr_step_plan_create: hipEventCreate failed");
    return HSR_ERR_HIP;
  }
  *out = p;
  return HSR_OK;
}

extern "C" void hsr_step_plan_destroy(hsr_step_plan* p) {
  if (!p) return;
  (void)hipEventDestroy(p->ev_k1);
  (void)hipEventDestroy(p->ev_fit);
  delete p;
}

extern "C" int hsr_step_plan_slots(const hsr_step_plan* p) { return p ? p->slots : -1; }

extern "C" int hsr_step_run(hsr_step_plan* p, const void* cube_dev, const float* real_dev, const uint8_t* mask_dev,
                            hsr_stream_t stream) {
  HSR_REQUIRE(p && cube_dev && real_dev, HSR_ERR_INVALID, "hsr_step_run: NULL argument");
  hipStream_t s = (hipStream_t)stream;
  int rc = run_k1(p, cube_dev, real_dev, mask_dev, s);
  if (rc != HSR_OK) return rc;
  rc = hsr_moments_reduce_solve(p->d.partials_dev, p->slots, p->d.nb, p->d.deg, p->d.min_count, p->d.moments_dev, p->d.coeffs_dev, s);
  if (rc != HSR_OK) return rc;
  return run_apply(p, mask_dev, s);
}

extern "C" int hsr_step_run_k1(hsr_step_plan* p, const void* cube_dev, const float* real_dev, const uint8_t* mask_dev,
                               hsr_stream_t stream) {
  HSR_REQUIRE(p && cube_dev && real_dev, HSR_ERR_INVALID, "hsr_step_run_k1: NULL argument");
  return run_k1(p, cube_dev, real_dev, mask_dev, (hipStream_t)stream);
}

extern "C" int hsr_step_run_reduce(hsr_step_plan* p, hsr_stream_t stream) {
  HSR_REQUIRE(p, HSR_ERR_INVALID, "hsr_step_run_reduce: NULL plan");
  return hsr_moments_reduce(p->d.partials_dev, p->slots, p->d.nb, p->d.deg, p->d.moments_dev, stream);
}

extern "C" int hsr_step_run_solve(hsr_step_plan* p, hsr_stream_t stream) {
  HSR_REQUIRE(p, HSR_ERR_INVALID, "hsr_step_run_solve: NULL plan");
  return hsr_poly_solve(p->d.moments_dev, p->d.nb, p->d.deg, p->d.min_count, p->d.coeffs_dev, stream);
}

extern "C" int hsr_step_run_apply(hsr_step_plan* p, const uint8_t* mask_dev, hsr_stream_t stream) {
  HSR_REQUIRE(p, HSR_ERR_INVALID, "hsr_step_run_apply: NULL plan");
  return run_apply(p, mask_dev, (hipStream_t)stream);
}

// ---- pipeline ------------------------------------------------------------------------------------------------
//     caller's stream :  K1(0)  K1(1)  K3(0)  K1(2)  K3(1)  ...
//     side stream     :  fit(0)        fit(1)        fit(2) ...          fit(i) runs under K1(i+1)
// K3(i) waits for ev_fit(i) and precedes K1(i+2) in stream order, so two slots need no further events.  The event that
// releases fit(i) is recorded behind K3(i-1), not between K1(i) and K3(i-1) (a record in between cost a 13 us bubble).
static int pipeline_new(hsr_step_plan* const* sl, int nslots, hsr_stream_t side_stream, int32_t exchange, hsr_pipeline** out,
                        const char* who) {
  HSR_REQUIRE(out, HSR_ERR_INVALID, "%s: NULL argument", who);
  for (int i = 0; i < nslots; ++i) {
    HSR_REQUIRE(sl[i], HSR_ERR_INVALID, "%s: NULL plan", who);
    for (int k = 0; k < i; ++k) {
      HSR_REQUIRE(sl[i] != sl[k], HSR_ERR_INVALID, "%s: distinct plans needed", who);
      // In one fused launch the pre-phase reads the coefficients of one slot, the tail writes the moments / coefficients and reads
      // the partials of another, K1 writes the partials of a third: aliased work buffers would race silently.
      const hsr_step_desc &a = sl[i]->d, &b = sl[k]->d;
      HSR_REQUIRE(nslots == 2 || (a.partials_dev != b.partials_dev && a.moments_dev != b.moments_dev && a.coeffs_dev != b.coeffs_dev &&
                                  a.pseudo_dev != b.pseudo_dev && a.matched_dev != b.matched_dev),
                  HSR_ERR_INVALID, "%s: plans %d and %d share a work buffer (partials / moments / coefficients / images must be distinct)", who, k, i);
    }
  }
  HSR_REQUIRE(side_stream != nullptr, HSR_ERR_INVALID, "%s: the side stream must be a real stream, not the default one", who);
  hsr_pipeline* pl = new (std::nothrow) hsr_pipeline();
  HSR_REQUIRE(pl, HSR_ERR_INVALID, "%s: out of host memory", who);
  pl->slot.assign(sl, sl + nslots);
  for (int i = 0; i < 4; ++i) pl->host_moments[i] = nullptr;
  pl->nslots = nslots;
  pl->kind = nslots == 2 ? kTwoSlot : (nslots == 3 ? kFused : kExchange);      // (the create functions set the other kinds)
  pl->group_T = 0;
  pl->group_moments = pl->group_total = pl->group_coeffs = nullptr;
  pl->side = (hipStream_t)side_stream;
  pl->n = 0;
  pl->exchange = exchange ? 1 : 0;
  pl->counter = nullptr;
  pl->sync = nullptr;
  pl->tickets = pl->published = 0;
  pl->x = hsr_exchange{};
  pl->host_error = 0;
  pl->deferred = -1;
  pl->deferred_solve = 0;
  for (int k = 0; k < nslots; ++k) {
    pl->slot[k]->pending = pl->slot[k]->fitted = pl->slot[k]->exchanged = false;
    pl->slot[k]->seq = 0;
  }
  if (nslots >= 3) {              // not a launch-path call: the device words of the fused pipelines
    if (hipMalloc(&pl->sync, 8 * sizeof(unsigned int)) != hipSuccess || hipMemset(pl->sync, 0, 8 * sizeof(unsigned int)) != hipSuccess ||
        hipDeviceSynchronize() != hipSuccess) {
      (void)hipGetLastError();
      if (pl->sync) (void)hipFree(pl->sync);
      delete pl;
      hsr::set_error("%s: could not allocate the pipeline's device words", who);
      return HSR_ERR_HIP;
    }
    pl->counter = pl->sync;
  }
  *out = pl;
  return HSR_OK;
}

extern "C" int hsr_pipeline_create(hsr_step_plan* slot0, hsr_step_plan* slot1, hsr_stream_t side_stream, int32_t exchange,
                                   hsr_pipeline** out) {
  hsr_step_plan* sl[2] = {slot0, slot1};
  return pipeline_new(sl, 2, side_stream, exchange, out, "hsr_pipeline_create");
}

// The fused launches need one tile geometry in every plan, 16-byte aligned pixel-major rows of 4 / 8 / 12 / 16 floats, and a K1
// kernel that can carry the job (hsr_srf_fused_launch_supported: weights in LDS, for uint16 tiles the ring kernel) - checked HERE
// so that a caller can fall back to the two-slot pipeline before any tile is in flight, not at the first carrying launch.
static int fused_geometry(hsr_step_plan* const* ps, int n, const char* who) {
  for (int i = 0; i < n; ++i) {
    HSR_REQUIRE(ps[i], HSR_ERR_INVALID, "%s: NULL plan", who);
    const hsr_step_desc &d = ps[i]->d, &d0 = ps[0]->d;
    HSR_REQUIRE(d.cube_dtype == d0.cube_dtype && d.out_bs == 1 && (d.out_ps & 3) == 0 && d.out_ps <= HSR_MAX_BANDS && d.matched_bs == 1 &&
                    d.matched_ps == d.out_ps && ((((uintptr_t)d.pseudo_dev) | ((uintptr_t)d.matched_dev)) & 15) == 0 &&
                    d.npix == d0.npix && d.out_ps == d0.out_ps && d.nb == d0.nb && d.deg == d0.deg && d.B == d0.B &&
                    d.opts.reserved_cus == d0.opts.reserved_cus,
                HSR_ERR_UNSUPPORTED, "%s: 16-byte aligned pixel-major rows of 4 / 8 / 12 / 16 floats, "
                "the same geometry, cube type and options in all plans", who);
  }
  const hsr_step_desc& d = ps[0]->d;
  return hsr_srf_fused_launch_supported(d.cube_dtype, d.B, d.nb, ps[0]->k0, ps[0]->klen, d.out_ps, d.deg, &d.opts);
}

// Fused pipeline over THREE plans: K3 of tile i-2 rides in the launch of K1 of tile i (hsr_srf_integrate_moments_apply), so the
// caller's stream carries ONE kernel per tile:
//     [K1(0)]  [K1(1) + fit(0)]  [K3(0) + K1(2) + fit(1)]  [K3(1) + K1(3) + fit(2)] ...   nothing else: the fit of tile i is tail
//     work of launch i+1 (first workgroups to finish), no side stream, no events, no free CUs
// No exchange (a collective cannot ride in a kernel's tail): with one, hsr_pipeline_create_exchange.
extern "C" int hsr_pipeline_create_fused(hsr_step_plan* slot0, hsr_step_plan* slot1, hsr_step_plan* slot2, hsr_stream_t side_stream,
                                         int32_t exchange, hsr_pipeline** out) {
  HSR_REQUIRE(!exchange, HSR_ERR_UNSUPPORTED, "hsr_pipeline_create_fused: no exchange in the three-slot form; use hsr_pipeline_create_exchange");
  hsr_step_plan* ps[3] = {slot0, slot1, slot2};
  int rc = fused_geometry(ps, 3, "hsr_pipeline_create_fused");
  if (rc != HSR_OK) return rc;
  return pipeline_new(ps, 3, side_stream, 0, out, "hsr_pipeline_create_fused");
}

extern "C" int hsr_pipeline_create_exchange(hsr_step_plan* const* slots4, hsr_stream_t side_stream, const hsr_exchange* x,
                                            hsr_pipeline** out) {
  const char* who = "hsr_pipeline_create_exchange";
  HSR_REQUIRE(slots4 && x && out, HSR_ERR_INVALID, "%s: NULL argument", who);
  HSR_REQUIRE((x->comm != nullptr) != (x->host_sum != nullptr), HSR_ERR_INVALID, "%s: exactly one of comm and host_sum must be given", who);
  HSR_REQUIRE(x->mode == HSR_SYNC_ALLREDUCE || x->mode == HSR_SYNC_BROADCAST, HSR_ERR_INVALID, "%s: mode %d", who, x->mode);
  HSR_REQUIRE(!x->comm || (x->root >= 0 && x->root < hsr_comm_ranks(x->comm)), HSR_ERR_INVALID, "%s: root %d", who, x->root);
  HSR_REQUIRE(x->rehearsal_us >= 0 && x->rehearsal_us <= 1000 && x->rehearsal_blocks >= 0 && x->rehearsal_blocks <= 64, HSR_ERR_INVALID,
              "%s: rehearsal stand-in of %d us x %d blocks (at most 1000 us, 64 blocks)", who, x->rehearsal_us, x->rehearsal_blocks);
  int rc = fused_geometry(slots4, 4, who);
  if (rc != HSR_OK) return rc;
  // Not a tuning matter: a K1 launch whose pre-phase polls for coefficients holds every CU it runs on while it waits, and the
  // kernels it waits for (the collective, the solve) are dispatched next to it only where an XCD has a completely free CU
  // (measured in round 2).  With fewer free CUs a late peer turns into a wait that only the polls' time limit ends.
  HSR_REQUIRE(slots4[0]->d.opts.reserved_cus >= 8, HSR_ERR_INVALID, "%s: the plans must leave at least 8 CUs free (hsr_srf_options.reserved_cus >= 8, one "
              "per XCD) for the side stream's kernels; got %d", who, slots4[0]->d.opts.reserved_cus);
  hsr_pipeline* pl = nullptr;
  rc = pipeline_new(slots4, 4, side_stream, 1, &pl, who);
  if (rc != HSR_OK) return rc;
  pl->x = *x;
  if (x->host_sum) {
    const size_t bytes = (size_t)pl->slot[0]->d.nb * hsr::moment_count(pl->slot[0]->d.deg) * sizeof(double);
    for (int k = 0; k < 4; ++k) {
      if (hipHostMalloc(&pl->host_moments[k], bytes, hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        hsr_pipeline_destroy(pl);
        hsr::set_error("%s: could not allocate the pinned staging of the host transport", who);
        return HSR_ERR_HIP;
      }
      pl->host_jobs[k] = {pl, pl->host_moments[k], (int32_t)(bytes / sizeof(double))};
    }
  }
  *out = pl;
  return HSR_OK;
}

extern "C" int hsr_pipeline_create_group(hsr_step_plan* const* slots, int32_t nslots, int32_t group_tiles, double* group_moments_dev,
                                         double* group_total_dev, double* group_coeffs_dev, hsr_stream_t side_stream, hsr_pipeline** out) {
  const char* who = "hsr_pipeline_create_group";
  HSR_REQUIRE(slots && out && group_moments_dev && group_total_dev && group_coeffs_dev, HSR_ERR_INVALID, "%s: NULL argument", who);
  HSR_REQUIRE(group_tiles >= 2 && group_tiles <= 64 && nslots == group_tiles + 2, HSR_ERR_INVALID,
              "%s: groups of 2 .. 64 tiles and group_tiles + 2 plans needed (got %d tiles, %d plans)", who, group_tiles, nslots);
  int rc = fused_geometry(slots, nslots, who);
  if (rc != HSR_OK) return rc;
  hsr_pipeline* pl = nullptr;
  rc = pipeline_new(slots, nslots, side_stream, 0, &pl, who);
  if (rc != HSR_OK) return rc;
  pl->kind = kGroup;
  pl->group_T = group_tiles;
  pl->group_moments = group_moments_dev;
  pl->group_total = group_total_dev;
  pl->group_coeffs = group_coeffs_dev;
  *out = pl;
  return HSR_OK;
}

extern "C" void hsr_pipeline_destroy(hsr_pipeline* pl) {
  if (!pl) return;
  if (pl->kind == kExchange) (void)hipStreamSynchronize(pl->side);      // a host_sum callback may still point at this object
  for (int k = 0; k < 4; ++k)
    if (pl->host_moments[k]) (void)hipHostFree(pl->host_moments[k]);
  if (pl->sync) (void)hipFree(pl->sync);
  delete pl;
}

static int finish_slot(hsr_pipeline* pl, hsr_step_plan* p, const uint8_t* mask, hipStream_t main) {
  int rc = HSR_OK;
  if (pl->kind == kFused) {                    // tail-fit pipeline: everything lives on the caller's stream
    if (!p->fitted)
      rc = hsr_moments_reduce_solve(p->d.partials_dev, p->slots, p->d.nb, p->d.deg, p->d.min_count, p->d.moments_dev, p->d.coeffs_dev, main);
    p->fitted = true;
  } else {
    rc = hsr::check_hip(hipStreamWaitEvent(main, p->ev_fit, 0), "hsr_pipeline: wait for the fit");
  }
  if (rc != HSR_OK) return rc;
  rc = run_apply(p, mask, main);
  p->pending = false;
  return rc;
}

// ---- four slots: the exchange issued from here ------------------------------------------------------------------------
static void host_sum_trampoline(void* arg) {
  auto* j = static_cast<hsr_pipeline::host_job*>(arg);
  if (j->pl->x.host_sum(j->pl->x.host_user, j->values, j->count) != 0) j->pl->host_error = 1;
}

// The slot reduction of a tile as a launch of its own on the caller's stream (tiles whose follower cannot carry it in its tail:
// fewer workgroups than bands, or no follower at a drain), published like a tail reduction.
static int reduce_and_publish(hsr_pipeline* pl, hsr_step_plan* p, hipStream_t main) {
  int rc = hsr_moments_reduce(p->d.partials_dev, p->slots, p->d.nb, p->d.deg, p->d.moments_dev, main);
  if (rc != HSR_OK) return rc;
  hipLaunchKernelGGL(publish_add_kernel, dim3(1), dim3(64), 0, main, pl->sync + 1, (unsigned int)p->d.nb);
  HSR_LAUNCH_CHECK("publish_add_kernel");
  pl->published += (unsigned int)p->d.nb;
  p->fitted = true;
  return HSR_OK;
}

// solve + publish of the deferred slot, optionally going on as the gate of the next tile's collective
static int launch_solve_publish(hsr_pipeline* pl, bool with_gate) {
  hsr_step_plan* p = pl->slot[pl->deferred];
  const hsr_step_desc& d = p->d;
  hipLaunchKernelGGL(solve_publish_kernel, dim3(1), dim3(64), 0, pl->side, d.moments_dev, d.nb, d.deg, (long long)d.min_count, d.coeffs_dev,
                     pl->deferred_solve, pl->sync + 4 + pl->deferred, p->seq, with_gate ? pl->sync + 1 : nullptr, pl->published, pl->sync + 2);
  HSR_LAUNCH_CHECK("solve_publish_kernel");
  pl->deferred = -1;
  return HSR_OK;
}

// gate -> collective -> solve + publish of the tile in slot k, on the side stream.  Called once per tile, in tile order, on every
// rank: the collectives of all ranks line up.  The solve + publish itself is enqueued with the NEXT tile's gate (or by a drain).
static int enqueue_exchange(hsr_pipeline* pl, int k) {
  hsr_step_plan* p = pl->slot[k];
  const hsr_step_desc& d = p->d;
  const int64_t nmom = (int64_t)d.nb * hsr::moment_count(d.deg);
  int rc = HSR_OK;
  if (pl->deferred >= 0) {
    rc = launch_solve_publish(pl, true);       // the previous tile's solve, then this tile's gate
    if (rc != HSR_OK) return rc;
  } else {
    hipLaunchKernelGGL(gate_kernel, dim3(1), dim3(64), 0, pl->side, pl->sync + 1, pl->published, pl->sync + 2, 1u);
    HSR_LAUNCH_CHECK("gate_kernel");
  }
  int solve_here = 1;
  if (pl->x.host_sum) {
    double* h = pl->host_moments[k];
    rc = hsr::check_hip(hipMemcpyAsync(h, d.moments_dev, nmom * sizeof(double), hipMemcpyDeviceToHost, pl->side), "hsr_pipeline: moments to the host");
    if (rc == HSR_OK) rc = hsr::check_hip(hipLaunchHostFunc(pl->side, host_sum_trampoline, &pl->host_jobs[k]), "hsr_pipeline: host_sum");
    if (rc == HSR_OK) rc = hsr::check_hip(hipMemcpyAsync(d.moments_dev, h, nmom * sizeof(double), hipMemcpyHostToDevice, pl->side), "hsr_pipeline: moments back");
  } else if (pl->x.mode == HSR_SYNC_ALLREDUCE) {
    rc = hsr_allreduce_f64(pl->x.comm, d.moments_dev, nmom, pl->side);
  } else {
    rc = hsr_reduce_f64(pl->x.comm, d.moments_dev, nmom, pl->x.root, pl->side);
    if (rc != HSR_OK) return rc;
    rc = hsr_poly_solve(d.moments_dev, d.nb, d.deg, d.min_count, d.coeffs_dev, pl->side);      // only the root's is kept
    if (rc != HSR_OK) return rc;
    rc = hsr_bcast(pl->x.comm, d.coeffs_dev, (int64_t)d.nb * (d.deg + 1) * (int64_t)sizeof(double), pl->x.root, pl->side);
    solve_here = 0;
  }
  if (rc != HSR_OK) return rc;
  if (pl->x.rehearsal_us > 0) {
    hipLaunchKernelGGL(rehearsal_collective_kernel, dim3(pl->x.rehearsal_blocks > 0 ? pl->x.rehearsal_blocks : 1), dim3(256), 0, pl->side,
                       (int)pl->x.rehearsal_us, pl->sync + 3);
    HSR_LAUNCH_CHECK("rehearsal_collective_kernel");
  }
  pl->deferred = k;
  pl->deferred_solve = solve_here;
  p->exchanged = true;
  return HSR_OK;
}

static int submit_exchange(hsr_pipeline* pl, const void* cube_dev, const float* real_dev, const uint8_t* mask_dev,
                           const uint8_t* prev_mask_dev, hipStream_t main, int32_t* finished_slot, void* k1_begin_event,
                           void* k1_end_event) {
  const int cur = (int)(pl->n % 4);
  hsr_step_plan* p = pl->slot[cur];
  const int ko = (int)((pl->n + 1) % 4), kl = (int)((pl->n + 3) % 4);
  hsr_step_plan* old = pl->n >= 3 ? pl->slot[ko] : nullptr;        // tile n - 3: its K3 rides in this launch
  hsr_step_plan* last = pl->n >= 1 ? pl->slot[kl] : nullptr;       // tile n - 1: its slot reduction rides in this launch's tail
  int rc = HSR_OK;
  hsr_apply_job job{};
  job.sync_error_dev = pl->sync + 2;
  const bool carry = old && old->pending;
  if (carry) {
    HSR_REQUIRE(old->exchanged, HSR_ERR_INVALID, "hsr_pipeline_submit: tile %lld has no exchange enqueued", (long long)(pl->n - 3));
    job.x_dev = old->d.pseudo_dev;
    job.out_dev = old->d.matched_dev;
    job.coeffs_dev = old->d.coeffs_dev;
    job.mask_dev = old->d.apply_mask ? prev_mask_dev : nullptr;
    job.npix = old->d.npix;
    job.clip = old->d.clip;
    job.coeffs_ready_dev = pl->sync + 4 + ko;
    job.coeffs_ready_value = old->seq;
  }
  const int grid = hsr_partial_slots(p->d.npix, &p->d.opts);
  const bool reduce_last = last && last->pending && !last->fitted;
  const bool ride = reduce_last && grid >= p->d.nb;                // one ticket per workgroup, one band per ticket
  if (ride) {
    job.fit_partials_dev = last->d.partials_dev;
    job.fit_slots = last->slots;
    job.fit_moments_dev = last->d.moments_dev;
    job.fit_coeffs_dev = last->d.coeffs_dev;                       // not written in this mode
    job.fit_min_count = last->d.min_count;
    job.fit_counter_dev = pl->sync;
    job.fit_ticket_base = pl->tickets;
    job.fit_ready_dev = pl->sync + 1;
  } else if (reduce_last) {
    rc = reduce_and_publish(pl, last, main);
    if (rc != HSR_OK) return rc;
  }
  if (k1_begin_event) rc = hsr::check_hip(hipEventRecord((hipEvent_t)k1_begin_event, main), "hsr_pipeline: record K1 begin");
  if (rc != HSR_OK) return rc;
  rc = run_k1(p, cube_dev, real_dev, mask_dev, main, (carry || ride) ? &job : nullptr);
  if (rc != HSR_OK) return rc;
  if (ride) {
    HSR_REQUIRE(p->slots == grid, HSR_ERR_INVALID, "hsr_pipeline_submit: the launch used %d workgroups, %d expected", p->slots, grid);
    pl->tickets += (unsigned int)p->slots;     // every workgroup of the launch drew one ticket
    pl->published += (unsigned int)p->d.nb;
    last->fitted = true;
  }
  if (k1_end_event) rc = hsr::check_hip(hipEventRecord((hipEvent_t)k1_end_event, main), "hsr_pipeline: record K1 end");
  if (rc != HSR_OK) return rc;
  if (reduce_last) {                           // its moments are (or will be, when this launch's tail runs) published: the exchange
    rc = enqueue_exchange(pl, kl);
    if (rc != HSR_OK) return rc;
  }
  if (carry) {
    old->pending = false;
    if (finished_slot) *finished_slot = ko;
  }
  p->pending = true;
  p->fitted = p->exchanged = false;
  p->seq = (unsigned int)(pl->n + 1);
  pl->n += 1;
  return HSR_OK;
}

// finish the tile in slot k outside a K1 launch (drain): reduction and exchange if they have not been enqueued yet, then K3 as its
// own launch behind an event of the side stream
static int finish_exchange_slot(hsr_pipeline* pl, int k, const uint8_t* mask, hipStream_t main) {
  hsr_step_plan* p = pl->slot[k];
  int rc = HSR_OK;
  if (!p->fitted) rc = reduce_and_publish(pl, p, main);
  if (rc == HSR_OK && !p->exchanged) rc = enqueue_exchange(pl, k);
  if (rc == HSR_OK && pl->deferred >= 0) rc = launch_solve_publish(pl, false);     // (this tile's, or a later one's: at most one is deferred)
  if (rc != HSR_OK) return rc;
  // (an EVENT here, not a polling gate: this is a drain, a bubble costs nothing - and a wave spinning on the caller's stream would
  // deadlock, until its time limit, against side-stream work queued behind it if the runtime serves both streams from one
  // hardware queue.  Everything the side stream still holds in front of the record is released by launches that are already
  // enqueued on the caller's stream.)
  rc = hsr::check_hip(hipEventRecord(p->ev_fit, pl->side), "hsr_pipeline: record fit");
  if (rc == HSR_OK) rc = hsr::check_hip(hipStreamWaitEvent(main, p->ev_fit, 0), "hsr_pipeline: wait for the fit");
  if (rc != HSR_OK) return rc;
  rc = run_apply(p, mask, main);
  p->pending = false;
  return rc;
}

// ---- T + 2 slots: ONE fit per group of T consecutive tiles (a mosaic held by one GPU) ----------------------------------
// Launch n = K3 of tile n - (T + 1) as the pre-phase | K1+K2 of tile n | slot reduction of tile n - 1 in the tail; the tail of the launch
// that follows a group's LAST tile also adds the group's T per-tile moment sets (the tree of hsr_moments_reduce over T slots) and
// solves.  A tile's K3 therefore rides T + 1 launches later, when its group's polynomial has been ready for at least one launch:
// nothing but one kernel per tile on the caller's stream, no side stream, no events, no CUs kept free.  Buffers of two
// consecutive groups never alias (parity).
static inline size_t group_mom_doubles(const hsr_pipeline* pl) {
  return (size_t)pl->slot[0]->d.nb * hsr::moment_count(pl->slot[0]->d.deg);
}
static inline double* group_entry(const hsr_pipeline* pl, int64_t tile) {      // moments of tile `tile` inside its group's array
  const int64_t g = tile / pl->group_T, i = tile % pl->group_T;
  return pl->group_moments + ((size_t)(g & 1) * pl->group_T + (size_t)i) * group_mom_doubles(pl);
}
static inline double* group_total_of(const hsr_pipeline* pl, int64_t tile) { return pl->group_total + (size_t)((tile / pl->group_T) & 1) * group_mom_doubles(pl); }
static inline double* group_coeffs_of(const hsr_pipeline* pl, int64_t tile) {
  return pl->group_coeffs + (size_t)((tile / pl->group_T) & 1) * pl->slot[0]->d.nb * (pl->slot[0]->d.deg + 1);
}

// slot reduction of tile `tile` (and, for a group's last tile, the group's fit) as launches of their own
static int group_fit_standalone(hsr_pipeline* pl, int64_t tile, hipStream_t main) {
  hsr_step_plan* p = pl->slot[tile % pl->nslots];
  const hsr_step_desc& d = p->d;
  int rc = hsr_moments_reduce(d.partials_dev, p->slots, d.nb, d.deg, group_entry(pl, tile), main);
  if (rc == HSR_OK && tile % pl->group_T == pl->group_T - 1)
    rc = hsr_moments_reduce_solve(group_entry(pl, tile - (pl->group_T - 1)), pl->group_T, d.nb, d.deg, d.min_count, group_total_of(pl, tile),
                                  group_coeffs_of(pl, tile), main);
  p->fitted = true;
  return rc;
}

static int submit_group(hsr_pipeline* pl, const void* cube_dev, const float* real_dev, const uint8_t* mask_dev,
                        const uint8_t* prev_mask_dev, hipStream_t main, int32_t* finished_slot, void* k1_begin_event,
                        void* k1_end_event) {
  const int S = pl->nslots, T = pl->group_T;
  const int64_t n = pl->n;
  hsr_step_plan* p = pl->slot[n % S];
  hsr_step_plan* old = n >= T + 1 ? pl->slot[(n - (T + 1)) % S] : nullptr;      // tile n - (T + 1): its K3 rides in this launch
  hsr_step_plan* last = n >= 1 ? pl->slot[(n - 1) % S] : nullptr;               // tile n - 1: its slot reduction rides in the tail
  int rc = HSR_OK;
  hsr_apply_job job{};
  const bool carry = old && old->pending;
  if (carry) {
    job.x_dev = old->d.pseudo_dev;
    job.out_dev = old->d.matched_dev;
    job.coeffs_dev = group_coeffs_of(pl, n - (T + 1));
    job.mask_dev = old->d.apply_mask ? prev_mask_dev : nullptr;
    job.npix = old->d.npix;
    job.clip = old->d.clip;
  }
  const int grid = hsr_partial_slots(p->d.npix, &p->d.opts);
  const bool fit_last = last && last->pending && !last->fitted;
  const bool ride = fit_last && grid >= p->d.nb;
  if (ride) {
    job.fit_partials_dev = last->d.partials_dev;
    job.fit_slots = last->slots;
    job.fit_moments_dev = group_entry(pl, n - 1);
    job.fit_coeffs_dev = group_coeffs_of(pl, n - 1);
    job.fit_min_count = last->d.min_count;
    job.fit_counter_dev = pl->counter;
    job.fit_ticket_base = pl->tickets;
    job.fit_group_tiles = T;
    job.fit_group_index = (int32_t)((n - 1) % T);
    job.fit_group_moments_dev = group_entry(pl, (n - 1) - (n - 1) % T);
    job.fit_group_total_dev = group_total_of(pl, n - 1);
  } else if (fit_last) {
    rc = group_fit_standalone(pl, n - 1, main);
    if (rc != HSR_OK) return rc;
  }
  if (k1_begin_event) rc = hsr::check_hip(hipEventRecord((hipEvent_t)k1_begin_event, main), "hsr_pipeline: record K1 begin");
  if (rc != HSR_OK) return rc;
  rc = run_k1(p, cube_dev, real_dev, mask_dev, main, (carry || ride) ? &job : nullptr);
  if (rc != HSR_OK) return rc;
  if (ride) {
    HSR_REQUIRE(p->slots == grid, HSR_ERR_INVALID, "hsr_pipeline_submit: the launch used %d workgroups, %d expected", p->slots, grid);
    pl->tickets += (unsigned int)p->slots;
    last->fitted = true;
  }
  if (k1_end_event) rc = hsr::check_hip(hipEventRecord((hipEvent_t)k1_end_event, main), "hsr_pipeline: record K1 end");
  if (rc != HSR_OK) return rc;
  if (carry) {
    old->pending = false;
    if (finished_slot) *finished_slot = (int)((n - (T + 1)) % S);
  }
  p->pending = true;
  p->fitted = false;
  pl->n += 1;
  return HSR_OK;
}

// drain: tile i (the oldest unfinished one).  Whole groups only: a group whose last tile has not been submitted has no polynomial.
static int finish_group_slot(hsr_pipeline* pl, int64_t i, const uint8_t* mask, hipStream_t main) {
  const int T = pl->group_T;
  HSR_REQUIRE(pl->n % T == 0, HSR_ERR_INVALID, "hsr_pipeline_flush: %lld tiles submitted, not a whole number of groups of %d - the last group "
              "has no fit yet", (long long)pl->n, T);
  int rc = HSR_OK;
  for (int64_t k = i; k < pl->n && rc == HSR_OK; ++k) {          // every reduction up to the end of the tile's group (only the last tile's can be open)
    hsr_step_plan* q = pl->slot[k % pl->nslots];
    if (q->pending && !q->fitted) rc = group_fit_standalone(pl, k, main);
  }
  if (rc != HSR_OK) return rc;
  hsr_step_plan* p = pl->slot[i % pl->nslots];
  const hsr_step_desc& d = p->d;
  rc = hsr_poly_apply(d.pseudo_dev, d.out_bs, d.out_ps, d.apply_mask ? mask : nullptr, group_coeffs_of(pl, i), d.nb, d.deg, d.npix, nullptr, d.clip,
                      d.matched_dev, d.matched_bs, d.matched_ps, main);
  p->pending = false;
  return rc;
}

// Starts tile i in slot i % S and finishes tile i-1 (two slots), i-2 (fused) or i-3 (fused with exchange): its K3.
// *finished_slot = slot of the finished tile, or -1.  prev_mask_dev: the mask of the tile being finished (only read when the
// plan applies the mask in K3).
// Two slots, exchange = 0: the fit (slot reduction + solve) is enqueued on the side stream here.  With exchange = 1 the side stream
// has been made to wait for K1(i) when this returns; the caller enqueues reduce -> collective -> solve on it and then calls
// hsr_pipeline_fit_done.
extern "C" int hsr_pipeline_submit(hsr_pipeline* pl, const void* cube_dev, const float* real_dev, const uint8_t* mask_dev,
                                   const uint8_t* prev_mask_dev, hsr_stream_t main_stream, int32_t* finished_slot,
                                   void* k1_begin_event, void* k1_end_event) {
  HSR_REQUIRE(pl && cube_dev && real_dev, HSR_ERR_INVALID, "hsr_pipeline_submit: NULL argument");
  hipStream_t main = (hipStream_t)main_stream;
  const int S = pl->nslots;
  const int cur = (int)(pl->n % S);
  hsr_step_plan* p = pl->slot[cur];
  HSR_REQUIRE(!p->pending, HSR_ERR_INVALID, "hsr_pipeline_submit: slot %d still holds an unfinished tile", cur);
  int rc = HSR_OK;
  if (finished_slot) *finished_slot = -1;
  if (pl->kind == kExchange) return submit_exchange(pl, cube_dev, real_dev, mask_dev, prev_mask_dev, main, finished_slot, k1_begin_event, k1_end_event);
  if (pl->kind == kGroup) return submit_group(pl, cube_dev, real_dev, mask_dev, prev_mask_dev, main, finished_slot, k1_begin_event, k1_end_event);
  if (pl->kind == kFused) {
    // fused: this launch carries K3 of tile n - 2 (slot (n + 1) % 3); its fit has had all of K1(n - 1) to finish
    hsr_step_plan* old = pl->n >= 2 ? pl->slot[(pl->n + 1) % 3] : nullptr;
    hsr_apply_job job{};
    const bool carry = old && old->pending;
    if (carry) {
      job.x_dev = old->d.pseudo_dev;
      job.out_dev = old->d.matched_dev;
      job.coeffs_dev = old->d.coeffs_dev;
      job.mask_dev = old->d.apply_mask ? prev_mask_dev : nullptr;
      job.npix = old->d.npix;
      job.clip = old->d.clip;
    }
    hsr_step_plan* last = pl->n >= 1 ? pl->slot[(pl->n + 2) % 3] : nullptr;     // tile n - 1: its fit rides in this launch's tail
    const int grid = hsr_partial_slots(p->d.npix, &p->d.opts);
    // (every workgroup of the launch draws ONE ticket and tickets 0 .. nb-1 fit one band each: a launch of fewer workgroups
    // than bands - a tile of fewer than nb 64-pixel groups - cannot carry the fit; that tile's fit runs as its own launch
    // below, when its K3 comes up.  Found by tools/dbg/stress_fused.py: a 2 x 158 tile with 7 bands kept two stale rows.)
    if (last && last->pending && !last->fitted && grid >= p->d.nb) {
      job.fit_partials_dev = last->d.partials_dev;
      job.fit_slots = last->slots;
      job.fit_moments_dev = last->d.moments_dev;
      job.fit_coeffs_dev = last->d.coeffs_dev;
      job.fit_min_count = last->d.min_count;
      job.fit_counter_dev = pl->counter;
      job.fit_ticket_base = pl->tickets;
    }
    if (carry && !old->fitted) {                 // its fit did not ride in the previous launch (see above): a launch of its own
      rc = hsr_moments_reduce_solve(old->d.partials_dev, old->slots, old->d.nb, old->d.deg, old->d.min_count, old->d.moments_dev,
                                    old->d.coeffs_dev, main);
      if (rc != HSR_OK) return rc;
      old->fitted = true;
    }
    if (k1_begin_event) rc = hsr::check_hip(hipEventRecord((hipEvent_t)k1_begin_event, main), "hsr_pipeline: record K1 begin");
    if (rc != HSR_OK) return rc;
    rc = run_k1(p, cube_dev, real_dev, mask_dev, main, (carry || job.fit_partials_dev) ? &job : nullptr);
    if (rc != HSR_OK) return rc;
    if (job.fit_partials_dev) {
      // the ticket base advances by the workgroups the launch REALLY had (run_k1 reports them), and that must be the number the
      // "can this launch carry the fit" test above was made with
      HSR_REQUIRE(p->slots == grid, HSR_ERR_INVALID, "hsr_pipeline_submit: the launch used %d workgroups, %d expected", p->slots, grid);
      pl->tickets += (unsigned int)p->slots;
      last->fitted = true;
    }
    if (k1_end_event) rc = hsr::check_hip(hipEventRecord((hipEvent_t)k1_end_event, main), "hsr_pipeline: record K1 end");
    if (rc != HSR_OK) return rc;
    if (carry) {
      old->pending = false;
      if (finished_slot) *finished_slot = (int)((pl->n + 1) % 3);
    }
    p->pending = true;                    // tail fits: nothing on the side stream, no events
    p->fitted = false;
    pl->n += 1;
    return HSR_OK;
  }
  hsr_step_plan* prev = pl->slot[cur ^ 1];
  if (k1_begin_event) rc = hsr::check_hip(hipEventRecord((hipEvent_t)k1_begin_event, main), "hsr_pipeline: record K1 begin");
  if (rc != HSR_OK) return rc;
  rc = run_k1(p, cube_dev, real_dev, mask_dev, main);
  if (rc != HSR_OK) return rc;
  if (k1_end_event) rc = hsr::check_hip(hipEventRecord((hipEvent_t)k1_end_event, main), "hsr_pipeline: record K1 end");
  if (rc != HSR_OK) return rc;
  if (prev->pending) {
    rc = finish_slot(pl, prev, prev_mask_dev, main);
    if (rc != HSR_OK) return rc;
    if (finished_slot) *finished_slot = cur ^ 1;
  }
  rc = hsr::check_hip(hipEventRecord(p->ev_k1, main), "hsr_pipeline: record K1");
  if (rc != HSR_OK) return rc;
  rc = hsr::check_hip(hipStreamWaitEvent(pl->side, p->ev_k1, 0), "hsr_pipeline: side stream wait");
  if (rc != HSR_OK) return rc;
  if (!pl->exchange) {
    rc = hsr_moments_reduce_solve(p->d.partials_dev, p->slots, p->d.nb, p->d.deg, p->d.min_count, p->d.moments_dev,
                                  p->d.coeffs_dev, pl->side);
    if (rc != HSR_OK) return rc;
    rc = hsr::check_hip(hipEventRecord(p->ev_fit, pl->side), "hsr_pipeline: record fit");
    if (rc != HSR_OK) return rc;
  }
  p->pending = true;
  pl->n += 1;
  return HSR_OK;
}

// two slots, exchange = 1: the caller has enqueued the fit of the slot submitted last on the side stream.
extern "C" int hsr_pipeline_fit_done(hsr_pipeline* pl) {
  HSR_REQUIRE(pl && pl->n > 0 && pl->kind == kTwoSlot, HSR_ERR_INVALID, "hsr_pipeline_fit_done: nothing submitted, or not a two-slot pipeline");
  hsr_step_plan* p = pl->slot[(pl->n - 1) % pl->nslots];
  return hsr::check_hip(hipEventRecord(p->ev_fit, pl->side), "hsr_pipeline: record fit");
}

// K3 of the OLDEST tile left in the pipeline; *finished_slot = its slot or -1.
extern "C" int hsr_pipeline_flush(hsr_pipeline* pl, const uint8_t* mask_dev, hsr_stream_t main_stream, int32_t* finished_slot) {
  HSR_REQUIRE(pl, HSR_ERR_INVALID, "hsr_pipeline_flush: NULL pipeline");
  if (finished_slot) *finished_slot = -1;
  const int S = pl->nslots;
  for (int64_t i = pl->n >= S - 1 ? pl->n - (S - 1) : 0; i < pl->n; ++i) {
    hsr_step_plan* p = pl->slot[i % S];
    if (!p->pending) continue;
    int rc = pl->kind == kExchange ? finish_exchange_slot(pl, (int)(i % S), mask_dev, (hipStream_t)main_stream)
             : pl->kind == kGroup  ? finish_group_slot(pl, i, mask_dev, (hipStream_t)main_stream)
                                   : finish_slot(pl, p, mask_dev, (hipStream_t)main_stream);
    if (rc == HSR_OK && finished_slot) *finished_slot = (int)(i % S);
    return rc;
  }
  return HSR_OK;
}

extern "C" int64_t hsr_pipeline_count(const hsr_pipeline* pl) { return pl ? pl->n : -1; }

extern "C" int hsr_pipeline_status(hsr_pipeline* pl, hsr_stream_t main_stream, uint32_t* sync_error_out) {
  HSR_REQUIRE(pl && sync_error_out, HSR_ERR_INVALID, "hsr_pipeline_status: NULL argument");
  *sync_error_out = 0;
  int rc = hsr::check_hip(hipStreamSynchronize((hipStream_t)main_stream), "hsr_pipeline_status: caller's stream");
  if (rc == HSR_OK) rc = hsr::check_hip(hipStreamSynchronize(pl->side), "hsr_pipeline_status: side stream");
  if (rc != HSR_OK || !pl->sync) return rc;
  unsigned int code = 0;
  rc = hsr::check_hip(hipMemcpy(&code, pl->sync + 2, sizeof code, hipMemcpyDeviceToHost), "hsr_pipeline_status: read the error word");
  *sync_error_out = code | (pl->host_error ? 16u : 0u);
  return rc;
}
