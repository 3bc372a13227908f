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

#include "hsr_common.h"

struct hsr_step_plan {
  hsr_step_desc d;
  int32_t k0[HSR_MAX_BANDS], klen[HSR_MAX_BANDS];
  int32_t slots;
  hipEvent_t ev_k1, ev_fit;       // pipeline: K1 of this slot enqueued / fit of this slot done
  bool pending;                   // pipeline: K1 + fit enqueued, K3 not yet
  bool fitted;                    // fused pipeline without an exchange: the tail fit of this slot's tile has been enqueued
};

struct hsr_pipeline {
  hsr_step_plan* slot[3];
  int nslots;                     // 2: K3(i-1) as its own launch behind K1(i);  3 (fused): K3(i-2) inside K1(i)'s launch
  hipStream_t side;
  int64_t n;                      // tiles submitted
  int exchange;                   // 1: the caller runs the fit (reduce -> collective -> solve) itself between
                                  //    hsr_pipeline_submit and hsr_pipeline_fit_done
  unsigned int* counter;          // fused, no exchange: ticket counter of the tail fits (device), and what it holds
  unsigned int tickets;
};

namespace {

// One wave that sleeps for ~n x 3.4 us (s_sleep 127 = 8 128 cycles).  It opens the side stream's work in the FUSED pipeline:
// there the fit of tile i becomes runnable at the very moment K1 of tile i+1 does, and when the fit's workgroups were placed
// first they sat on CUs K1 needs two full workgroup slots of - K1 then ran a second round for the workgroups that did not fit
// (rocprofv3, r03: 292 us instead of 205 us, the 6 us fit kernel 170 us).  The sleeper needs 1 wave, no LDS and a handful of
// registers, so it fits next to a full set of K1 workgroups wherever it lands; when it ends K1 is resident and the fit goes to
// the CUs K1 leaves free (hsr_srf_options.reserved_cus), as in the two-slot pipeline.
__global__ void side_delay_kernel(int n) {
  for (int i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(127);
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
static int pipeline_new(hsr_step_plan* s0, hsr_step_plan* s1, hsr_step_plan* s2, hsr_stream_t side_stream, int32_t exchange,
                        hsr_pipeline** out, const char* who) {
  HSR_REQUIRE(s0 && s1 && s0 != s1 && out, HSR_ERR_INVALID, "%s: distinct plans needed", who);
  HSR_REQUIRE(side_stream != nullptr, HSR_ERR_INVALID, "%s: the side stream must be a real stream, not the default one", who);
  hsr_pipeline* pl = new (std::nothrow) hsr_pipeline();
  HSR_REQUIRE(pl, HSR_ERR_INVALID, "%s: out of host memory", who);
  pl->slot[0] = s0;
  pl->slot[1] = s1;
  pl->slot[2] = s2;
  pl->nslots = s2 ? 3 : 2;
  pl->side = (hipStream_t)side_stream;
  pl->n = 0;
  pl->exchange = exchange ? 1 : 0;
  pl->counter = nullptr;
  pl->tickets = 0;
  for (int k = 0; k < pl->nslots; ++k) pl->slot[k]->pending = pl->slot[k]->fitted = false;
  if (s2 && !exchange) {          // not a launch-path call: the tail fits' ticket counter
    if (hipMalloc(&pl->counter, sizeof(unsigned int)) != hipSuccess || hipMemset(pl->counter, 0, sizeof(unsigned int)) != hipSuccess) {
      (void)hipGetLastError();
      delete pl;
      hsr::set_error("%s: could not allocate the ticket counter", who);
      return HSR_ERR_HIP;
    }
  }
  *out = pl;
  return HSR_OK;
}

extern "C" int hsr_pipeline_create(hsr_step_plan* slot0, hsr_step_plan* slot1, hsr_stream_t side_stream, int32_t exchange,
                                   hsr_pipeline** out) {
  return pipeline_new(slot0, slot1, nullptr, side_stream, exchange, out, "hsr_pipeline_create");
}

// Fused pipeline over THREE plans: K3 of tile i-2 rides in the launch of K1 of tile i (hsr_srf_integrate_moments_apply), so the
// caller's stream carries ONE kernel per tile:
//     exchange = 0 :  [K1(0)]  [K1(1) + fit(0)]  [K3(0) + K1(2) + fit(1)]  [K3(1) + K1(3) + fit(2)] ...   nothing else: the fit of
//                     tile i is tail work of launch i+1 (first workgroups to finish), no side stream, no events, no free CUs
//     exchange = 1 :  [K1(0)]  [K1(1)]  [K3(0) + K1(2)] ...   + side stream: delay, then the caller's reduce -> collective -> solve
// All three plans must describe the same float32 tile geometry with 16-byte aligned pixel-major rows of 4 / 8 / 12 / 16 floats.
extern "C" int hsr_pipeline_create_fused(hsr_step_plan* slot0, hsr_step_plan* slot1, hsr_step_plan* slot2, hsr_stream_t side_stream,
                                         int32_t exchange, hsr_pipeline** out) {
  HSR_REQUIRE(slot0 && slot1 && slot2 && slot2 != slot0 && slot2 != slot1, HSR_ERR_INVALID, "hsr_pipeline_create_fused: three distinct plans needed");
  hsr_step_plan* ps[3] = {slot0, slot1, slot2};
  for (hsr_step_plan* p : ps) {
    const hsr_step_desc& d = p->d;
    HSR_REQUIRE(d.cube_dtype == slot0->d.cube_dtype && d.out_bs == 1 && (d.out_ps & 3) == 0 && d.out_ps <= HSR_MAX_BANDS && d.matched_bs == 1 &&
                    d.matched_ps == d.out_ps && ((((uintptr_t)d.pseudo_dev) | ((uintptr_t)d.matched_dev)) & 15) == 0 &&
                    d.npix == slot0->d.npix && d.out_ps == slot0->d.out_ps && d.nb == slot0->d.nb && d.deg == slot0->d.deg,
                HSR_ERR_UNSUPPORTED, "hsr_pipeline_create_fused: 16-byte aligned pixel-major rows of 4 / 8 / 12 / 16 floats, "
                "the same geometry and cube type in all three plans");
  }
  return pipeline_new(slot0, slot1, slot2, side_stream, exchange, out, "hsr_pipeline_create_fused");
}

extern "C" void hsr_pipeline_destroy(hsr_pipeline* pl) {
  if (!pl) return;
  if (pl->counter) (void)hipFree(pl->counter);
  delete pl;
}

static int finish_slot(hsr_pipeline* pl, hsr_step_plan* p, const uint8_t* mask, hipStream_t main) {
  int rc = HSR_OK;
  if (pl->nslots == 3 && !pl->exchange) {      // tail-fit pipeline: everything lives on the caller's stream
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

// Starts tile i in slot i % 2 and finishes tile i-1 (its K3).  *finished_slot = slot of the finished tile, or -1.
// prev_mask_dev: the mask of tile i-1 (only read when the plan applies the mask in K3).
// With exchange = 0 the fit (slot reduction + solve) is enqueued on the side stream here.  With exchange = 1 the side stream
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
  if (S == 3) {
    // fused: this launch carries K3 of tile n - 2 (slot (n + 1) % 3); its fit has had all of K1(n - 1) to finish
    hsr_step_plan* old = pl->n >= 2 ? pl->slot[(pl->n + 1) % 3] : nullptr;
    hsr_apply_job job{};
    const bool carry = old && old->pending;
    const bool tail = pl->exchange == 0;
    if (carry) {
      if (!tail) {
        rc = hsr::check_hip(hipStreamWaitEvent(main, old->ev_fit, 0), "hsr_pipeline: wait for the fit");
        if (rc != HSR_OK) return rc;
      }
      job.x_dev = old->d.pseudo_dev;
      job.out_dev = old->d.matched_dev;
      job.coeffs_dev = old->d.coeffs_dev;
      job.mask_dev = old->d.apply_mask ? prev_mask_dev : nullptr;
      job.npix = old->d.npix;
      job.clip = old->d.clip;
    }
    hsr_step_plan* last = (tail && pl->n >= 1) ? pl->slot[(pl->n + 2) % 3] : nullptr;     // tile n - 1: its fit rides in this launch's tail
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
    if (carry && tail && !old->fitted) {         // its fit did not ride in the previous launch (see above): a launch of its own
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
      pl->tickets += (unsigned int)grid;         // every workgroup of the launch drew one ticket
      last->fitted = true;
    }
    if (k1_end_event) rc = hsr::check_hip(hipEventRecord((hipEvent_t)k1_end_event, main), "hsr_pipeline: record K1 end");
    if (rc != HSR_OK) return rc;
    if (carry) {
      old->pending = false;
      if (finished_slot) *finished_slot = (int)((pl->n + 1) % 3);
    }
  } else {
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
  }
  if (S == 3 && !pl->exchange) {        // tail fits: nothing on the side stream, no events
    p->pending = true;
    p->fitted = false;
    pl->n += 1;
    return HSR_OK;
  }
  rc = hsr::check_hip(hipEventRecord(p->ev_k1, main), "hsr_pipeline: record K1");
  if (rc != HSR_OK) return rc;
  rc = hsr::check_hip(hipStreamWaitEvent(pl->side, p->ev_k1, 0), "hsr_pipeline: side stream wait");
  if (rc != HSR_OK) return rc;
  if (S == 3) hipLaunchKernelGGL(side_delay_kernel, dim3(1), dim3(64), 0, pl->side, 3);
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

// exchange = 1: the caller has enqueued the fit of the slot submitted last on the side stream.
extern "C" int hsr_pipeline_fit_done(hsr_pipeline* pl) {
  HSR_REQUIRE(pl && pl->n > 0, HSR_ERR_INVALID, "hsr_pipeline_fit_done: nothing submitted");
  hsr_step_plan* p = pl->slot[(pl->n - 1) % pl->nslots];
  return hsr::check_hip(hipEventRecord(p->ev_fit, pl->side), "hsr_pipeline: record fit");
}

// K3 of the tile left in the pipeline; *finished_slot = its slot or -1.
extern "C" int hsr_pipeline_flush(hsr_pipeline* pl, const uint8_t* mask_dev, hsr_stream_t main_stream, int32_t* finished_slot) {
  HSR_REQUIRE(pl, HSR_ERR_INVALID, "hsr_pipeline_flush: NULL pipeline");
  if (finished_slot) *finished_slot = -1;
  const int S = pl->nslots;
  for (int64_t i = pl->n >= S - 1 ? pl->n - (S - 1) : 0; i < pl->n; ++i) {        // the OLDEST unfinished tile
    hsr_step_plan* p = pl->slot[i % S];
    if (!p->pending) continue;
    int rc = finish_slot(pl, p, mask_dev, (hipStream_t)main_stream);
    if (rc == HSR_OK && finished_slot) *finished_slot = (int)(i % S);
    return rc;
  }
  return HSR_OK;
}

extern "C" int64_t hsr_pipeline_count(const hsr_pipeline* pl) { return pl ? pl->n : -1; }
