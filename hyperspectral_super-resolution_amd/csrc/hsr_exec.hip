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
};

struct hsr_pipeline {
  hsr_step_plan* slot[2];
  hipStream_t side;
  int64_t n;                      // tiles submitted
  int exchange;                   // 1: the caller runs the fit (reduce -> collective -> solve) itself between
                                  //    hsr_pipeline_submit and hsr_pipeline_fit_done
};

namespace {

int run_k1(hsr_step_plan* p, const void* cube, const float* real, const uint8_t* mask, hipStream_t s) {
  const hsr_step_desc& d = p->d;
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
extern "C" int hsr_pipeline_create(hsr_step_plan* slot0, hsr_step_plan* slot1, hsr_stream_t side_stream, int32_t exchange,
                                   hsr_pipeline** out) {
  HSR_REQUIRE(slot0 && slot1 && slot0 != slot1 && out, HSR_ERR_INVALID, "hsr_pipeline_create: two distinct plans needed");
  HSR_REQUIRE(side_stream != nullptr, HSR_ERR_INVALID, "hsr_pipeline_create: the side stream must be a real stream, not the default one");
  hsr_pipeline* pl = new (std::nothrow) hsr_pipeline();
  HSR_REQUIRE(pl, HSR_ERR_INVALID, "hsr_pipeline_create: out of host memory");
  pl->slot[0] = slot0;
  pl->slot[1] = slot1;
  pl->side = (hipStream_t)side_stream;
  pl->n = 0;
  pl->exchange = exchange ? 1 : 0;
  slot0->pending = slot1->pending = false;
  *out = pl;
  return HSR_OK;
}

extern "C" void hsr_pipeline_destroy(hsr_pipeline* pl) { delete pl; }

static int finish_slot(hsr_pipeline* pl, hsr_step_plan* p, const uint8_t* mask, hipStream_t main) {
  int rc = hsr::check_hip(hipStreamWaitEvent(main, p->ev_fit, 0), "hsr_pipeline: wait for the fit");
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
  const int cur = (int)(pl->n & 1);
  hsr_step_plan* p = pl->slot[cur];
  hsr_step_plan* prev = pl->slot[cur ^ 1];
  HSR_REQUIRE(!p->pending, HSR_ERR_INVALID, "hsr_pipeline_submit: slot %d still holds an unfinished tile", cur);
  int rc = HSR_OK;
  if (k1_begin_event) rc = hsr::check_hip(hipEventRecord((hipEvent_t)k1_begin_event, main), "hsr_pipeline: record K1 begin");
  if (rc != HSR_OK) return rc;
  rc = run_k1(p, cube_dev, real_dev, mask_dev, main);
  if (rc != HSR_OK) return rc;
  if (k1_end_event) rc = hsr::check_hip(hipEventRecord((hipEvent_t)k1_end_event, main), "hsr_pipeline: record K1 end");
  if (rc != HSR_OK) return rc;
  if (finished_slot) *finished_slot = -1;
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

// exchange = 1: the caller has enqueued the fit of the slot submitted last on the side stream.
extern "C" int hsr_pipeline_fit_done(hsr_pipeline* pl) {
  HSR_REQUIRE(pl && pl->n > 0, HSR_ERR_INVALID, "hsr_pipeline_fit_done: nothing submitted");
  hsr_step_plan* p = pl->slot[(pl->n - 1) & 1];
  return hsr::check_hip(hipEventRecord(p->ev_fit, pl->side), "hsr_pipeline: record fit");
}

// K3 of the tile left in the pipeline; *finished_slot = its slot or -1.
extern "C" int hsr_pipeline_flush(hsr_pipeline* pl, const uint8_t* mask_dev, hsr_stream_t main_stream, int32_t* finished_slot) {
  HSR_REQUIRE(pl, HSR_ERR_INVALID, "hsr_pipeline_flush: NULL pipeline");
  if (finished_slot) *finished_slot = -1;
  if (pl->n == 0) return HSR_OK;
  const int last = (int)((pl->n - 1) & 1);
  hsr_step_plan* p = pl->slot[last];
  if (!p->pending) return HSR_OK;
  int rc = finish_slot(pl, p, mask_dev, (hipStream_t)main_stream);
  if (rc == HSR_OK && finished_slot) *finished_slot = last;
  return rc;
}

extern "C" int64_t hsr_pipeline_count(const hsr_pipeline* pl) { return pl ? pl->n : -1; }
