// Entropic OT sub-step of fit_ot_poly_rgb / ot_match_rgb_sinkhorn_pot on the device.
//
// Reference call sites: s2_emit/poly_regression.py:49-56 and s2_emit/color.py:97-104
//   M = ot.dist(X, Y, metric="sqeuclidean");  P = ot.sinkhorn(a, b, M, reg, numItermax, stopThr)
//   Ybar = (P @ Y) / (P.sum(axis=1, keepdims=True) + 1e-32)
// POT is an unpinned dependency that is absent offline: PARITY UNPINNED.  The kernels follow POT's published
// sinkhorn_knopp (K = exp(-M/reg); v <- b / (K^T u); u <- a / (K v); breakdown test every iteration, error
// ||v * (K^T u) - b||_2 every 10th) and are checked against the NumPy restatement in oracle/ and OT invariants.
//
// Shape of the problem: n = m = 5000 samples of 3 channels, float64 -> K is 200 MB, touched twice per
// iteration by two mat-vecs of 0.25 flop/B: HBM-bound streaming (and K fits the 256 MB Infinity Cache).
//   * row pass    u_i = a_i / sum_j K_ij v_j     one wave per row, 16-byte loads, fixed xor-butterfly
//   * column pass (K^T u)_j                      workgroup = 512 columns x 64 rows, coalesced along j; the
//                                                per-chunk partial sums are added in a fixed order by a second
//                                                small kernel -> no float atomics, bitwise reproducible
// The iteration state (breakdown / convergence iteration) lives on the device and every kernel of a later
// iteration returns at once, so a whole Sinkhorn solve is enqueued without a single host synchronisation.
#include <math.h>

#include "hsr_common.h"

namespace hsr {

constexpr int kOtColsPerBlock = 512;   // 256 threads x double2
constexpr int kOtRowsPerChunk = 64;
constexpr int32_t kOtNever = 0x7fffffff;

struct OtState {
  int32_t break_iter;   // first iteration with a numerical breakdown (its update is discarded), kOtNever if none
  int32_t conv_iter;    // first checked iteration with err < stopThr, kOtNever if none
  int32_t checks;       // error evaluations done
  int32_t pad;
  double err;           // last evaluated error
};

struct OtWork {
  double* K;         // [n][m]
  double* partial;   // [nchunks][m]
  double* errpart;   // [ceil(m / 64)] squared-residual sums of the error check
  double* u[2];      // [n]
  double* v[2];      // [m]
  OtState* state;
  int64_t n, m;
  int32_t nchunks;
};

static size_t ot_align(size_t x) { return (x + 255) & ~(size_t)255; }

static size_t ot_layout(int64_t n, int64_t m, OtWork* w, unsigned char* base) {
  const int32_t nchunks = (int32_t)((n + kOtRowsPerChunk - 1) / kOtRowsPerChunk);
  size_t off = 0;
  auto take = [&](size_t bytes) {
    unsigned char* p = base ? base + off : nullptr;
    off += ot_align(bytes);
    return p;
  };
  unsigned char* st = take(sizeof(OtState));
  unsigned char* K = take((size_t)n * m * 8);
  unsigned char* pa = take((size_t)nchunks * m * 8);
  unsigned char* ep = take((size_t)((m + 63) / 64) * 8);
  unsigned char* u0 = take((size_t)n * 8);
  unsigned char* u1 = take((size_t)n * 8);
  unsigned char* v0 = take((size_t)m * 8);
  unsigned char* v1 = take((size_t)m * 8);
  if (w) {
    w->state = (OtState*)st;
    w->K = (double*)K;
    w->partial = (double*)pa;
    w->errpart = (double*)ep;
    w->u[0] = (double*)u0;
    w->u[1] = (double*)u1;
    w->v[0] = (double*)v0;
    w->v[1] = (double*)v1;
    w->n = n;
    w->m = m;
    w->nchunks = nchunks;
  }
  return off;
}

__device__ __forceinline__ bool ot_stopped(const OtState* s, int32_t ii) {
  return s->break_iter < ii || s->conv_iter < ii;
}

// K_ij = exp(max(|x_i|^2 + |y_j|^2 - 2 x_i.y_j, 0) / -reg); u = 1/n, v = 1/m; state reset
__global__ __launch_bounds__(256) void ot_init_kernel(OtWork w, const double* __restrict__ X, const double* __restrict__ Y,
                                                      double reg) {
  const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t i0 = (int64_t)blockIdx.y * kOtRowsPerChunk;
  if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {
    w.state->break_iter = kOtNever;
    w.state->conv_iter = kOtNever;
    w.state->checks = 0;
    w.state->err = HUGE_VAL;
  }
  if (blockIdx.y == 0 && j < w.m) w.v[0][j] = 1.0 / (double)w.m;
  if (blockIdx.x == 0 && threadIdx.x < kOtRowsPerChunk && i0 + threadIdx.x < w.n) w.u[0][i0 + threadIdx.x] = 1.0 / (double)w.n;
  if (j >= w.m) return;
  const double y0 = Y[j * 3], y1 = Y[j * 3 + 1], y2 = Y[j * 3 + 2];
  const double b2 = y0 * y0 + y1 * y1 + y2 * y2;
  for (int r = 0; r < kOtRowsPerChunk; ++r) {
    const int64_t i = i0 + r;
    if (i >= w.n) break;
    const double x0 = X[i * 3], x1 = X[i * 3 + 1], x2 = X[i * 3 + 2];
    const double a2 = x0 * x0 + x1 * x1 + x2 * x2;
    double d = a2 + b2 - 2.0 * (x0 * y0 + x1 * y1 + x2 * y2);
    d = d > 0.0 ? d : 0.0;
    w.K[i * w.m + j] = exp(d / (-reg));
  }
}

// partial[chunk][j] = sum_{i in chunk} K_ij u_i   (rows added in index order)
__global__ __launch_bounds__(256) void ot_col_partial_kernel(OtWork w, int32_t ii, int32_t ubuf) {
  if (ot_stopped(w.state, ii)) return;
  __shared__ double us[kOtRowsPerChunk];
  const int64_t i0 = (int64_t)blockIdx.y * kOtRowsPerChunk;
  const int rows = (int)((w.n - i0) < kOtRowsPerChunk ? (w.n - i0) : kOtRowsPerChunk);
  if (threadIdx.x < kOtRowsPerChunk) us[threadIdx.x] = threadIdx.x < rows ? w.u[ubuf][i0 + threadIdx.x] : 0.0;
  __syncthreads();
  const int64_t j = (int64_t)blockIdx.x * kOtColsPerBlock + 2 * threadIdx.x;
  if (j >= w.m) return;
  const double* kp = w.K + i0 * w.m + j;
  double s0 = 0.0, s1 = 0.0;
  if (j + 1 < w.m && (w.m & 1) == 0) {
    constexpr int U = 8;
    int r = 0;
    for (; r + U <= rows; r += U) {
      double2 kv[U];
#pragma unroll
      for (int q = 0; q < U; ++q) kv[q] = *reinterpret_cast<const double2*>(kp + (int64_t)(r + q) * w.m);
#pragma unroll
      for (int q = 0; q < U; ++q) {
        s0 += kv[q].x * us[r + q];
        s1 += kv[q].y * us[r + q];
      }
    }
    for (; r < rows; ++r) {
      const double2 kv = *reinterpret_cast<const double2*>(kp + (int64_t)r * w.m);
      s0 += kv.x * us[r];
      s1 += kv.y * us[r];
    }
  } else {
    for (int r = 0; r < rows; ++r) {
      s0 += kp[(int64_t)r * w.m] * us[r];
      if (j + 1 < w.m) s1 += kp[(int64_t)r * w.m + 1] * us[r];
    }
  }
  double* pp = w.partial + (int64_t)blockIdx.y * w.m + j;
  pp[0] = s0;
  if (j + 1 < w.m) pp[1] = s1;
}

// (K^T u)_j = sum of the chunk partials in a fixed order: a workgroup owns 64 columns, its 4 waves each add a
// contiguous quarter of the chunks, the quarters are combined through LDS in wave order.
__device__ __forceinline__ double ot_col_sum(const OtWork& w, int64_t j, double (*quart)[64]) {
  const int q = threadIdx.x >> 6, l = threadIdx.x & 63;
  const int per = (w.nchunks + 3) >> 2;
  const int c0 = q * per, c1 = (c0 + per) < w.nchunks ? (c0 + per) : w.nchunks;
  double s = 0.0;
  if (j < w.m)
    for (int c = c0; c < c1; ++c) s += w.partial[(int64_t)c * w.m + j];
  quart[q][l] = s;
  __syncthreads();
  return ((quart[0][l] + quart[1][l]) + quart[2][l]) + quart[3][l];
}

// v_new_j = b_j / (K^T u)_j; breakdown flags
__global__ __launch_bounds__(256) void ot_col_finish_kernel(OtWork w, int32_t ii, int32_t vbuf_out, double bval) {
  if (ot_stopped(w.state, ii)) return;
  __shared__ double quart[4][64];
  const int64_t j = (int64_t)blockIdx.x * 64 + (threadIdx.x & 63);
  const double s = ot_col_sum(w, j, quart);
  if (threadIdx.x >= 64 || j >= w.m) return;
  const double v = bval / s;
  w.v[vbuf_out][j] = v;
  if (s == 0.0 || !isfinite(v)) atomicMin(&w.state->break_iter, ii);
}

// squared residual of the marginal, per workgroup of 64 columns: errpart[blk] = sum_j (v_j (K^T u)_j - b_j)^2
__global__ __launch_bounds__(256) void ot_error_partial_kernel(OtWork w, int32_t ii, int32_t vbuf, double bval) {
  if (ot_stopped(w.state, ii) || w.state->break_iter <= ii) return;
  __shared__ double quart[4][64];
  const int64_t j = (int64_t)blockIdx.x * 64 + (threadIdx.x & 63);
  const double s = ot_col_sum(w, j, quart);
  if (threadIdx.x >= 64) return;
  double r = j < w.m ? w.v[vbuf][j] * s - bval : 0.0;
  r = wave_sum(r * r);
  if (threadIdx.x == 0) w.errpart[blockIdx.x] = r;
}

// u_new_i = a_i / sum_j K_ij v_j : one wave per row
__global__ __launch_bounds__(256) void ot_row_kernel(OtWork w, int32_t ii, int32_t vbuf, int32_t ubuf_out, double aval) {
  if (ot_stopped(w.state, ii)) return;
  const int lane = threadIdx.x & 63;
  const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= w.n) return;
  const double* kr = w.K + i * w.m;
  const double* vv = w.v[vbuf];
  double s = 0.0;
  if ((w.m & 1) == 0) {
    const int64_t m2 = w.m >> 1;
    constexpr int U = 4;
    int64_t c = lane;
    for (; c + (U - 1) * 64 < m2; c += U * 64) {
      double2 kv[U], xv[U];
#pragma unroll
      for (int q = 0; q < U; ++q) {
        kv[q] = reinterpret_cast<const double2*>(kr)[c + q * 64];
        xv[q] = reinterpret_cast<const double2*>(vv)[c + q * 64];
      }
#pragma unroll
      for (int q = 0; q < U; ++q) s += kv[q].x * xv[q].x + kv[q].y * xv[q].y;
    }
    for (; c < m2; c += 64) {
      const double2 kv = reinterpret_cast<const double2*>(kr)[c];
      const double2 xv = reinterpret_cast<const double2*>(vv)[c];
      s += kv.x * xv.x + kv.y * xv.y;
    }
  } else {
    for (int64_t j = lane; j < w.m; j += 64) s += kr[j] * vv[j];
  }
  s = wave_sum(s);
  if (lane == 0) {
    const double u = aval / s;
    w.u[ubuf_out][i] = u;
    if (!isfinite(u)) atomicMin(&w.state->break_iter, ii);
  }
}

// err = || v * (K^T u) - b ||_2: one wave adds the per-workgroup sums in index order
__global__ __launch_bounds__(64) void ot_error_kernel(OtWork w, int32_t ii, int32_t nparts, double stop_thr) {
  if (ot_stopped(w.state, ii) || w.state->break_iter <= ii) return;
  double acc = 0.0;
  for (int p = threadIdx.x; p < nparts; p += 64) acc += w.errpart[p];
  acc = wave_sum(acc);
  if (threadIdx.x == 0) {
    const double err = sqrt(acc);
    w.state->err = err;
    w.state->checks += 1;
    if (err < stop_thr) w.state->conv_iter = ii;
  }
}

// Ybar_i = (sum_j P_ij Y_j) / (sum_j P_ij + 1e-32), P_ij = (u_i K_ij) v_j, with the (u, v) the loop ended on
__global__ __launch_bounds__(256) void ot_barycentric_kernel(OtWork w, const double* __restrict__ Y, int32_t iters,
                                                             double* __restrict__ ybar) {
  const OtState* st = w.state;
  // breakdown at ii: the values before that iteration; convergence at ii / running out: after the last update
  int buf;
  if (st->break_iter != kOtNever && st->break_iter <= st->conv_iter) buf = st->break_iter & 1;
  else if (st->conv_iter != kOtNever) buf = (st->conv_iter + 1) & 1;
  else buf = iters & 1;
  const int lane = threadIdx.x & 63;
  const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= w.n) return;
  const double ui = w.u[buf][i];
  const double* kr = w.K + i * w.m;
  const double* vv = w.v[buf];
  double s = 0.0, t0 = 0.0, t1 = 0.0, t2 = 0.0;
  for (int64_t j = lane; j < w.m; j += 64) {
    const double p = (ui * kr[j]) * vv[j];
    s += p;
    t0 += p * Y[j * 3];
    t1 += p * Y[j * 3 + 1];
    t2 += p * Y[j * 3 + 2];
  }
  s = wave_sum(s);
  t0 = wave_sum(t0);
  t1 = wave_sum(t1);
  t2 = wave_sum(t2);
  if (lane == 0) {
    const double den = s + 1e-32;
    ybar[i * 3] = t0 / den;
    ybar[i * 3 + 1] = t1 / den;
    ybar[i * 3 + 2] = t2 / den;
  }
}

}  // namespace hsr

extern "C" int64_t hsr_ot_work_bytes(int64_t n, int64_t m) {
  if (n < 1 || m < 1) return 0;
  return (int64_t)hsr::ot_layout(n, m, nullptr, nullptr);
}

// The solve in three stages, for callers that want to look at the state between blocks of iterations (one host
// synchronisation per look) instead of enqueueing all numItermax iterations: after convergence the remaining
// launches return at once, but ~800 empty launches still cost ~2 ms.
static int ot_prepare(hsr::OtWork& w, int64_t n, int64_t m, void* work_dev, const char* who) {
  using namespace hsr;
  HSR_REQUIRE(work_dev, HSR_ERR_INVALID, "%s: NULL workspace", who);
  HSR_REQUIRE(n >= 1 && m >= 1 && n <= (1 << 20) && m <= (1 << 20), HSR_ERR_UNSUPPORTED, "%s: n=%lld m=%lld outside [1, 2^20]",
              who, (long long)n, (long long)m);
  HSR_REQUIRE(((uintptr_t)work_dev & 255) == 0, HSR_ERR_INVALID, "%s: workspace not 256-byte aligned", who);
  ot_layout(n, m, &w, (unsigned char*)work_dev);
  return HSR_OK;
}

extern "C" int hsr_ot_begin(const double* x_dev, int64_t n, const double* y_dev, int64_t m, double reg, void* work_dev,
                            hsr_stream_t stream) {
  using namespace hsr;
  HSR_REQUIRE(x_dev && y_dev, HSR_ERR_INVALID, "hsr_ot_begin: NULL pointer");
  HSR_REQUIRE(reg > 0.0, HSR_ERR_INVALID, "hsr_ot_begin: reg must be > 0");
  OtWork w{};
  int rc = ot_prepare(w, n, m, work_dev, "hsr_ot_begin");
  if (rc != HSR_OK) return rc;
  hipLaunchKernelGGL(ot_init_kernel, dim3((unsigned)((m + 255) / 256), (unsigned)w.nchunks), dim3(256), 0, (hipStream_t)stream,
                     w, x_dev, y_dev, reg);
  HSR_LAUNCH_CHECK("ot_init_kernel");
  return HSR_OK;
}

extern "C" int hsr_ot_iterate(int64_t n, int64_t m, int32_t first_iter, int32_t count, double stop_thr, void* work_dev,
                              int32_t* info_dev, hsr_stream_t stream) {
  using namespace hsr;
  HSR_REQUIRE(first_iter >= 0 && count >= 0, HSR_ERR_INVALID, "hsr_ot_iterate: bad iteration range");
  OtWork w{};
  int rc = ot_prepare(w, n, m, work_dev, "hsr_ot_iterate");
  if (rc != HSR_OK) return rc;
  hipStream_t s = (hipStream_t)stream;
  const double aval = 1.0 / (double)n, bval = 1.0 / (double)m;   // uniform marginals, as the reference builds them
  const dim3 gcol((unsigned)((m + kOtColsPerBlock - 1) / kOtColsPerBlock), (unsigned)w.nchunks);
  const unsigned gfin = (unsigned)((m + 63) / 64), grow = (unsigned)((n + 3) / 4);
  for (int32_t ii = first_iter; ii < first_iter + count; ++ii) {
    const int cur = ii & 1, nxt = cur ^ 1;
    hipLaunchKernelGGL(ot_col_partial_kernel, gcol, dim3(256), 0, s, w, ii, cur);
    hipLaunchKernelGGL(ot_col_finish_kernel, dim3(gfin), dim3(256), 0, s, w, ii, nxt, bval);
    hipLaunchKernelGGL(ot_row_kernel, dim3(grow), dim3(256), 0, s, w, ii, nxt, nxt, aval);
    if (ii % 10 == 0) {   // POT's schedule: error every 10th iteration, with the updated u and v
      hipLaunchKernelGGL(ot_col_partial_kernel, gcol, dim3(256), 0, s, w, ii, nxt);
      hipLaunchKernelGGL(ot_error_partial_kernel, dim3(gfin), dim3(256), 0, s, w, ii, nxt, bval);
      hipLaunchKernelGGL(ot_error_kernel, dim3(1), dim3(64), 0, s, w, ii, (int32_t)gfin, stop_thr);
    }
  }
  HSR_LAUNCH_CHECK("ot sinkhorn iterations");
  if (info_dev) {   // {break_iter, conv_iter, checks, pad, err(double)} = 24 bytes
    rc = check_hip(hipMemcpyAsync(info_dev, w.state, sizeof(OtState), hipMemcpyDeviceToDevice, s), "hipMemcpyAsync");
    if (rc != HSR_OK) return rc;
  }
  return HSR_OK;
}

extern "C" int hsr_ot_finish(const double* y_dev, int64_t n, int64_t m, int32_t iterations_done, void* work_dev,
                             double* ybar_dev, int32_t* info_dev, hsr_stream_t stream) {
  using namespace hsr;
  HSR_REQUIRE(y_dev && ybar_dev && iterations_done >= 0, HSR_ERR_INVALID, "hsr_ot_finish: bad argument");
  OtWork w{};
  int rc = ot_prepare(w, n, m, work_dev, "hsr_ot_finish");
  if (rc != HSR_OK) return rc;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(ot_barycentric_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, s, w, y_dev, iterations_done, ybar_dev);
  HSR_LAUNCH_CHECK("ot_barycentric_kernel");
  if (info_dev) {
    rc = check_hip(hipMemcpyAsync(info_dev, w.state, sizeof(OtState), hipMemcpyDeviceToDevice, s), "hipMemcpyAsync");
    if (rc != HSR_OK) return rc;
  }
  return HSR_OK;
}

extern "C" int hsr_ot_sinkhorn_barycentric(const double* x_dev, int64_t n, const double* y_dev, int64_t m, double reg,
                                           int32_t num_iter_max, double stop_thr, void* work_dev, double* ybar_dev,
                                           int32_t* info_dev, hsr_stream_t stream) {
  HSR_REQUIRE(num_iter_max >= 0, HSR_ERR_INVALID, "hsr_ot_sinkhorn_barycentric: numItermax must be >= 0");
  int rc = hsr_ot_begin(x_dev, n, y_dev, m, reg, work_dev, stream);
  if (rc == HSR_OK) rc = hsr_ot_iterate(n, m, 0, num_iter_max, stop_thr, work_dev, nullptr, stream);
  if (rc == HSR_OK) rc = hsr_ot_finish(y_dev, n, m, num_iter_max, work_dev, ybar_dev, info_dev, stream);
  return rc;
}
