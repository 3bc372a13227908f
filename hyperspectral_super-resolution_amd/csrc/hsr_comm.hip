// C1 from C: the exchange of the fit over the GPUs of a node through RCCL's C API (hsr_comm_init /
// hsr_allreduce_f64 / hsr_bcast; 8e: all-reduce of nb x (3deg+2) moment doubles, or reduce + broadcast of the
// coefficients).  What is summed is the reference's per-channel polyfit input (s2_emit/poly_regression.py:59-60) over the
// tiles the reference treats as independent units (tiles_helpers/utils.py:223-305).
//
// The step executor (hsr_exec.hip) enqueues these on its side stream, so a pipelined step makes NO torch.distributed
// call (18 us of host time each in round 3) and the caller's stream sees one kernel per tile.
//
// RCCL is bound at run time (dlopen + dlsym), not at link time: the library then loads - and every entry point that
// needs no collective works - on a machine without RCCL, and inside a PyTorch process the communicator lives in the
// librccl.so.1 that torch has already mapped (same soname) instead of a second copy with its own state.
#include <dlfcn.h>
#include <string.h>
#include <rccl/rccl.h>

#include <mutex>
#include <new>

#include "hsr_common.h"

namespace {

struct RcclApi {
  void* handle = nullptr;
  ncclResult_t (*GetVersion)(int*) = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Reduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  bool ok = false;
  char why[256] = {0};
};

RcclApi g_api;
std::once_flag g_once;

template <typename F>
bool bind(void* h, const char* name, F* out) {
  *out = reinterpret_cast<F>(dlsym(h, name));
  return *out != nullptr;
}

void load_rccl() {
  // an already mapped librccl.so.1 first (PyTorch's own copy has that soname), then the ROCm installation's
  const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  void* h = dlopen(names[0], RTLD_NOW | RTLD_NOLOAD);
  for (int i = 0; !h && i < 3; ++i) h = dlopen(names[i], RTLD_NOW | RTLD_GLOBAL);
  if (!h) {
    snprintf(g_api.why, sizeof g_api.why, "librccl.so.1 not loadable: %s", dlerror());
    return;
  }
  g_api.handle = h;
  const bool all = bind(h, "ncclGetVersion", &g_api.GetVersion) && bind(h, "ncclGetUniqueId", &g_api.GetUniqueId) &&
                   bind(h, "ncclCommInitRank", &g_api.CommInitRank) && bind(h, "ncclCommDestroy", &g_api.CommDestroy) &&
                   bind(h, "ncclAllReduce", &g_api.AllReduce) && bind(h, "ncclReduce", &g_api.Reduce) &&
                   bind(h, "ncclBroadcast", &g_api.Broadcast) && bind(h, "ncclGetErrorString", &g_api.GetErrorString);
  if (!all) {
    snprintf(g_api.why, sizeof g_api.why, "librccl.so.1 lacks one of the ncclComm* / ncclAllReduce / ncclReduce / ncclBroadcast symbols");
    return;
  }
  g_api.ok = true;
}

int rccl_ready(const char* who) {
  std::call_once(g_once, load_rccl);
  HSR_REQUIRE(g_api.ok, HSR_ERR_UNSUPPORTED, "%s: RCCL is not available (%s)", who, g_api.why);
  return HSR_OK;
}

int check_nccl(ncclResult_t r, const char* what) {
  if (r == ncclSuccess) return HSR_OK;
  hsr::set_error("%s: RCCL error %d (%s)", what, (int)r, g_api.GetErrorString ? g_api.GetErrorString(r) : "?");
  return HSR_ERR_HIP;
}

}  // namespace

struct hsr_comm {
  ncclComm_t comm;
  int32_t rank, nranks;
};

extern "C" int hsr_comm_available(void) {
  std::call_once(g_once, load_rccl);
  return g_api.ok ? 1 : 0;
}

extern "C" int hsr_comm_version(void) {
  if (rccl_ready("hsr_comm_version") != HSR_OK) return -1;
  int v = -1;
  return g_api.GetVersion(&v) == ncclSuccess ? v : -1;
}

extern "C" int hsr_comm_unique_id(void* id_out) {
  HSR_REQUIRE(id_out, HSR_ERR_INVALID, "hsr_comm_unique_id: NULL argument");
  static_assert(sizeof(ncclUniqueId) == HSR_COMM_ID_BYTES, "hsr.h: HSR_COMM_ID_BYTES");
  int rc = rccl_ready("hsr_comm_unique_id");
  if (rc != HSR_OK) return rc;
  return check_nccl(g_api.GetUniqueId(static_cast<ncclUniqueId*>(id_out)), "ncclGetUniqueId");
}

extern "C" int hsr_comm_init(int32_t rank, int32_t nranks, const void* unique_id, hsr_comm** out) {
  HSR_REQUIRE(unique_id && out, HSR_ERR_INVALID, "hsr_comm_init: NULL argument");
  HSR_REQUIRE(nranks >= 1 && rank >= 0 && rank < nranks, HSR_ERR_INVALID, "hsr_comm_init: rank %d of %d", rank, nranks);
  int rc = rccl_ready("hsr_comm_init");
  if (rc != HSR_OK) return rc;
  hsr_comm* c = new (std::nothrow) hsr_comm();
  HSR_REQUIRE(c, HSR_ERR_INVALID, "hsr_comm_init: out of host memory");
  ncclUniqueId id;
  memcpy(&id, unique_id, sizeof id);
  rc = check_nccl(g_api.CommInitRank(&c->comm, nranks, id, rank), "ncclCommInitRank");     // on the CURRENT device
  if (rc != HSR_OK) {
    delete c;
    return rc;
  }
  c->rank = rank;
  c->nranks = nranks;
  *out = c;
  return HSR_OK;
}

extern "C" int hsr_comm_destroy(hsr_comm* c) {
  if (!c) return HSR_OK;
  int rc = HSR_OK;
  if (g_api.ok) rc = check_nccl(g_api.CommDestroy(c->comm), "ncclCommDestroy");
  delete c;
  return rc;
}

extern "C" int hsr_comm_rank(const hsr_comm* c) { return c ? c->rank : -1; }
extern "C" int hsr_comm_ranks(const hsr_comm* c) { return c ? c->nranks : -1; }

extern "C" int hsr_allreduce_f64(hsr_comm* c, double* buf_dev, int64_t count, hsr_stream_t stream) {
  HSR_REQUIRE(c && buf_dev && count >= 1, HSR_ERR_INVALID, "hsr_allreduce_f64: NULL argument or count < 1");
  return check_nccl(g_api.AllReduce(buf_dev, buf_dev, (size_t)count, ncclFloat64, ncclSum, c->comm, (hipStream_t)stream), "ncclAllReduce");
}

extern "C" int hsr_reduce_f64(hsr_comm* c, double* buf_dev, int64_t count, int32_t root, hsr_stream_t stream) {
  HSR_REQUIRE(c && buf_dev && count >= 1 && root >= 0 && root < c->nranks, HSR_ERR_INVALID, "hsr_reduce_f64: bad argument");
  return check_nccl(g_api.Reduce(buf_dev, buf_dev, (size_t)count, ncclFloat64, ncclSum, root, c->comm, (hipStream_t)stream), "ncclReduce");
}

extern "C" int hsr_allreduce_u32(hsr_comm* c, uint32_t* buf_dev, int64_t count, hsr_stream_t stream) {
  HSR_REQUIRE(c && buf_dev && count >= 1, HSR_ERR_INVALID, "hsr_allreduce_u32: NULL argument or count < 1");
  return check_nccl(g_api.AllReduce(buf_dev, buf_dev, (size_t)count, ncclUint32, ncclSum, c->comm, (hipStream_t)stream), "ncclAllReduce");
}

extern "C" int hsr_bcast(hsr_comm* c, void* buf_dev, int64_t bytes, int32_t root, hsr_stream_t stream) {
  HSR_REQUIRE(c && buf_dev && bytes >= 1 && root >= 0 && root < c->nranks, HSR_ERR_INVALID, "hsr_bcast: bad argument");
  return check_nccl(g_api.Broadcast(buf_dev, buf_dev, (size_t)bytes, ncclInt8, root, c->comm, (hipStream_t)stream), "ncclBroadcast");
}
