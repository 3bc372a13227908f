// Allocation kinds for tools/placement_lab.py ("try allocations the search cannot").
// Diagnostic helper, not part of libhsr.  hipcc -shared -fPIC -o tools/libplacement_alloc.so tools/placement_alloc.hip
//
//   kind 0  hipMalloc                                   (what torch's caching allocator calls underneath)
//   kind 1  hipExtMallocWithFlags(hipDeviceMallocContiguous)   physically contiguous backing
//   kind 2  HIP VMM: ONE physical handle of the whole size (hipMemCreate), one hipMemMap
//   kind 3  HIP VMM: one physical handle per `chunk` bytes, mapped back to back into one VA range
//   kind 4  hipExtMallocWithFlags(hipDeviceMallocUncached)      (fine-grained / uncached MTYPE)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

struct lab_block {
    void* ptr;
    size_t bytes;
    int kind;
    std::vector<hipMemGenericAllocationHandle_t>* handles;
};

static char g_err[256];

extern "C" const char* lab_last_error() { return g_err; }

static bool ok(hipError_t e, const char* what) {
    if (e == hipSuccess) return true;
    snprintf(g_err, sizeof g_err, "%s: %s", what, hipGetErrorString(e));
    (void)hipGetLastError();
    return false;
}

extern "C" size_t lab_granularity(int device, int recommended) {
    hipMemAllocationProp prop;
    memset(&prop, 0, sizeof prop);
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = device;
    size_t g = 0;
    if (!ok(hipMemGetAllocationGranularity(&g, &prop, recommended ? hipMemAllocationGranularityRecommended
                                                                  : hipMemAllocationGranularityMinimum), "granularity"))
        return 0;
    return g;
}

extern "C" lab_block* lab_alloc(int kind, size_t bytes, size_t chunk, int device) {
    g_err[0] = 0;
    lab_block* b = new lab_block{nullptr, bytes, kind, nullptr};
    if (kind == 0) {
        if (!ok(hipMalloc(&b->ptr, bytes), "hipMalloc")) { delete b; return nullptr; }
        return b;
    }
    if (kind == 1 || kind == 4) {
        unsigned flags = kind == 1 ? hipDeviceMallocContiguous : hipDeviceMallocUncached;
        if (!ok(hipExtMallocWithFlags(&b->ptr, bytes, flags), "hipExtMallocWithFlags")) { delete b; return nullptr; }
        return b;
    }
    hipMemAllocationProp prop;
    memset(&prop, 0, sizeof prop);
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = device;
    size_t gran = lab_granularity(device, 1);
    if (!gran) { delete b; return nullptr; }
    if (kind == 2) chunk = 0;
    size_t piece = chunk ? (chunk + gran - 1) / gran * gran : (bytes + gran - 1) / gran * gran;
    size_t total = (bytes + piece - 1) / piece * piece;
    b->bytes = total;
    if (!ok(hipMemAddressReserve(&b->ptr, total, 0, nullptr, 0), "hipMemAddressReserve")) { delete b; return nullptr; }
    b->handles = new std::vector<hipMemGenericAllocationHandle_t>();
    for (size_t off = 0; off < total; off += piece) {
        hipMemGenericAllocationHandle_t h;
        if (!ok(hipMemCreate(&h, piece, &prop, 0), "hipMemCreate")) return nullptr;       // leaks on failure: lab tool
        b->handles->push_back(h);
        if (!ok(hipMemMap((char*)b->ptr + off, piece, 0, h, 0), "hipMemMap")) return nullptr;
    }
    hipMemAccessDesc acc;
    memset(&acc, 0, sizeof acc);
    acc.location.type = hipMemLocationTypeDevice;
    acc.location.id = device;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    if (!ok(hipMemSetAccess(b->ptr, total, &acc, 1), "hipMemSetAccess")) return nullptr;
    return b;
}

extern "C" void* lab_ptr(lab_block* b) { return b ? b->ptr : nullptr; }
extern "C" size_t lab_bytes(lab_block* b) { return b ? b->bytes : 0; }

extern "C" int lab_free(lab_block* b) {
    if (!b) return 0;
    if (b->kind == 0 || b->kind == 1 || b->kind == 4) {
        (void)hipFree(b->ptr);
    } else {
        (void)hipMemUnmap(b->ptr, b->bytes);
        for (auto h : *b->handles) (void)hipMemRelease(h);
        (void)hipMemAddressFree(b->ptr, b->bytes);
        delete b->handles;
    }
    delete b;
    return 0;
}
