// Exchange steps of the passage-sharded search for a caller WITHOUT torch: RCCL collectives over a
// communicator the caller owns, issued on the caller's stream, followed by the library's own merge / selection kernels.
// The reference has no counterpart (its multi-GPU runs fall back to the CPU search, src/executors/FLMR_executor.py:778-783).
//
// libflmr_hip.so does NOT link librccl: the five entry points it needs are resolved on first use from the RCCL that is
// already loaded in the process (the one that created the caller's ncclComm_t -- mixing two RCCL instances would not
// work), falling back to a regular dlopen of librccl.so.1.  A process that never calls these functions never loads RCCL.
#include <dlfcn.h>

#include <mutex>

#include "flmr_common.h"

// The handful of RCCL / NCCL ABI names this file needs, declared here so that building libflmr_hip.so does not require the
// RCCL headers (the symbols themselves come from the process at run time).  Values are those of nccl.h (stable since NCCL 2).
typedef struct ncclComm* ncclComm_t;
typedef enum { ncclSuccess = 0 } ncclResult_t;
typedef enum { ncclInt32 = 2, ncclUint64 = 5, ncclFloat32 = 7 } ncclDataType_t;
typedef enum { ncclSum = 0 } ncclRedOp_t;

namespace {
struct rccl_api {
    ncclResult_t (*all_gather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*all_reduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*group_start)() = nullptr;
    ncclResult_t (*group_end)() = nullptr;
    ncclResult_t (*comm_count)(const ncclComm_t, int*) = nullptr;
    const char* (*error_string)(ncclResult_t) = nullptr;
    bool ok = false;
    char why[256] = {0};
};
rccl_api g_rccl;
std::once_flag g_rccl_once;

void load_rccl() {
    void* h = nullptr;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names) {
        h = dlopen(n, RTLD_NOW | RTLD_NOLOAD);   // the instance the caller's communicator came from
        if (h) break;
    }
    for (int i = 0; !h && i < 3; i++) h = dlopen(names[i], RTLD_NOW | RTLD_LOCAL);
    if (!h) {
        snprintf(g_rccl.why, sizeof(g_rccl.why), "librccl.so.1 not found (%s)", dlerror());
        return;
    }
    g_rccl.all_gather = reinterpret_cast<decltype(g_rccl.all_gather)>(dlsym(h, "ncclAllGather"));
    g_rccl.all_reduce = reinterpret_cast<decltype(g_rccl.all_reduce)>(dlsym(h, "ncclAllReduce"));
    g_rccl.group_start = reinterpret_cast<decltype(g_rccl.group_start)>(dlsym(h, "ncclGroupStart"));
    g_rccl.group_end = reinterpret_cast<decltype(g_rccl.group_end)>(dlsym(h, "ncclGroupEnd"));
    g_rccl.comm_count = reinterpret_cast<decltype(g_rccl.comm_count)>(dlsym(h, "ncclCommCount"));
    g_rccl.error_string = reinterpret_cast<decltype(g_rccl.error_string)>(dlsym(h, "ncclGetErrorString"));
    g_rccl.ok = g_rccl.all_gather && g_rccl.all_reduce && g_rccl.group_start && g_rccl.group_end && g_rccl.comm_count &&
                g_rccl.error_string;
    if (!g_rccl.ok) snprintf(g_rccl.why, sizeof(g_rccl.why), "librccl.so.1 lacks an expected symbol");
}

int rccl(const rccl_api** out) {
    std::call_once(g_rccl_once, load_rccl);
    if (!g_rccl.ok) FLMR_FAIL(FLMR_ERR_UNSUPPORTED, "RCCL unavailable: %s", g_rccl.why);
    *out = &g_rccl;
    return FLMR_OK;
}

#define FLMR_RCCL(api, expr)                                                                             \
    do {                                                                                                 \
        ncclResult_t r__ = (expr);                                                                       \
        if (r__ != ncclSuccess) FLMR_FAIL(FLMR_ERR_HIP, "%s -> %s", #expr, (api)->error_string(r__));     \
    } while (0)

int comm_size(const rccl_api* api, void* comm, int32_t nranks) {
    int n = 0;
    FLMR_RCCL(api, api->comm_count(reinterpret_cast<ncclComm_t>(comm), &n));
    if (n != nranks) FLMR_FAIL(FLMR_ERR_INVALID, "communicator has %d ranks, nranks=%d", n, nranks);
    return FLMR_OK;
}
}  // namespace

// fast mode: all-gather of every rank's top-k (score, global pid) lists, merged to the global top-k on every rank
extern "C" int flmr_topk_allgather(void* comm, int32_t nranks, const float* scores, const int32_t* pids, int32_t nqueries,
                                   int32_t k, float* gathered_scores, int32_t* gathered_pids, float* out_scores,
                                   int32_t* out_pids, int32_t* out_counts, flmr_stream_t stream) {
    if (!comm || !scores || !pids || !gathered_scores || !gathered_pids || !out_scores || !out_pids || !out_counts)
        FLMR_FAIL(FLMR_ERR_INVALID, "NULL argument");
    if (nranks < 1 || nqueries < 1 || k < 1) FLMR_FAIL(FLMR_ERR_INVALID, "bad sizes");
    const rccl_api* api = nullptr;
    int rc = rccl(&api);
    if (rc) return rc;
    rc = comm_size(api, comm, nranks);
    if (rc) return rc;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    ncclComm_t c = reinterpret_cast<ncclComm_t>(comm);
    const size_t n = (size_t)nqueries * k;
    // one fused launch for the two arrays (they travel together: 8 bytes per result)
    // (the group is ALWAYS closed: returning between start and end would leave it open and wedge the caller's next collective;
    // the first error is the one reported)
    FLMR_RCCL(api, api->group_start());
    ncclResult_t first = api->all_gather(scores, gathered_scores, n, ncclFloat32, c, st);
    if (first == ncclSuccess) first = api->all_gather(pids, gathered_pids, n, ncclInt32, c, st);
    const ncclResult_t ended = api->group_end();
    if (first == ncclSuccess) first = ended;
    if (first != ncclSuccess) FLMR_FAIL(FLMR_ERR_HIP, "ncclAllGather (top-k exchange) -> %s", api->error_string(first));
    return flmr_merge_topk(gathered_scores, gathered_pids, nranks, nqueries, k, out_scores, out_pids, out_counts, stream);
}

// exact mode, after phase 1: every rank's [count] keys -> out [nranks, count] on every rank (then flmr_select_keys)
extern "C" int flmr_keys_allgather(void* comm, int32_t nranks, const uint64_t* keys, int64_t count, uint64_t* out,
                                   flmr_stream_t stream) {
    if (!comm || !keys || !out) FLMR_FAIL(FLMR_ERR_INVALID, "NULL argument");
    if (nranks < 1 || count < 1) FLMR_FAIL(FLMR_ERR_INVALID, "bad sizes");
    const rccl_api* api = nullptr;
    int rc = rccl(&api);
    if (rc) return rc;
    rc = comm_size(api, comm, nranks);
    if (rc) return rc;
    FLMR_RCCL(api, api->all_gather(keys, out, (size_t)count, ncclUint64, reinterpret_cast<ncclComm_t>(comm),
                                   reinterpret_cast<hipStream_t>(stream)));
    return FLMR_OK;
}

// exact mode, after phases 2 and 3: the slot-aligned key arrays (exactly one non-zero contributor per slot) combine by SUM,
// in place
extern "C" int flmr_keys_allreduce_sum(void* comm, uint64_t* keys, int64_t count, flmr_stream_t stream) {
    if (!comm || !keys) FLMR_FAIL(FLMR_ERR_INVALID, "NULL argument");
    if (count < 1) FLMR_FAIL(FLMR_ERR_INVALID, "bad sizes");
    const rccl_api* api = nullptr;
    const int rc = rccl(&api);
    if (rc) return rc;
    FLMR_RCCL(api, api->all_reduce(keys, keys, (size_t)count, ncclUint64, ncclSum, reinterpret_cast<ncclComm_t>(comm),
                                   reinterpret_cast<hipStream_t>(stream)));
    return FLMR_OK;
}
