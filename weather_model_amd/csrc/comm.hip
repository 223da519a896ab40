// comm.hip -- the exchange step of SURVEY.md 8(b)/(e) for hosts without torch.distributed: one RCCL
// communicator per process (one process per GPU) and a sum all-reduce of ONE flat buffer -- the
// float64 partial Gram between wm_gram_f32 and wm_normal_solve_f64, or the flat float32 gradient
// between wm_model_train_f32 and wm_adam_step_f32 -- on the caller's stream.
//
// The reference has no distributed code (a dead MirroredStrategy, singleRegression.py:488-496); the
// data-parallel exchange is new work defined by BASELINE.json.  librccl is opened lazily with
// dlopen, so libwm.so loads on a box without RCCL and a PyTorch host keeps the copy it already
// loaded; weather_model_amd/dp.py (torch.distributed, backend nccl = RCCL) stays the default path.
#include "common.h"

#include <dlfcn.h>
#include <string.h>
#include <mutex>
#include <rccl/rccl.h>       // types and enums only: every function is resolved with dlsym

namespace {

struct Rccl {
    void *handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t,
                              hipStream_t) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
};

std::mutex g_mu;
Rccl g_rccl;
ncclComm_t g_comm = nullptr;
int g_rank = -1, g_nranks = 0, g_dev = -1;

int load_rccl()
{
    if (g_rccl.handle) return WM_OK;
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void *h = nullptr;
    for (const char *n : names)
        if ((h = dlopen(n, RTLD_NOW | RTLD_GLOBAL)) != nullptr) break;
    WM_REQUIRE(h != nullptr, WM_ERR_UNSUPPORTED, "wm_comm: librccl.so not found (%s)", dlerror());
    Rccl r;
    r.handle = h;
    r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(h, "ncclGetUniqueId");
    r.CommInitRank = (decltype(r.CommInitRank))dlsym(h, "ncclCommInitRank");
    r.AllReduce = (decltype(r.AllReduce))dlsym(h, "ncclAllReduce");
    r.CommDestroy = (decltype(r.CommDestroy))dlsym(h, "ncclCommDestroy");
    r.GetErrorString = (decltype(r.GetErrorString))dlsym(h, "ncclGetErrorString");
    WM_REQUIRE(r.GetUniqueId && r.CommInitRank && r.AllReduce && r.CommDestroy && r.GetErrorString,
               WM_ERR_UNSUPPORTED, "wm_comm: librccl.so lacks an expected symbol");
    g_rccl = r;
    return WM_OK;
}

#define WM_RCCL(call)                                                                          \
    do {                                                                                       \
        ncclResult_t r_ = (call);                                                              \
        if (r_ != ncclSuccess) {                                                               \
            wm_set_error("%s failed: %s", #call, g_rccl.GetErrorString(r_));                   \
            return WM_ERR_HIP;                                                                 \
        }                                                                                      \
    } while (0)

}  // namespace

extern "C" int wm_comm_unique_id(void *id128)
{
    WM_REQUIRE(id128 != nullptr, WM_ERR_INVALID, "wm_comm_unique_id: null pointer");
    std::lock_guard<std::mutex> lock(g_mu);
    if (int rc = load_rccl()) return rc;
    static_assert(sizeof(ncclUniqueId) == WM_COMM_ID_BYTES, "wm.h: WM_COMM_ID_BYTES");
    WM_RCCL(g_rccl.GetUniqueId((ncclUniqueId *)id128));
    return WM_OK;
}

extern "C" int wm_comm_init(int rank, int nranks, const void *unique_id)
{
    WM_REQUIRE(unique_id != nullptr, WM_ERR_INVALID, "wm_comm_init: null unique id");
    WM_REQUIRE(nranks >= 1 && rank >= 0 && rank < nranks, WM_ERR_INVALID,
               "wm_comm_init: rank %d of %d", rank, nranks);
    std::lock_guard<std::mutex> lock(g_mu);
    WM_REQUIRE(g_comm == nullptr, WM_ERR_INVALID,
               "wm_comm_init: this process already holds a communicator (wm_comm_destroy first)");
    if (int rc = load_rccl()) return rc;
    int dev = 0;
    WM_HIP(hipGetDevice(&dev));
    ncclUniqueId id;
    memcpy(&id, unique_id, sizeof(id));
    ncclComm_t comm = nullptr;
    WM_RCCL(g_rccl.CommInitRank(&comm, nranks, id, rank));
    g_comm = comm;
    g_rank = rank;
    g_nranks = nranks;
    g_dev = dev;
    return WM_OK;
}

extern "C" int wm_comm_size(void)
{
    std::lock_guard<std::mutex> lock(g_mu);
    return g_comm ? g_nranks : 0;
}

extern "C" int wm_allreduce_sum(void *buf, size_t count, int dtype, void *stream)
{
    WM_REQUIRE(buf != nullptr || count == 0, WM_ERR_INVALID, "wm_allreduce_sum: null buffer");
    WM_REQUIRE(dtype == WM_F32 || dtype == WM_F64, WM_ERR_INVALID,
               "wm_allreduce_sum: dtype %d (WM_F32 or WM_F64)", dtype);
    std::lock_guard<std::mutex> lock(g_mu);
    WM_REQUIRE(g_comm != nullptr, WM_ERR_INVALID, "wm_allreduce_sum: no communicator (wm_comm_init)");
    if (count == 0) return WM_OK;
    int dev = 0;
    WM_HIP(hipGetDevice(&dev));
    WM_REQUIRE(dev == g_dev, WM_ERR_INVALID,
               "wm_allreduce_sum: the communicator belongs to device %d, device %d is current",
               g_dev, dev);
    // one collective on the whole flat buffer, in place, on the caller's stream: at these sizes
    // (<= ~110 KB) the exchange is latency-bound and must not be fragmented
    WM_RCCL(g_rccl.AllReduce(buf, buf, count, dtype == WM_F64 ? ncclFloat64 : ncclFloat32, ncclSum,
                             g_comm, (hipStream_t)stream));
    return WM_OK;
}

extern "C" int wm_comm_destroy(void)
{
    std::lock_guard<std::mutex> lock(g_mu);
    if (!g_comm) return WM_OK;
    ncclComm_t c = g_comm;
    g_comm = nullptr;
    g_rank = -1;
    g_nranks = 0;
    g_dev = -1;
    WM_RCCL(g_rccl.CommDestroy(c));
    return WM_OK;
}
