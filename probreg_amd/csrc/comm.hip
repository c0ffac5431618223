// The one collective of the CPD hot path: a SUM all-reduce of the fp64 moment block per EM iteration
// (rigid / affine: 32 doubles; non-rigid: the per-point block as well), issued by the library itself on the plan's
// stream through RCCL - an EM iteration is then enqueue-only on every rank: no Python, no torch.distributed, no stream
// switch between the E-step's last kernel and the M-step.
//
// librccl is bound at run time (dlopen): a single-GPU user never loads it, and a process that already carries an RCCL
// (PyTorch ships its own librccl.so.1) keeps using that one instead of getting a second copy.  The handful of RCCL
// declarations needed are restated here (rccl.h: ncclUniqueId 128 bytes, ncclFloat64 = 8, ncclSum = 0).
#include <dlfcn.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>

#include "cpd_plan.h"

namespace {

struct RcclApi {
    void* lib = nullptr;
    int (*GetUniqueId)(void* id) = nullptr;
    int (*CommInitRank)(void** comm, int nranks, prg::UniqueId id, int rank) = nullptr;
    int (*CommDestroy)(void* comm) = nullptr;
    int (*AllReduce)(const void* send, void* recv, size_t count, int dtype, int op, void* comm, hipStream_t st) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    int (*GetVersion)(int*) = nullptr;
    int (*CommCount)(void* comm, int* count) = nullptr;
    int (*CommUserRank)(void* comm, int* rank) = nullptr;
    bool ok = false;
    char why[256] = {0};
};

RcclApi g_rccl;
std::once_flag g_rccl_once;

void load_rccl() {
    RcclApi& a = g_rccl;
    const char* env = getenv("PRG_RCCL_LIB");
    const char* names[] = {env, "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
    // an RCCL the process already carries (PyTorch's) first
    for (const char* n : names)
        if (n && !a.lib) a.lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);
    for (const char* n : names)
        if (n && !a.lib) a.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
    if (!a.lib) {
        snprintf(a.why, sizeof(a.why), "librccl not found (%s); set PRG_RCCL_LIB", dlerror());
        return;
    }
    auto sym = [&](const char* n) { return dlsym(a.lib, n); };
    a.GetUniqueId = reinterpret_cast<decltype(a.GetUniqueId)>(sym("ncclGetUniqueId"));
    a.CommInitRank = reinterpret_cast<decltype(a.CommInitRank)>(sym("ncclCommInitRank"));
    a.CommDestroy = reinterpret_cast<decltype(a.CommDestroy)>(sym("ncclCommDestroy"));
    a.AllReduce = reinterpret_cast<decltype(a.AllReduce)>(sym("ncclAllReduce"));
    a.GetErrorString = reinterpret_cast<decltype(a.GetErrorString)>(sym("ncclGetErrorString"));
    a.GetVersion = reinterpret_cast<decltype(a.GetVersion)>(sym("ncclGetVersion"));
    a.CommCount = reinterpret_cast<decltype(a.CommCount)>(sym("ncclCommCount"));
    a.CommUserRank = reinterpret_cast<decltype(a.CommUserRank)>(sym("ncclCommUserRank"));
    if (!a.GetUniqueId || !a.CommInitRank || !a.CommDestroy || !a.AllReduce || !a.GetErrorString) {
        snprintf(a.why, sizeof(a.why), "librccl lacks one of ncclGetUniqueId / ncclCommInitRank / ncclCommDestroy / ncclAllReduce");
        return;
    }
    a.ok = true;
}

int need_rccl() {
    std::call_once(g_rccl_once, load_rccl);
    PRG_REQUIRE(g_rccl.ok, PRG_ERR_STATE, "RCCL unavailable: %s", g_rccl.why);
    return PRG_OK;
}

#define PRG_NCCL(expr)                                                                                         \
    do {                                                                                                       \
        int _r = (expr);                                                                                       \
        if (_r != 0) {                                                                                         \
            prg::set_error("%s failed: %s (%s:%d)", #expr, g_rccl.GetErrorString(_r), __FILE__, __LINE__);     \
            return PRG_ERR_HIP;                                                                                \
        }                                                                                                      \
    } while (0)

constexpr int kNcclFloat64 = 8, kNcclSum = 0;

}  // namespace

namespace prg {

int comm_all_reduce_f64(prg_comm* c, double* buf_dev, int64_t count, hipStream_t st) {
    PRG_REQUIRE(c && c->nccl, PRG_ERR_INVALID, "prg_comm_all_reduce_f64: NULL communicator");
    if (count <= 0) return PRG_OK;
    PRG_NCCL(g_rccl.AllReduce(buf_dev, buf_dev, (size_t)count, kNcclFloat64, kNcclSum, c->nccl, st));
    ++c->calls;
    return PRG_OK;
}

}  // namespace prg

extern "C" {

int prg_comm_available(int* version) {
    if (version) *version = 0;
    PRG_TRY(need_rccl());
    if (version && g_rccl.GetVersion) (void)g_rccl.GetVersion(version);
    return PRG_OK;
}

int prg_comm_unique_id(unsigned char* id_out) {
    PRG_REQUIRE(id_out, PRG_ERR_INVALID, "prg_comm_unique_id: NULL argument");
    PRG_TRY(need_rccl());
    prg::UniqueId id;
    memset(&id, 0, sizeof(id));
    PRG_NCCL(g_rccl.GetUniqueId(&id));
    memcpy(id_out, id.bytes, PRG_COMM_ID_BYTES);
    return PRG_OK;
}

int prg_comm_create(prg_comm** out, const unsigned char* id_in, int rank, int nranks, int device) {
    PRG_REQUIRE(out && id_in, PRG_ERR_INVALID, "prg_comm_create: NULL argument");
    PRG_REQUIRE(nranks >= 1 && rank >= 0 && rank < nranks, PRG_ERR_INVALID, "prg_comm_create: rank %d outside a world of %d", rank, nranks);
    PRG_TRY(need_rccl());
    prg::DeviceGuard g(device);
    PRG_REQUIRE(g.ok, PRG_ERR_HIP, "prg_comm_create: cannot select device %d", device);
    prg::UniqueId id;
    memcpy(id.bytes, id_in, PRG_COMM_ID_BYTES);
    void* comm = nullptr;
    PRG_NCCL(g_rccl.CommInitRank(&comm, nranks, id, rank));
    prg_comm* c = new prg_comm;
    c->nccl = comm;
    c->rank = rank;
    c->nranks = nranks;
    c->device = device;
    c->owned = true;
    *out = c;
    return PRG_OK;
}

int prg_comm_adopt(prg_comm** out, void* nccl_comm, int device) {
    PRG_REQUIRE(out && nccl_comm, PRG_ERR_INVALID, "prg_comm_adopt: NULL argument");
    PRG_TRY(need_rccl());
    prg_comm* c = new prg_comm;
    c->nccl = nccl_comm;
    c->device = device;
    c->owned = false;
    if (g_rccl.CommCount) (void)g_rccl.CommCount(nccl_comm, &c->nranks);
    if (g_rccl.CommUserRank) (void)g_rccl.CommUserRank(nccl_comm, &c->rank);
    *out = c;
    return PRG_OK;
}

int prg_comm_info(prg_comm* c, int* rank, int* nranks, int64_t* calls) {
    PRG_REQUIRE(c, PRG_ERR_INVALID, "prg_comm_info: NULL communicator");
    if (rank) *rank = c->rank;
    if (nranks) *nranks = c->nranks;
    if (calls) *calls = c->calls;
    return PRG_OK;
}

int prg_comm_destroy(prg_comm* c) {
    if (!c) return PRG_OK;
    int st = PRG_OK;
    if (c->owned && c->nccl && g_rccl.ok) {
        prg::DeviceGuard g(c->device);
        const int r = g_rccl.CommDestroy(c->nccl);
        if (r != 0) {
            prg::set_error("ncclCommDestroy failed: %s", g_rccl.GetErrorString(r));
            st = PRG_ERR_HIP;
        }
    }
    delete c;
    return st;
}

int prg_comm_all_reduce_f64(prg_comm* c, double* buf_dev, int64_t count, void* hip_stream) {
    PRG_REQUIRE(c && buf_dev, PRG_ERR_INVALID, "prg_comm_all_reduce_f64: NULL argument");
    prg::DeviceGuard g(c->device);
    return prg::comm_all_reduce_f64(c, buf_dev, count, reinterpret_cast<hipStream_t>(hip_stream));
}

}  // extern "C"
