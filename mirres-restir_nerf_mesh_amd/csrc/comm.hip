// comm.hip — the per-sample halo exchange of the exact strip scheme issued by mirres_render itself (round 6; SURVEY section 8e primary scheme).
//
// Until round 5 the exchange was a host callback: ctypes -> Python -> torch.distributed.batch_isend_irecv -> return, 512 times per frame. Measured over RCCL on one
// rank (scripts/dev_halo_host_cost.py, profiles/r06_halo_host_cost_python_callback.txt): 56-63 us of host time per callback in the median, 240-330 us in the mean —
// the host enqueues a sample in 260-345 us where the strip's own period is 355 us: the chain of a strip of eight is HOST-bound before any link is. Here the same two
// sends and two receives go out as ncclSend / ncclRecv inside one ncclGroupStart / ncclGroupEnd on the chain's stream, from C, on a communicator of the library's own
// (created from an id the ranks exchange through torch.distributed once). RCCL is not linked: librccl is dlopen-ed (the copy the process already holds — torch's —
// when it can be found, so that the process does not carry two RCCLs), and every symbol is looked up once. gloo (CPU tests) keeps the callback.
#include <dlfcn.h>
#include <cstring>
#include "engine.hpp"

namespace mr {

struct Id128 { char bytes[128]; };      // ncclUniqueId
struct NcclApi {
    void* lib = nullptr;
    int (*GetUniqueId)(void*) = nullptr;
    int (*CommInitRank)(void**, int, Id128 /* by value */, int) = nullptr;
    int (*CommDestroy)(void*) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*Send)(const void*, size_t, int, int, void*, hipStream_t) = nullptr;
    int (*Recv)(void*, size_t, int, int, void*, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
};
static NcclApi g_nccl;

static int load_rccl(const char* path) {
    if (g_nccl.lib) return 0;
    void* h = nullptr;
    const char* names[] = {path, "librccl.so", "librccl.so.1"};
    for (const char* n : names) { if (n && n[0]) { h = dlopen(n, RTLD_NOW | RTLD_NOLOAD); if (h) break; } }      // the copy already in the process first
    for (const char* n : names) { if (h) break; if (n && n[0]) h = dlopen(n, RTLD_NOW | RTLD_LOCAL); }
    if (!h) { set_error("mirres_comm: cannot load librccl (%s)", dlerror()); return MIRRES_E_STATE; }
#define MR_SYM(field, name) do { *reinterpret_cast<void**>(&g_nccl.field) = dlsym(h, name); if (!g_nccl.field) { set_error("mirres_comm: librccl has no %s", name); return MIRRES_E_STATE; } } while (0)
    MR_SYM(GetUniqueId, "ncclGetUniqueId"); MR_SYM(CommInitRank, "ncclCommInitRank"); MR_SYM(CommDestroy, "ncclCommDestroy");
    MR_SYM(GroupStart, "ncclGroupStart"); MR_SYM(GroupEnd, "ncclGroupEnd"); MR_SYM(Send, "ncclSend"); MR_SYM(Recv, "ncclRecv"); MR_SYM(GetErrorString, "ncclGetErrorString");
#undef MR_SYM
    g_nccl.lib = h;
    return 0;
}
#define MR_NCCL(x) do { const int r_ = (x); if (r_ != 0) { set_error("%s: %s", #x, g_nccl.GetErrorString ? g_nccl.GetErrorString(r_) : "rccl error"); return MIRRES_E_STATE; } } while (0)

struct Comm { void* nccl; int world, rank; };

// one exchange step on the packed reservoirs `rec` ([local rows, fx, 8] floats): for every neighbour send rows [s0, s1) and receive rows [r0, r1)
int comm_exchange_halos(void* comm_, float* rec, int fx, int n, const int* peer, const int* s0, const int* s1, const int* r0, const int* r1, hipStream_t s) {
    Comm* c = static_cast<Comm*>(comm_);
    if (!c || !c->nccl) { set_error("mirres_render: halo_comm is not a communicator"); return MIRRES_E_ARG; }
    const size_t row = (size_t)fx * 8;
    const int ncclFloat32 = 7;
    MR_NCCL(g_nccl.GroupStart());
    for (int k = 0; k < n; k++) {
        if (s1[k] > s0[k]) MR_NCCL(g_nccl.Send(rec + (size_t)s0[k] * row, (size_t)(s1[k] - s0[k]) * row, ncclFloat32, peer[k], c->nccl, s));
        if (r1[k] > r0[k]) MR_NCCL(g_nccl.Recv(rec + (size_t)r0[k] * row, (size_t)(r1[k] - r0[k]) * row, ncclFloat32, peer[k], c->nccl, s));
    }
    MR_NCCL(g_nccl.GroupEnd());
    return 0;
}

}  // namespace mr

using namespace mr;

extern "C" {

int mirres_comm_unique_id(const char* librccl_path, void* id128) {
    if (!id128) { set_error("mirres_comm_unique_id: null"); return MIRRES_E_ARG; }
    if (int rc = load_rccl(librccl_path)) return rc;
    MR_NCCL(g_nccl.GetUniqueId(id128));
    return MIRRES_OK;
}

int mirres_comm_create(void** comm, const char* librccl_path, const void* id128, int world, int rank) {
    if (!comm || !id128 || world < 1 || rank < 0 || rank >= world) { set_error("mirres_comm_create: bad argument"); return MIRRES_E_ARG; }
    if (int rc = load_rccl(librccl_path)) return rc;
    Id128 id; memcpy(id.bytes, id128, 128);
    void* nc = nullptr;
    MR_NCCL(g_nccl.CommInitRank(&nc, world, id, rank));
    Comm* c = new Comm{nc, world, rank};
    *comm = c;
    return MIRRES_OK;
}

void mirres_comm_destroy(void* comm) {
    Comm* c = static_cast<Comm*>(comm);
    if (!c) return;
    if (c->nccl && g_nccl.CommDestroy) (void)g_nccl.CommDestroy(c->nccl);
    delete c;
}

}  // extern "C"
