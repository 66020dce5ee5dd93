// sm_gather.hip -- result collection across the GPUs of one node over RCCL (xGMI), behind the C ABI.
//
// The hot path shards with NO data-path collective: pair j -> device j mod n, every device computes
// its own maps.  What the north star names RCCL for is the collection of the results:
// "RCCL broadcast/gather over xGMI only for result collection".  The reference has no counterpart
// (one GPU, cudaMemcpy to the host: src/stereo.cu:402-403, src/image.cu:15-23); this is new work.
//
// One PROCESS drives n devices here, as the C batch host does (host/stereopar_batch.c: a thread pair
// per device): the communicator is made with ncclCommInitAll, and every call below enqueues the
// operations of all n ranks inside one ncclGroupStart / ncclGroupEnd, which is how a single thread
// must drive several ranks.  (The one-process-per-GPU shape -- bench.py under torchrun -- uses the
// same RCCL calls through torch.distributed: stereomatching_amd/shard.py.)
//
// librccl.so is loaded when sm_comm_create is first called, not when libstereo_hip.so is: a host that
// never collects over xGMI does not pay for (or depend on) it.

#include "sm_internal.h"

#include <dlfcn.h>
#include <mutex>
#include <rccl/rccl.h>
#include <stdlib.h>
#include <string.h>

namespace {

struct RcclApi {
    void *handle;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *);
    ncclResult_t (*CommDestroy)(ncclComm_t);
    ncclResult_t (*GroupStart)();
    ncclResult_t (*GroupEnd)();
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t);
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t);
    ncclResult_t (*Broadcast)(const void *, void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t);
    const char *(*GetErrorString)(ncclResult_t);
};

RcclApi g_rccl;
std::mutex g_once;                          // (communicators may be created from several host threads)
char g_path[1024];                          // sm_comm_set_rccl_library: tried before the default names
// A library that exports the symbol sm_rccl_host_stand_in declares itself a HOST stand-in for RCCL (tests/rccl_stub.c:
// ranks are records, Send / Recv are memcpy at ncclGroupEnd): its communicators are driven with host buffers and no
// device is touched -- the grouped point-to-point logic below can then run where there is one GPU, or none.
bool g_host_stand_in;

int load_rccl()
{
    std::lock_guard<std::mutex> lock(g_once);
    if (g_rccl.handle) return SM_OK;
    void *h = nullptr;
    if (g_path[0]) {
        h = dlopen(g_path, RTLD_NOW | RTLD_LOCAL);
        if (!h) return sm_fail(SM_ERR_HIP, "sm_comm_create: cannot load %s (%s)", g_path, dlerror());
    }
    for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
        if (h) break;
        h = dlopen(name, RTLD_NOW | RTLD_LOCAL);
    }
    if (!h) return sm_fail(SM_ERR_HIP, "sm_comm_create: cannot load librccl.so (%s)", dlerror());
    g_host_stand_in = dlsym(h, "sm_rccl_host_stand_in") != nullptr;
    RcclApi a;
    a.handle = h;
#define SM_SYM(field, sym)                                                         \
    a.field = reinterpret_cast<decltype(a.field)>(dlsym(h, sym));                    \
    if (!a.field) { dlclose(h); return sm_fail(SM_ERR_HIP, "sm_comm_create: librccl.so has no %s", sym); }
    SM_SYM(CommInitAll, "ncclCommInitAll")
    SM_SYM(CommDestroy, "ncclCommDestroy")
    SM_SYM(GroupStart, "ncclGroupStart")
    SM_SYM(GroupEnd, "ncclGroupEnd")
    SM_SYM(Send, "ncclSend")
    SM_SYM(Recv, "ncclRecv")
    SM_SYM(Broadcast, "ncclBroadcast")
    SM_SYM(GetErrorString, "ncclGetErrorString")
#undef SM_SYM
    g_rccl = a;
    return SM_OK;
}

}  // namespace

struct sm_comm {
    int n;
    int *devices;
    ncclComm_t *comms;      // rank r = devices[r]
};

#define SM_RCCL(call)                                                                       \
    do {                                                                                    \
        const ncclResult_t r_ = (call);                                                     \
        if (r_ != ncclSuccess)                                                              \
            return sm_fail(SM_ERR_HIP, "%s failed: %s", #call, g_rccl.GetErrorString(r_));  \
    } while (0)

extern "C" int sm_comm_set_rccl_library(const char *path)
{
    std::lock_guard<std::mutex> lock(g_once);
    if (g_rccl.handle) return sm_fail(SM_ERR_ARG, "sm_comm_set_rccl_library: the RCCL library is loaded already");
    if (path && strlen(path) >= sizeof g_path) return sm_fail(SM_ERR_ARG, "sm_comm_set_rccl_library: path too long");
    strcpy(g_path, path ? path : "");
    return SM_OK;
}

extern "C" int sm_comm_create(const int *devices, int n, sm_comm **out)
{
    if (!out) return sm_fail(SM_ERR_ARG, "sm_comm_create: out is NULL");
    *out = nullptr;
    if (!devices || n < 1 || n > 64) return sm_fail(SM_ERR_ARG, "sm_comm_create: need 1..64 devices");
    const int rc = load_rccl();             // (first: whether devices are looked at at all depends on the library)
    if (rc) return rc;
    int visible = n;
    if (!g_host_stand_in) SM_HIP(hipGetDeviceCount(&visible));
    for (int i = 0; i < n; i++) {
        if (devices[i] < 0 || devices[i] >= visible)
            return sm_fail(SM_ERR_ARG, "sm_comm_create: device %d is not one of the %d visible", devices[i], visible);
        for (int j = 0; j < i; j++)
            if (devices[j] == devices[i])     // RCCL refuses two ranks on one device; say so before it does
                return sm_fail(SM_ERR_ARG, "sm_comm_create: device %d is listed twice (one rank per device)", devices[i]);
    }
    sm_comm *c = (sm_comm *)calloc(1, sizeof *c);
    if (c) {
        c->devices = (int *)malloc(sizeof(int) * n);
        c->comms = (ncclComm_t *)calloc(n, sizeof(ncclComm_t));
    }
    if (!c || !c->devices || !c->comms) {
        if (c) { free(c->devices); free(c->comms); free(c); }
        return sm_fail(SM_ERR_NOMEM, "error: out of memory");
    }
    c->n = n;
    memcpy(c->devices, devices, sizeof(int) * n);
    const ncclResult_t r = g_rccl.CommInitAll(c->comms, n, c->devices);
    if (r != ncclSuccess) {
        free(c->devices); free(c->comms); free(c);
        return sm_fail(SM_ERR_HIP, "ncclCommInitAll over %d device(s) failed: %s", n, g_rccl.GetErrorString(r));
    }
    *out = c;
    return SM_OK;
}

extern "C" void sm_comm_destroy(sm_comm *comm)
{
    if (!comm) return;
    for (int r = 0; r < comm->n; r++)
        if (comm->comms[r]) (void)g_rccl.CommDestroy(comm->comms[r]);
    free(comm->devices);
    free(comm->comms);
    free(comm);
}

extern "C" int sm_comm_size(const sm_comm *comm) { return comm ? comm->n : 0; }

extern "C" int sm_broadcast(sm_comm *comm, void *const *d_buf, size_t bytes, void *const *streams)
{
    if (!comm || !d_buf) return sm_fail(SM_ERR_ARG, "sm_broadcast: comm / d_buf is NULL");
    for (int r = 0; r < comm->n; r++)
        if (!d_buf[r] && bytes) return sm_fail(SM_ERR_ARG, "sm_broadcast: d_buf[%d] is NULL", r);
    if (!bytes) return SM_OK;
    SM_RCCL(g_rccl.GroupStart());
    for (int r = 0; r < comm->n; r++) {
        const ncclResult_t e = g_rccl.Broadcast(d_buf[0], d_buf[r], bytes, ncclUint8, 0, comm->comms[r],
                                                streams ? (hipStream_t)streams[r] : nullptr);
        if (e != ncclSuccess) {
            (void)g_rccl.GroupEnd();
            return sm_fail(SM_ERR_HIP, "ncclBroadcast (rank %d) failed: %s", r, g_rccl.GetErrorString(e));
        }
    }
    SM_RCCL(g_rccl.GroupEnd());
    return SM_OK;
}

extern "C" int sm_gather_maps(sm_comm *comm, void *const *d_src, const size_t *bytes, void *d_dst,
                              void *const *streams)
{
    if (!comm || !d_src || !bytes) return sm_fail(SM_ERR_ARG, "sm_gather_maps: comm / d_src / bytes is NULL");
    size_t total = 0;
    for (int r = 0; r < comm->n; r++) {
        if (bytes[r] && !d_src[r]) return sm_fail(SM_ERR_ARG, "sm_gather_maps: d_src[%d] is NULL", r);
        total += bytes[r];
    }
    if (total && !d_dst) return sm_fail(SM_ERR_ARG, "sm_gather_maps: d_dst is NULL");
    // the root's own share: a copy on its own device, ordered on its stream
    if (bytes[0] && d_src[0] != d_dst) {
        if (g_host_stand_in) memcpy(d_dst, d_src[0], bytes[0]);
        else {
            SM_HIP(hipSetDevice(comm->devices[0]));
            SM_HIP(hipMemcpyAsync(d_dst, d_src[0], bytes[0], hipMemcpyDeviceToDevice,
                                  streams ? (hipStream_t)streams[0] : nullptr));
        }
    }
    if (comm->n == 1) return SM_OK;
    // every other rank SENDS exactly its own maps, the root receives them back to back in rank order:
    // all transfers of the group are in flight together (point to point over xGMI; nobody else moves a byte)
    SM_RCCL(g_rccl.GroupStart());
    size_t off = bytes[0];
    ncclResult_t e = ncclSuccess;
    for (int r = 1; r < comm->n && e == ncclSuccess; r++) {
        if (!bytes[r]) continue;
        e = g_rccl.Send(d_src[r], bytes[r], ncclUint8, 0, comm->comms[r], streams ? (hipStream_t)streams[r] : nullptr);
        if (e == ncclSuccess)
            e = g_rccl.Recv((char *)d_dst + off, bytes[r], ncclUint8, r, comm->comms[0],
                            streams ? (hipStream_t)streams[0] : nullptr);
        off += bytes[r];
    }
    const ncclResult_t ge = g_rccl.GroupEnd();
    if (e != ncclSuccess) return sm_fail(SM_ERR_HIP, "ncclSend / ncclRecv failed: %s", g_rccl.GetErrorString(e));
    if (ge != ncclSuccess) return sm_fail(SM_ERR_HIP, "ncclGroupEnd failed: %s", g_rccl.GetErrorString(ge));
    return SM_OK;
}
