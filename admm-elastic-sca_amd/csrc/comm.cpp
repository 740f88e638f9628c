// comm.cpp -- the all-reduce of the multi-GPU iteration: RCCL bound at run time (dlopen), caller hooks (device or host buffers),
// and the C ABI entry points that install them (include/admm_hip.h "multi-GPU").
#include "ctx.hpp"
#include <dlfcn.h>

using namespace admm_host;
using namespace admm_lib;

namespace admm_lib {

int fail(admm_hip_ctx *c, int code, const char *fmt, ...) {
    char buf[512];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
    if (c) c->err = buf;
    fprintf(stderr, "admm_hip: %s\n", buf);
    return code;
}

// ---- RCCL, bound at run time (dlopen): the library has no link-time dependency on it, single-GPU users never load it ----
struct nccl_uid { char internal[128]; };
struct RcclApi {
    void *handle = nullptr;
    int (*GetUniqueId)(nccl_uid *) = nullptr;
    int (*CommInitRank)(void **, int, nccl_uid, int) = nullptr;
    int (*CommDestroy)(void *) = nullptr;
    int (*AllReduce)(const void *, void *, size_t, int, int, void *, hipStream_t) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    int (*CommGetAsyncError)(void *, int *) = nullptr;      // optional: absent in very old builds
};
RcclApi *rccl_api(std::string *why) {
    static RcclApi api; static bool tried = false; static std::string err;
    if (!tried) {
        tried = true;
        // the copy already in the process first (PyTorch ships its own librccl.so): two RCCL instances must not share a job
        const char *names[] = {getenv("ADMM_HIP_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
        for (int pass = 0; pass < 2 && !api.handle; ++pass)
            for (const char *nm : names) { if (!nm || !*nm) continue; api.handle = dlopen(nm, RTLD_NOW | RTLD_GLOBAL | (pass == 0 ? RTLD_NOLOAD : 0)); if (api.handle) break; }
        if (!api.handle) err = std::string("librccl.so not found (") + (dlerror() ? dlerror() : "no dlerror") + "); set ADMM_HIP_RCCL_LIB";
        else {
            api.GetUniqueId = (int (*)(nccl_uid *))dlsym(api.handle, "ncclGetUniqueId");
            api.CommInitRank = (int (*)(void **, int, nccl_uid, int))dlsym(api.handle, "ncclCommInitRank");
            api.CommDestroy = (int (*)(void *))dlsym(api.handle, "ncclCommDestroy");
            api.AllReduce = (int (*)(const void *, void *, size_t, int, int, void *, hipStream_t))dlsym(api.handle, "ncclAllReduce");
            api.GetErrorString = (const char *(*)(int))dlsym(api.handle, "ncclGetErrorString");
            api.CommGetAsyncError = (int (*)(void *, int *))dlsym(api.handle, "ncclCommGetAsyncError");
            if (!api.GetUniqueId || !api.CommInitRank || !api.CommDestroy || !api.AllReduce) { err = "librccl.so lacks ncclGetUniqueId / ncclCommInitRank / ncclCommDestroy / ncclAllReduce"; api.handle = nullptr; }
        }
    }
    if (!api.handle) { if (why) *why = err; return nullptr; }
    return &api;
}
// sum `count` doubles in place across the ranks, on the context's stream: RCCL directly when a communicator is installed
// (admm_hip_rccl_init / admm_hip_set_rccl_comm: no host code between the kernels, capturable), otherwise the caller's hook
int do_allreduce(admm_hip_ctx *ctx, double *buf, int64_t count) {
    if (ctx->rccl_comm) {
        RcclApi *R = rccl_api(nullptr);
        const int rc = R ? R->AllReduce(buf, buf, (size_t)count, /*ncclDouble*/ 8, /*ncclSum*/ 0, ctx->rccl_comm, ctx->stream) : -1;
        if (rc != 0) return fail(ctx, ADMM_ERR_COMM, "ncclAllReduce failed: %s", (R && R->GetErrorString) ? R->GetErrorString(rc) : "RCCL not loaded");
        return ADMM_OK;
    }
    if (!ctx->allreduce) return fail(ctx, ADMM_ERR_COMM, "world size %d but neither an RCCL communicator nor an all-reduce hook is installed", ctx->world);
    if (ctx->allreduce(ctx->allreduce_user, buf, count, (void *)ctx->stream) != 0) return fail(ctx, ADMM_ERR_COMM, "all-reduce hook failed");
    return ADMM_OK;
}

// ncclCommGetAsyncError of the installed communicator (a peer that died, a failed transport): ADMM_OK while healthy / none installed
int comm_poll(admm_hip_ctx *ctx, int *nccl_result) {
    if (nccl_result) *nccl_result = 0;
    if (!ctx->rccl_comm) return ADMM_OK;
    RcclApi *R = rccl_api(nullptr);
    if (!R || !R->CommGetAsyncError) return ADMM_OK;
    int async = 0;
    const int rc = R->CommGetAsyncError(ctx->rccl_comm, &async);
    if (rc != 0) async = rc;
    if (nccl_result) *nccl_result = async;
    if (async != 0) return fail(ctx, ADMM_ERR_COMM, "RCCL communicator reports an asynchronous error (rank %d of %d): %s", ctx->rank, ctx->world, R->GetErrorString ? R->GetErrorString(async) : "?");
    return ADMM_OK;
}

// admm_hip_destroy: what this unit owns in the context
void comm_release(admm_hip_ctx *ctx) {
    if (ctx->rccl_comm && ctx->rccl_owned) { RcclApi *R = rccl_api(nullptr); if (R) (void)R->CommDestroy(ctx->rccl_comm); }
    ctx->rccl_comm = nullptr; ctx->rccl_owned = false;
    if (ctx->h_comm) (void)hipHostFree(ctx->h_comm);
    if (ctx->d_small) (void)hipFree(ctx->d_small);
    ctx->h_comm = nullptr; ctx->h_comm_cap = 0; ctx->d_small = nullptr; ctx->d_small_cap = 0;
}

} // namespace admm_lib

extern "C" {

int admm_hip_set_allreduce(admm_hip_ctx *ctx, admm_hip_allreduce_fn fn, void *user) {
    if (!ctx) return ADMM_ERR_ARG;
    ctx->allreduce = fn; ctx->allreduce_user = user;
    ctx->graphs_stale = true;
    return ADMM_OK;
}

// transports that only see host memory (MPI without GPU support, shared memory between the ranks of a node): the buffer is
// staged through pinned host memory around the caller's function
static int host_allreduce_trampoline(void *self, void *dev_buf, int64_t count, void *hip_stream) {
    admm_hip_ctx *ctx = (admm_hip_ctx *)self;
    hipStream_t st = (hipStream_t)hip_stream;
    if (!ctx->host_allreduce) return 1;
    if ((size_t)count > ctx->h_comm_cap) {
        (void)hipStreamSynchronize(st);      // the previous call's host-to-device copy may still be reading the old staging buffer
        if (ctx->h_comm) (void)hipHostFree(ctx->h_comm);      // (only the staging buffer: every stream / event of the context belongs to admm_hip_destroy)
        ctx->h_comm = nullptr; ctx->h_comm_cap = 0;
        if (hipHostMalloc((void **)&ctx->h_comm, sizeof(double) * (size_t)count, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return 1; }
        ctx->h_comm_cap = (size_t)count;
    }
    const size_t bytes = sizeof(double) * (size_t)count;
    if (hipMemcpyAsync(ctx->h_comm, dev_buf, bytes, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) return 1;
    if (ctx->host_allreduce(ctx->host_allreduce_user, ctx->h_comm, count) != 0) return 1;
    if (hipMemcpyAsync(dev_buf, ctx->h_comm, bytes, hipMemcpyHostToDevice, st) != hipSuccess) return 1;
    return 0;
}
int admm_hip_set_host_allreduce(admm_hip_ctx *ctx, admm_hip_host_allreduce_fn fn, void *user) {
    if (!ctx) return ADMM_ERR_ARG;
    ctx->host_allreduce = fn; ctx->host_allreduce_user = user;
    ctx->allreduce = fn ? host_allreduce_trampoline : nullptr; ctx->allreduce_user = fn ? ctx : nullptr;
    return ADMM_OK;
}

int admm_hip_rccl_unique_id(void *id128) {
    if (!id128) return ADMM_ERR_ARG;
    std::string why;
    RcclApi *R = rccl_api(&why);
    if (!R) { fprintf(stderr, "admm_hip: %s\n", why.c_str()); return ADMM_ERR_COMM; }
    nccl_uid id;
    if (R->GetUniqueId(&id) != 0) return ADMM_ERR_COMM;
    std::memcpy(id128, &id, sizeof id);
    return ADMM_OK;
}
int admm_hip_rccl_init(admm_hip_ctx *ctx, const void *id128, int rank, int world) {
    if (!ctx || !id128 || world < 1 || rank < 0 || rank >= world) return ADMM_ERR_ARG;
    if (ctx->device_id < 0) return fail(ctx, ADMM_ERR_HIP, "host-only context: no RCCL communicator");
    std::string why;
    RcclApi *R = rccl_api(&why);
    if (!R) return fail(ctx, ADMM_ERR_COMM, "%s", why.c_str());
    HIPCHK(hipSetDevice(ctx->device_id));       // the communicator binds to the calling thread's current device
    nccl_uid id; std::memcpy(&id, id128, sizeof id);
    void *comm = nullptr;
    const int rc = R->CommInitRank(&comm, world, id, rank);
    if (rc != 0 || !comm) return fail(ctx, ADMM_ERR_COMM, "ncclCommInitRank(rank %d of %d, device %d) failed: %s", rank, world, ctx->device_id, R->GetErrorString ? R->GetErrorString(rc) : "?");
    if (ctx->rccl_comm && ctx->rccl_owned) { (void)hipStreamSynchronize(ctx->stream); (void)R->CommDestroy(ctx->rccl_comm); }
    ctx->rccl_comm = comm; ctx->rccl_owned = true;
    ctx->graphs_stale = true;      // (a captured multi-GPU iteration names the communicator it was captured with)
    return ADMM_OK;
}
int admm_hip_set_rccl_comm(admm_hip_ctx *ctx, void *nccl_comm) {
    if (!ctx) return ADMM_ERR_ARG;
    std::string why;
    RcclApi *R = rccl_api(&why);
    if (nccl_comm && !R) return fail(ctx, ADMM_ERR_COMM, "%s", why.c_str());
    if (ctx->rccl_comm && ctx->rccl_owned && R) { if (ctx->stream) (void)hipStreamSynchronize(ctx->stream); (void)R->CommDestroy(ctx->rccl_comm); }
    ctx->rccl_comm = nccl_comm; ctx->rccl_owned = false;
    ctx->graphs_stale = true;
    return ADMM_OK;
}
int admm_hip_rccl_async_error(admm_hip_ctx *ctx, int *nccl_result) {
    if (!ctx) return ADMM_ERR_ARG;
    return comm_poll(ctx, nccl_result);
}
// parity / bring-up hook: sums `count` doubles of a caller-owned DEVICE buffer through the installed communicator or hook
int admm_hip_debug_allreduce(admm_hip_ctx *ctx, void *dev_buf, int64_t count) {
    if (!ctx || ctx->device_id < 0 || !dev_buf || count < 0) return ADMM_ERR_ARG;
    HIPCHK(hipSetDevice(ctx->device_id));
    TRY(do_allreduce(ctx, (double *)dev_buf, count));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return ADMM_OK;
}

// a small HOST vector summed across the ranks through the transport the iterations use (the class mirror: the released
// MovingAnchors' positions, owner's values + zeros elsewhere); world 1: nothing to do
int admm_hip_allreduce_host(admm_hip_ctx *ctx, double *host_buf, int64_t count) {
    if (!ctx || ctx->device_id < 0 || !host_buf || count < 0) return ADMM_ERR_ARG;
    if (ctx->world <= 1 || count == 0) return ADMM_OK;
    HIPCHK(hipSetDevice(ctx->device_id));
    if ((size_t)count > ctx->d_small_cap) {
        HIPCHK(hipStreamSynchronize(ctx->stream));
        if (ctx->d_small) (void)hipFree(ctx->d_small);
        ctx->d_small = nullptr; ctx->d_small_cap = 0;
        HIPCHK(hipMalloc((void **)&ctx->d_small, sizeof(double) * (size_t)count));
        ctx->d_small_cap = (size_t)count;
    }
    HIPCHK(hipMemcpyAsync(ctx->d_small, host_buf, sizeof(double) * (size_t)count, hipMemcpyHostToDevice, ctx->stream));
    TRY(do_allreduce(ctx, ctx->d_small, count));
    HIPCHK(hipMemcpyAsync(host_buf, ctx->d_small, sizeof(double) * (size_t)count, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return ADMM_OK;
}

} // extern "C"
