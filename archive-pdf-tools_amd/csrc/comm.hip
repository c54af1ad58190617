// Control-plane collectives for the page sharding of SURVEY.md 8e: pages are independent, so the only traffic
// between ranks is bytes of control data -- the work-queue descriptor from rank 0, per-page result records back,
// the maximum of the elapsed times.  They go over RCCL (xGMI on the node); pixels never leave their GPU.
//
// librccl.so is opened at run time (dlopen): libmrchip.so keeps no link-time dependency on it and single-GPU
// users never load it.  The unique id travels through the caller (a file or a socket, mrchip/dist.py): RCCL
// itself only needs every rank to call mrchip_comm_init with the same 128 bytes.
#include <dlfcn.h>

#include "mrchip_internal.h"

using namespace mrchip;

namespace {

// the few RCCL entry points used, with the types of rccl.h (ncclUniqueId is 128 opaque bytes passed by value)
struct UniqueId { char internal[128]; };
typedef int (*fn_get_id)(UniqueId *);
typedef int (*fn_init_rank)(void **, int, UniqueId, int);
typedef int (*fn_destroy)(void *);
typedef const char *(*fn_errstr)(int);
typedef int (*fn_bcast)(const void *, void *, size_t, int, int, void *, hipStream_t);
typedef int (*fn_allgather)(const void *, void *, size_t, int, void *, hipStream_t);
typedef int (*fn_allreduce)(const void *, void *, size_t, int, int, void *, hipStream_t);
constexpr int kUint8 = 1, kFloat64 = 8, kMax = 2, kSum = 0;       // ncclUint8, ncclFloat64, ncclMax, ncclSum

struct Rccl {
    void *so = nullptr;
    fn_get_id get_id = nullptr; fn_init_rank init_rank = nullptr; fn_destroy destroy = nullptr; fn_errstr errstr = nullptr;
    fn_bcast bcast = nullptr; fn_allgather allgather = nullptr; fn_allreduce allreduce = nullptr;
};

Rccl *rccl() {
    static Rccl r;
    static bool tried = false;
    if (!tried) {
        tried = true;
        const char *names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"};
        for (const char *n : names) {
            r.so = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
            if (r.so) break;
        }
        if (r.so) {
            r.get_id = (fn_get_id)dlsym(r.so, "ncclGetUniqueId");
            r.init_rank = (fn_init_rank)dlsym(r.so, "ncclCommInitRank");
            r.destroy = (fn_destroy)dlsym(r.so, "ncclCommDestroy");
            r.errstr = (fn_errstr)dlsym(r.so, "ncclGetErrorString");
            r.bcast = (fn_bcast)dlsym(r.so, "ncclBroadcast");
            r.allgather = (fn_allgather)dlsym(r.so, "ncclAllGather");
            r.allreduce = (fn_allreduce)dlsym(r.so, "ncclAllReduce");
            if (!r.get_id || !r.init_rank || !r.destroy || !r.bcast || !r.allgather || !r.allreduce) { dlclose(r.so); r.so = nullptr; }
        }
    }
    return r.so ? &r : nullptr;
}

}  // namespace

struct mrchip_comm {
    mrchip_ctx *ctx = nullptr;
    void *comm = nullptr;
    int rank = 0, world = 1;
    hipStream_t s = nullptr;
    void *dbuf = nullptr;      // device staging (RCCL moves device memory)
    size_t dbytes = 0;
};

#define NCCL_TRY(c, expr)                                                                        \
    do {                                                                                         \
        int _r = (expr);                                                                         \
        if (_r != 0) {                                                                           \
            Rccl *_l = rccl();                                                                   \
            set_error("%s -> %s", #expr, (_l && _l->errstr) ? _l->errstr(_r) : "rccl error");    \
            return MRCHIP_E_HIP;                                                                 \
        }                                                                                        \
    } while (0)

static int comm_stage(mrchip_comm *c, size_t bytes) {
    if (bytes > c->dbytes) {
        if (c->dbuf) HIP_TRY(hipFree(c->dbuf));
        c->dbytes = (bytes + 4095) & ~(size_t)4095;
        HIP_TRY(hipMalloc(&c->dbuf, c->dbytes));
    }
    return 0;
}

MRCHIP_EXPORT int mrchip_comm_unique_id(unsigned char *id128) {
    Rccl *l = rccl();
    if (!l) { set_error("librccl.so cannot be loaded: %s", dlerror()); return MRCHIP_E_UNSUPPORTED; }
    if (!id128) { set_error("comm_unique_id: bad arguments"); return MRCHIP_E_ARG; }
    UniqueId id;
    NCCL_TRY(nullptr, l->get_id(&id));
    memcpy(id128, id.internal, 128);
    return 0;
}

MRCHIP_EXPORT mrchip_comm *mrchip_comm_init(mrchip_ctx *ctx, int rank, int world, const unsigned char *id128) {
    Rccl *l = rccl();
    if (!l) { set_error("librccl.so cannot be loaded: %s", dlerror()); return nullptr; }
    if (!ctx || !id128 || world < 1 || rank < 0 || rank >= world) { set_error("comm_init: bad arguments"); return nullptr; }
    if (hipSetDevice(ctx->device) != hipSuccess) { set_error("comm_init: hipSetDevice failed"); return nullptr; }
    mrchip_comm *c = new mrchip_comm();
    c->ctx = ctx; c->rank = rank; c->world = world; c->s = ctx->streams[NSTREAMS - 1];
    UniqueId id;
    memcpy(id.internal, id128, 128);
    const int r = l->init_rank(&c->comm, world, id, rank);
    if (r != 0) { set_error("ncclCommInitRank -> %s", l->errstr ? l->errstr(r) : "rccl error"); delete c; return nullptr; }
    return c;
}

MRCHIP_EXPORT void mrchip_comm_destroy(mrchip_comm *c) {
    if (!c) return;
    (void)hipSetDevice(c->ctx->device);
    (void)hipStreamSynchronize(c->s);
    Rccl *l = rccl();
    if (l && c->comm) (void)l->destroy(c->comm);
    if (c->dbuf) (void)hipFree(c->dbuf);
    delete c;
}

// `bytes` of host memory from rank `root` to every rank (in place)
MRCHIP_EXPORT int mrchip_comm_bcast(mrchip_comm *c, void *buf, size_t bytes, int root) {
    if (!c || !buf) { set_error("comm_bcast: bad arguments"); return MRCHIP_E_ARG; }
    if (bytes == 0) return 0;
    HIP_TRY(hipSetDevice(c->ctx->device));
    TRY(comm_stage(c, bytes));
    if (c->rank == root) TRY(upload_1d(c->s, c->dbuf, buf, bytes));
    NCCL_TRY(c, rccl()->bcast(c->dbuf, c->dbuf, bytes, kUint8, root, c->comm, c->s));
    TRY(download_1d(c->s, buf, c->dbuf, bytes));
    HIP_TRY(hipStreamSynchronize(c->s));
    return 0;
}

// every rank contributes `bytes` of host memory; recv (world * bytes) holds them in rank order on every rank
MRCHIP_EXPORT int mrchip_comm_allgather(mrchip_comm *c, const void *send, size_t bytes, void *recv) {
    if (!c || !send || !recv) { set_error("comm_allgather: bad arguments"); return MRCHIP_E_ARG; }
    if (bytes == 0) return 0;
    HIP_TRY(hipSetDevice(c->ctx->device));
    TRY(comm_stage(c, bytes * (size_t)(c->world + 1)));
    unsigned char *d = static_cast<unsigned char *>(c->dbuf);
    TRY(upload_1d(c->s, d, send, bytes));
    NCCL_TRY(c, rccl()->allgather(d, d + bytes, bytes, kUint8, c->comm, c->s));
    TRY(download_1d(c->s, recv, d + bytes, bytes * (size_t)c->world));
    HIP_TRY(hipStreamSynchronize(c->s));
    return 0;
}

// op: 0 sum, 1 max over ranks of n doubles (in place); with n = 1 and a dummy value this is also the barrier
MRCHIP_EXPORT int mrchip_comm_allreduce_f64(mrchip_comm *c, double *vals, int n, int op) {
    if (!c || !vals || n < 1 || (op != 0 && op != 1)) { set_error("comm_allreduce_f64: bad arguments"); return MRCHIP_E_ARG; }
    HIP_TRY(hipSetDevice(c->ctx->device));
    TRY(comm_stage(c, (size_t)n * 8));
    TRY(upload_1d(c->s, c->dbuf, vals, (size_t)n * 8));
    NCCL_TRY(c, rccl()->allreduce(c->dbuf, c->dbuf, (size_t)n, kFloat64, op == 1 ? kMax : kSum, c->comm, c->s));
    TRY(download_1d(c->s, vals, c->dbuf, (size_t)n * 8));
    HIP_TRY(hipStreamSynchronize(c->s));
    return 0;
}
