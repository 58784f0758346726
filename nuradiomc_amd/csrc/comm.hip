// comm.hip -- the multi-GPU side of the C ABI: one process per GPU, events sharded with no exchange inside the compute;
// the only collective of the path is the gather of the per-rank triggered-event masks over xGMI (plus barrier / counter
// reductions for the measurement).  RCCL is bound directly (librccl.so, opened on first use so that single-GPU users never
// load it); the 128-byte communicator id travels between the processes through whatever host channel the launcher offers
// (nuradiomc_amd/comm.py: a TCP socket on MASTER_ADDR).
#include "../../include/nrhip.h"
#include "ctx.h"
#include <rccl/rccl.h>
#include <dlfcn.h>
#include <cstring>
#include <cstdio>

namespace {

struct Rccl {
    void* handle = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
};
Rccl g_rccl;

int load_rccl()
{
    if (g_rccl.handle) return 0;
    void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("/opt/rocm/lib/librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) return nrhip_fail_msg("nrhip_comm: librccl.so not found");
#define SYM(name)                                                                  \
    g_rccl.name = (decltype(g_rccl.name))dlsym(h, "nccl" #name);                    \
    if (!g_rccl.name) return nrhip_fail_msg("nrhip_comm: librccl.so lacks nccl" #name)
    SYM(GetUniqueId);
    SYM(CommInitRank);
    SYM(CommDestroy);
    SYM(AllGather);
    SYM(AllReduce);
    SYM(GetErrorString);
#undef SYM
    g_rccl.handle = h;
    return 0;
}

int rccl_fail(const char* what, ncclResult_t r)
{
    char buf[256];
    snprintf(buf, sizeof buf, "%s: %s", what, g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "RCCL error");
    return nrhip_fail_msg(buf);
}
#define RCCLCHK(x)                                           \
    do {                                                     \
        ncclResult_t r_ = (x);                               \
        if (r_ != ncclSuccess) return rccl_fail(#x, r_);     \
    } while (0)

__global__ void mask_or_kernel(long n, unsigned char* __restrict__ dst, const unsigned char* __restrict__ src, int overwrite)
{
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    // (OR: only where src is set, so that two streams OR-ing 0 / 1 masks into the same destination cannot lose each other's ones)
    if (i < n) {
        if (overwrite) dst[i] = src[i];
        else if (src[i]) dst[i] = (unsigned char)(dst[i] | src[i]);
    }
}

}  // namespace

struct nrhip_comm {
    nrhip_ctx* ctx;
    ncclComm_t comm;
    int rank, world;
    int64_t* token;  // device word of the barrier's all-reduce
};

extern "C" {

int nrhip_mask_or(nrhip_ctx* ctx, int64_t n, uint8_t* dst, const uint8_t* src, int32_t overwrite)
{
    if (!ctx || (n > 0 && (!dst || !src))) return nrhip_fail_msg("nrhip_mask_or: NULL argument");
    if (n <= 0) return 0;
    HIPCHK(hipSetDevice(ctx->device));
    hipLaunchKernelGGL(mask_or_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, (long)n, dst, src, overwrite);
    HIPCHK(hipGetLastError());
    return 0;
}

int nrhip_comm_get_unique_id(uint8_t id[NRHIP_COMM_ID_BYTES])
{
    if (!id) return nrhip_fail_msg("nrhip_comm_get_unique_id: NULL argument");
    if (load_rccl()) return -1;
    static_assert(NRHIP_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "id size");
    ncclUniqueId u;
    RCCLCHK(g_rccl.GetUniqueId(&u));
    memcpy(id, u.internal, NRHIP_COMM_ID_BYTES);
    return 0;
}

int nrhip_comm_create(nrhip_ctx* ctx, const uint8_t id[NRHIP_COMM_ID_BYTES], int32_t rank, int32_t world_size, nrhip_comm** out)
{
    if (!ctx || !id || !out) return nrhip_fail_msg("nrhip_comm_create: NULL argument");
    if (world_size < 1 || rank < 0 || rank >= world_size) return nrhip_fail_msg("nrhip_comm_create: bad rank / world size");
    if (load_rccl()) return -1;
    // everything that can fail on this rank alone comes BEFORE the collective ncclCommInitRank (a rank that drops out inside
    // it would leave the others blocked there); callers vote on nrhip_comm_get_unique_id (library present) and should set
    // NCCL_* time-outs for what remains (a peer dying inside the init)
    HIPCHK(hipSetDevice(ctx->device));
    int64_t* token = nullptr;
    HIPCHK(hipMalloc((void**)&token, sizeof(int64_t)));
    if (hipMemset(token, 0, sizeof(int64_t)) != hipSuccess) {
        (void)hipFree(token);
        return nrhip_fail_msg("nrhip_comm_create: hipMemset of the barrier word failed");
    }
    ncclUniqueId u;
    memcpy(u.internal, id, NRHIP_COMM_ID_BYTES);
    ncclComm_t c;
    const ncclResult_t r = g_rccl.CommInitRank(&c, world_size, u, rank);
    if (r != ncclSuccess) {
        (void)hipFree(token);
        return rccl_fail("ncclCommInitRank", r);
    }
    *out = new nrhip_comm{ctx, c, rank, world_size, token};
    return 0;
}

void nrhip_comm_destroy(nrhip_comm* c)
{
    if (!c) return;
    if (c->ctx) {
        (void)hipSetDevice(c->ctx->device);
        (void)hipStreamSynchronize(c->ctx->stream);
    }
    if (g_rccl.CommDestroy) (void)g_rccl.CommDestroy(c->comm);
    if (c->token) (void)hipFree(c->token);
    delete c;
}

int nrhip_comm_allgather_u8(nrhip_comm* c, const uint8_t* send, uint8_t* recv, int64_t count_per_rank)
{
    if (!c || !send || !recv || count_per_rank < 0) return nrhip_fail_msg("nrhip_comm_allgather_u8: bad argument");
    if (count_per_rank == 0) return 0;
    HIPCHK(hipSetDevice(c->ctx->device));
    RCCLCHK(g_rccl.AllGather(send, recv, (size_t)count_per_rank, ncclUint8, c->comm, c->ctx->stream));
    return 0;
}

int nrhip_comm_allreduce_i64_sum(nrhip_comm* c, int64_t* buf, int32_t n)
{
    if (!c || !buf || n < 0) return nrhip_fail_msg("nrhip_comm_allreduce_i64_sum: bad argument");
    if (n == 0) return 0;
    HIPCHK(hipSetDevice(c->ctx->device));
    RCCLCHK(g_rccl.AllReduce(buf, buf, (size_t)n, ncclInt64, ncclSum, c->comm, c->ctx->stream));
    return 0;
}

int nrhip_comm_allreduce_f64_max(nrhip_comm* c, double* buf, int32_t n)
{
    if (!c || !buf || n < 0) return nrhip_fail_msg("nrhip_comm_allreduce_f64_max: bad argument");
    if (n == 0) return 0;
    HIPCHK(hipSetDevice(c->ctx->device));
    RCCLCHK(g_rccl.AllReduce(buf, buf, (size_t)n, ncclFloat64, ncclMax, c->comm, c->ctx->stream));
    return 0;
}

int nrhip_comm_barrier(nrhip_comm* c)
{
    // every rank's stream work is finished, then one 8-byte all-reduce, then that is finished: a barrier over the GPUs
    if (!c) return nrhip_fail_msg("nrhip_comm_barrier: NULL argument");
    HIPCHK(hipSetDevice(c->ctx->device));
    HIPCHK(hipStreamSynchronize(c->ctx->stream));
    RCCLCHK(g_rccl.AllReduce(c->token, c->token, 1, ncclInt64, ncclSum, c->comm, c->ctx->stream));
    HIPCHK(hipStreamSynchronize(c->ctx->stream));
    return 0;
}

}  // extern "C"
