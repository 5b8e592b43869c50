// comm.hip -- multi-GPU exchange: RCCL (ncclAllGather) over xGMI, one process per GPU.
//
// The reference has no distributed path (only a multiprocessing.Pool).  The path shards by
// blocks of query points; the two real exchange steps are the SPFH table between K6 and K7 (every
// keypoint needs the SPFH rows of its neighbours, which may belong to another shard) and the descriptor
// rows before matching.  Both are plain all-gathers of equally sized per-rank blocks.
#include <rccl/rccl.h>

#include "common.h"

#define SF_NCCL(call)                                                                       \
    do {                                                                                    \
        ncclResult_t r_ = (call);                                                           \
        if (r_ != ncclSuccess) {                                                            \
            sf_set_error("%s failed: %s", #call, ncclGetErrorString(r_));                   \
            return SF_ERR_COMM;                                                             \
        }                                                                                   \
    } while (0)

extern "C" int sf_comm_unique_id(char id[128])
{
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is expected to be 128 bytes");
    ncclUniqueId uid;
    SF_NCCL(ncclGetUniqueId(&uid));
    memcpy(id, &uid, sizeof(uid));
    return SF_OK;
}

extern "C" int sf_comm_init(sf_ctx *ctx, const char id[128], int nranks, int rank)
{
    if (!ctx || !id || nranks < 1 || rank < 0 || rank >= nranks) { sf_set_error("sf_comm_init: bad argument"); return SF_ERR_ARG; }
    if (ctx->comm) { sf_set_error("sf_comm_init: communicator already initialised"); return SF_ERR_STATE; }
    SF_HIP(hipSetDevice(ctx->device));
    ncclUniqueId uid;
    memcpy(&uid, id, sizeof(uid));
    ncclComm_t comm = nullptr;
    SF_NCCL(ncclCommInitRank(&comm, nranks, uid, rank));
    ctx->comm = comm;
    ctx->nranks = nranks;
    ctx->rank = rank;
    return SF_OK;
}

extern "C" int sf_comm_allgather(sf_ctx *ctx, const void *send, void *recv, size_t bytes_per_rank)
{
    if (!ctx || !send || !recv) { sf_set_error("sf_comm_allgather: null argument"); return SF_ERR_ARG; }
    // Without a communicator a lone context has nothing to exchange: the block only has to be in place.  Once
    // sf_comm_init has run -- with ONE rank too -- every call goes through ncclAllGather, so a single-GPU box
    // exercises the very RCCL path the N-rank job uses.
    if (!ctx->comm) {
        if (ctx->nranks != 1) { sf_set_error("sf_comm_allgather: communicator not initialised"); return SF_ERR_STATE; }
        if (send != recv && bytes_per_rank)
            SF_HIP(hipMemcpyAsync(recv, send, bytes_per_rank, hipMemcpyDeviceToDevice, ctx->stream));
        return SF_OK;
    }
    if (!bytes_per_rank) return SF_OK;
    sf_launch_timer t_(ctx, "c_allgather");
    SF_NCCL(ncclAllGather(send, recv, bytes_per_rank, ncclChar, (ncclComm_t)ctx->comm, ctx->stream));
    return SF_OK;
}

extern "C" int sf_comm_destroy(sf_ctx *ctx)
{
    if (!ctx) return SF_OK;
    if (ctx->comm) {
        (void)hipStreamSynchronize(ctx->stream);
        (void)ncclCommDestroy((ncclComm_t)ctx->comm);
        ctx->comm = nullptr;
        ctx->nranks = 1;
        ctx->rank = 0;
    }
    return SF_OK;
}
