// comm.hip -- multi-GPU exchange: RCCL over xGMI, one process per GPU.
//
// The reference has no distributed path (only a multiprocessing.Pool).  The path shards by
// blocks of query points; the two real exchange steps are the SPFH table between K6 and K7 (every
// keypoint needs the SPFH rows of its neighbours, which may belong to another shard) and the descriptor
// rows before matching.  The first is a neighbour-to-neighbour exchange of boundary rows (grouped
// ncclSend / ncclRecv, sf_comm_exchange) or an all-gather of the whole table; the second an all-gather of
// equally sized per-rank blocks; the matching's reciprocity test adds an all-reduce(min) of packed
// (distance, index) keys.
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

// Point-to-point exchange, all operations in ONE RCCL group: operation i sends send_bytes[i] bytes at send[i] to rank
// peer[i] and receives recv_bytes[i] bytes from it into recv[i] (either may be 0).  xGMI is point to point, so a
// rank's two z-slab neighbours are reached over two different links at once.  A lone context without a communicator
// may only name itself as the peer (device-to-device copy); with a communicator -- of one rank too -- a self
// exchange goes through ncclSend / ncclRecv like any other.
extern "C" int sf_comm_exchange(sf_ctx *ctx, int n_ops, const int *peer, const void *const *send, const size_t *send_bytes,
                                void *const *recv, const size_t *recv_bytes)
{
    if (!ctx || n_ops < 0 || (n_ops && (!peer || !send || !send_bytes || !recv || !recv_bytes))) {
        sf_set_error("sf_comm_exchange: bad argument");
        return SF_ERR_ARG;
    }
    for (int i = 0; i < n_ops; ++i) {
        if (peer[i] < 0 || peer[i] >= ctx->nranks) { sf_set_error("sf_comm_exchange: peer %d outside 0..%d", peer[i], ctx->nranks - 1); return SF_ERR_ARG; }
        if ((send_bytes[i] && !send[i]) || (recv_bytes[i] && !recv[i])) { sf_set_error("sf_comm_exchange: null buffer"); return SF_ERR_ARG; }
        if (peer[i] == ctx->rank && send_bytes[i] != recv_bytes[i]) { sf_set_error("sf_comm_exchange: a self exchange must send what it receives"); return SF_ERR_ARG; }
    }
    if (!ctx->comm) {
        if (ctx->nranks != 1) { sf_set_error("sf_comm_exchange: communicator not initialised"); return SF_ERR_STATE; }
        for (int i = 0; i < n_ops; ++i)
            if (send_bytes[i] && send[i] != recv[i])
                SF_HIP(hipMemcpyAsync(recv[i], send[i], send_bytes[i], hipMemcpyDeviceToDevice, ctx->stream));
        return SF_OK;
    }
    bool any = false;
    for (int i = 0; i < n_ops; ++i) any |= send_bytes[i] || recv_bytes[i];
    if (!any) return SF_OK;
    sf_launch_timer t_(ctx, "c_exchange");
    SF_NCCL(ncclGroupStart());
    for (int i = 0; i < n_ops; ++i) {
        ncclResult_t r = ncclSuccess;
        if (send_bytes[i]) r = ncclSend(send[i], send_bytes[i], ncclChar, peer[i], (ncclComm_t)ctx->comm, ctx->stream);
        if (r == ncclSuccess && recv_bytes[i]) r = ncclRecv(recv[i], recv_bytes[i], ncclChar, peer[i], (ncclComm_t)ctx->comm, ctx->stream);
        if (r != ncclSuccess) {
            (void)ncclGroupEnd();
            sf_set_error("sf_comm_exchange: %s", ncclGetErrorString(r));
            return SF_ERR_COMM;
        }
    }
    SF_NCCL(ncclGroupEnd());
    return SF_OK;
}

// Element-wise all-reduce (in place allowed) of n 32-bit integers (maximum) or n unsigned 64-bit integers (minimum).
// Without a communicator a lone context keeps its own values.
int sf_comm_allreduce_max_i32(sf_ctx *ctx, const int *send, int *recv, size_t n)
{
    if (!ctx->comm) {
        if (ctx->nranks != 1) { sf_set_error("sf_comm_allreduce: communicator not initialised"); return SF_ERR_STATE; }
        if (send != recv && n) SF_HIP(hipMemcpyAsync(recv, send, n * sizeof(int), hipMemcpyDeviceToDevice, ctx->stream));
        return SF_OK;
    }
    if (!n) return SF_OK;
    sf_launch_timer t_(ctx, "c_allreduce");
    SF_NCCL(ncclAllReduce(send, recv, n, ncclInt32, ncclMax, (ncclComm_t)ctx->comm, ctx->stream));
    return SF_OK;
}

extern "C" int sf_comm_allreduce_min_u64(sf_ctx *ctx, const void *send, void *recv, size_t n)
{
    if (!ctx || !send || !recv) { sf_set_error("sf_comm_allreduce_min_u64: null argument"); return SF_ERR_ARG; }
    if (!ctx->comm) {
        if (ctx->nranks != 1) { sf_set_error("sf_comm_allreduce: communicator not initialised"); return SF_ERR_STATE; }
        if (send != recv && n) SF_HIP(hipMemcpyAsync(recv, send, n * 8, hipMemcpyDeviceToDevice, ctx->stream));
        return SF_OK;
    }
    if (!n) return SF_OK;
    sf_launch_timer t_(ctx, "c_allreduce");
    SF_NCCL(ncclAllReduce(send, recv, n, ncclUint64, ncclMin, (ncclComm_t)ctx->comm, ctx->stream));
    return SF_OK;
}

// While on, every radius search of this context folds its list statistics over ALL ranks (one small all-reduce in front of
// the read-back the search ends with anyway): sf_nbrs_max_count_all then says how long the longest list of ANY rank is,
// which is what ranks that exchange SPFH rows must size their tables by -- the same storage on every rank.  Every rank
// must then run its searches in the same order (they are collective calls).
extern "C" int sf_comm_collective_stats(sf_ctx *ctx, int on)
{
    if (!ctx) { sf_set_error("null ctx"); return SF_ERR_ARG; }
    ctx->collective_stats = on != 0;
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
