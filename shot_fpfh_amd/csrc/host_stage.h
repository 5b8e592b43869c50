// host_stage.h -- what every entry point of the list-driven kernels starts and ends with: the checks of a list set against its
// cloud, and the staging of host / device operands (shared by normals_lrf.hip and shot.hip).
#pragma once
#include "common.h"

static inline int check_nbrs(sf_ctx *ctx, sf_cloud *c, sf_nbrs *nb, const char *who)
{
    if (!ctx || !c || !nb) { sf_set_error("%s: null argument", who); return SF_ERR_ARG; }
    if (!c->xs) { sf_set_error("%s: grid not built", who); return SF_ERR_STATE; }
    SF_CHECK(sf_nbrs_on_grid(nb, c, who));
    SF_HIP(hipSetDevice(ctx->device));
    return SF_OK;
}

// Host <-> device staging of one entry point.  Buffers come from the context pool through the caller's guard (released on
// every return path); copies are asynchronous on the context stream and stage_sync() closes the call when any host
// buffer was involved.
static inline int stage_in(sf_pool_guard &g, const double *src, size_t count, int flags, const double **dev)
{
    if (!src) { *dev = nullptr; return SF_OK; }
    if (flags & SF_IN_DEVICE) { *dev = src; return SF_OK; }
    double *owned = nullptr;
    SF_CHECK(g.alloc(&owned, count));
    if (count) SF_HIP(hipMemcpyAsync(owned, src, count * sizeof(double), hipMemcpyHostToDevice, g.ctx->stream));
    *dev = owned;
    return SF_OK;
}

static inline int stage_out(sf_pool_guard &g, double *dst, size_t count, int flags, double **dev)
{
    if (flags & SF_OUT_DEVICE) { *dev = dst; return SF_OK; }
    return g.alloc(dev, count);
}

static inline int finish_out(sf_ctx *ctx, double *dst, size_t count, int flags, const double *dev)
{
    if (!(flags & SF_OUT_DEVICE) && count)
        SF_HIP(hipMemcpyAsync(dst, dev, count * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    return SF_OK;
}

static inline int stage_sync(sf_ctx *ctx, int flags)
{
    if ((flags & (SF_IN_DEVICE | SF_OUT_DEVICE)) != (SF_IN_DEVICE | SF_OUT_DEVICE)) SF_HIP(hipStreamSynchronize(ctx->stream));
    return SF_OK;
}


