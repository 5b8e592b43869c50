// device_util.h -- device-side helpers shared by the kernels (grid addressing, wave64 primitives).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

struct sf_cloud;

struct sf_grid_desc {
    double lo[3];
    double inv_cell;
    int dim[3];
};

#ifdef __HIPCC__
// Cell coordinate of v on one axis, clamped into the grid.  Clamping is monotone, so two values
// whose unclamped coordinates differ by <= 1 still differ by <= 1 after it -- which is what the
// 3x3x3 stencil of the radius search relies on for queries outside the cloud's bounding box too.
__host__ __device__ inline int sf_cell_coord(double v, double lo, double inv_cell, int dim)
{
    double t = floor((v - lo) * inv_cell);
    if (!(t >= 0.0)) t = 0.0; // negative or NaN
    double top = (double)(dim - 1);
    if (t > top) t = top;
    return (int)t;
}

__device__ inline int sf_lane() { return (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

// number of set bits of `mask` strictly below this lane
__device__ inline int sf_prefix_count(unsigned long long mask)
{
    return (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
}

__device__ inline double sf_wave_sum(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

// XCD-aware block remap.  Workgroups are dealt round-robin over the 8 XCDs (blockIdx % 8 says which blocks
// share an XCD / an L2).  Consecutive queries are spatial neighbours (cell-sorted order), so giving each
// XCD ONE contiguous eighth of the queries makes the cells a query needs hot in that XCD's own 4 MB L2
// instead of replicating the whole moving working set in all eight.  Launch grids padded with
// sf_xcd_grid(); the returned virtual block id may be >= the real block count (the caller's bound check
// on the query index covers it).  Placement only changes speed, never results.
__device__ inline long long sf_xcd_block()
{
    const unsigned b = blockIdx.x, chunk = gridDim.x >> 3; // gridDim.x is a multiple of 8
    return (long long)(b & 7u) * chunk + (b >> 3);
}

__device__ inline int sf_uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }

__device__ inline long long sf_uniform64(long long v)
{
    unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v);
    unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)((unsigned long long)v >> 32));
    return (long long)(((unsigned long long)hi << 32) | lo);
}
#endif

static inline unsigned sf_xcd_grid(long long blocks) { return (unsigned)(((blocks + 7) / 8) * 8); }

sf_grid_desc sf_make_grid_desc(const sf_cloud *c);
int sf_cloud_ensure_sorted_normals(struct sf_ctx *ctx, sf_cloud *c);
