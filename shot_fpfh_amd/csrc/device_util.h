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

__device__ inline int sf_uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }

__device__ inline long long sf_uniform64(long long v)
{
    unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v);
    unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)((unsigned long long)v >> 32));
    return (long long)(((unsigned long long)hi << 32) | lo);
}
#endif

sf_grid_desc sf_make_grid_desc(const sf_cloud *c);
int sf_cloud_ensure_sorted_normals(struct sf_ctx *ctx, sf_cloud *c);
