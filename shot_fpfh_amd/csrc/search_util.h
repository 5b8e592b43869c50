// search_util.h -- what the radius search (search.hip) and the k-NN search (knn.hip) share: the sort configuration, the stencil
// bounds of a query, the two-candidate load and the workgroup shape of the stencil sweeps.
#pragma once
#include <rocprim/device/device_radix_sort.hpp>
#include "common.h"
#include "device_util.h"

// (see grid.hip: Onesweep instead of the merge-sort fallback up to 2^20 items)
// ... and rocPRIM 4.2 carries no tuned Onesweep configuration for gfx950: the generic one sorts 4 bits per pass.  Ten bits per
// pass (1024-thread blocks, 6 items per thread, match ranking) sorts the 20-bit cell ids of a 1M-point cloud in two passes:
// 0.117 -> 0.068 ms (8 bits: 0.092-0.102, 11 bits: 0.090, 12 bits: does not fit LDS; tools/ab_k1.sh)
#ifndef SF_SORT_BITS
#define SF_SORT_BITS 10
#endif
#ifndef SF_SORT_BLOCK
#define SF_SORT_BLOCK 1024
#endif
#ifndef SF_SORT_ITEMS
#define SF_SORT_ITEMS 6
#endif
using sf_onesweep = rocprim::radix_sort_onesweep_config<rocprim::kernel_config<SF_SORT_BLOCK, SF_SORT_ITEMS>, rocprim::kernel_config<SF_SORT_BLOCK, SF_SORT_ITEMS>,
                                                        SF_SORT_BITS, rocprim::block_radix_rank_algorithm::match>;
using sf_sort_config = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config, sf_onesweep, 65536>;

namespace {


struct to_i64 {
    __host__ __device__ int64_t operator()(int32_t v) const { return (int64_t)v; }
};

__device__ inline void stencil_bounds(double v, double lo, double inv_cell, int dim, int &c0, int &c1)
{
    double t = floor((v - lo) * inv_cell);
    double top = (double)(dim - 1);
    double a = t - 1.0, b = t + 1.0;
    if (!(a >= 0.0)) a = 0.0;
    if (a > top) a = top;
    if (!(b >= 0.0)) b = 0.0;
    if (b > top) b = top;
    c0 = (int)a;
    c1 = (int)b;
}

// Two consecutive doubles fetched with ONE 16-byte load.  The vector-memory pipe of a CU accepts one wave
// instruction per 16 cycles whatever its width per lane, so K2 tests two candidates per lane: 3 loads per
// 128 candidates instead of 3 per 64.  (8-byte alignment only; global dwordx4 loads need dword alignment.)
struct __attribute__((aligned(8))) sf_dbl2 {
    double a, b;
};


// MODE 0: count only.  MODE 1: fill at the exact CSR offsets of a previous count + scan.
// MODE 2: optimistic single pass -- query q owns the fixed slot [q*cap, (q+1)*cap) of idx; hits beyond cap
//         are counted but not stored, and the host falls back to the exact two-pass scheme if any list
//         overflowed (HBM is plentiful: slots cost cap*4 B per query).
#ifndef SF_K2_STAGE
#define SF_K2_STAGE 1 // list entries leave through an LDS ring, 64 positions (256 aligned bytes) per store
#endif
#ifndef SF_K2_WPB
#define SF_K2_WPB 8 // waves per workgroup, four queries each (0.521 / 0.516 / 0.498 / 0.489 ms at C3 for 1 / 2 / 4 / 8)
#endif

#define SF_K2_SAMPLE 2048 // queries whose lists are counted before a first search sizes its slots

} // namespace

// search.hip, for knn.hip: queries into processing order, and the lists of a strided sample of them counted at a radius
int sf_k2_prepare_queries(sf_ctx *ctx, sf_cloud *c, sf_nbrs *nb, const double *queries, int flags);
int sf_k2_count_sample(sf_ctx *ctx, sf_cloud *c, sf_nbrs *nb, double r2, int32_t *sel_dev, int32_t *cnt_dev);

