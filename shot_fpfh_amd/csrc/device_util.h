// device_util.h -- device-side helpers shared by the kernels (grid addressing, wave64 primitives).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

struct sf_cloud;

struct sf_grid_desc {
    double lo[3];
    double inv_cell;   // 1 / edge of a cell along y and z
    double inv_cell_x; // xsub / edge: cells are xsub times finer along x
    double cell;       // the edge
    int dim[3];        // dim[0] counts fine x cells
    int xsub;
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

// Sum over the 64 lanes of a wave, result in every lane.  Built from DPP row operations (register-to-
// register, a few cycles each) instead of ds_bpermute shuffles, whose LDS round trips made the six-step
// dependency chain of a reduction the dominant latency of the list-sweeping kernels.
//   quad_perm xor1, xor2 -> row_half_mirror -> row_mirror : every lane holds its 16-lane row's sum
//   row_bcast15 (rows 1,3) -> row_bcast31 (rows 2,3)      : lane 63 holds the wave's sum -> v_readlane
__device__ inline double sf_wave_sum(double v)
{
    // (the four in-row steps write every lane: __builtin_amdgcn_mov_dpp leaves the destination's old value undefined, so the
    // compiler does not zero a register pair in front of each step -- two v_mov_b32 per step with update_dpp(0, ..))
#define SF_DPP_ADD_ALL(ctrl)                                                                                    \
    {                                                                                                           \
        const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), ctrl, 0xf, 0xf, true);                       \
        const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), ctrl, 0xf, 0xf, true);                       \
        v += __hiloint2double(hi, lo);                                                                          \
    }
#define SF_DPP_ADD(ctrl, row_mask)                                                                              \
    {                                                                                                           \
        const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), ctrl, row_mask, 0xf, false);           \
        const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), ctrl, row_mask, 0xf, false);           \
        v += __hiloint2double(hi, lo);                                                                          \
    }
    SF_DPP_ADD_ALL(0xB1)   // quad_perm [1,0,3,2]
    SF_DPP_ADD_ALL(0x4E)   // quad_perm [2,3,0,1]
    SF_DPP_ADD_ALL(0x141)  // row_half_mirror
    SF_DPP_ADD_ALL(0x140)  // row_mirror
    SF_DPP_ADD(0x142, 0xa) // row_bcast15 into rows 1 and 3 (other rows add the old value 0)
    SF_DPP_ADD(0x143, 0xc) // row_bcast31 into rows 2 and 3
#undef SF_DPP_ADD
#undef SF_DPP_ADD_ALL
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 63),
                            __builtin_amdgcn_readlane(__double2loint(v), 63));
}

// Maximum of non-negative values over the 64 lanes (same DPP ladder; rows that receive nothing keep max(v, 0)).
__device__ inline double sf_wave_max_nonneg(double v)
{
#define SF_DPP_MAX(ctrl, row_mask)                                                                              \
    {                                                                                                           \
        const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), ctrl, row_mask, 0xf, false);           \
        const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), ctrl, row_mask, 0xf, false);           \
        v = fmax(v, __hiloint2double(hi, lo));                                                                  \
    }
    SF_DPP_MAX(0xB1, 0xf)
    SF_DPP_MAX(0x4E, 0xf)
    SF_DPP_MAX(0x141, 0xf)
    SF_DPP_MAX(0x140, 0xf)
    SF_DPP_MAX(0x142, 0xa)
    SF_DPP_MAX(0x143, 0xc)
#undef SF_DPP_MAX
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 63),
                            __builtin_amdgcn_readlane(__double2loint(v), 63));
}

// Eight sums over the 64 lanes at once, by a transposing butterfly: after the exchanges with lane ^ 32, ^ 16 and ^ 8 a
// lane keeps ONE of the eight partial sums, v[4 b5 + 2 b4 + b3] (b = bits of the lane number), which three DPP steps
// then complete over the eight lanes that share those bits.  Returns that total: lane 8 i (and its seven neighbours)
// holds the sum of v[i].  7 exchange-and-add steps instead of 8 x 6.
__device__ inline double sf_wave_sum8(const double (&v)[8])
{
    const int lane = sf_lane();
    const bool b5 = lane & 32, b4 = lane & 16, b3 = lane & 8;
    double k4[4], k2[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const double mine = b5 ? v[4 + i] : v[i], send = b5 ? v[i] : v[4 + i];
        k4[i] = mine + __shfl_xor(send, 32);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const double mine = b4 ? k4[2 + i] : k4[i], send = b4 ? k4[i] : k4[2 + i];
        k2[i] = mine + __shfl_xor(send, 16);
    }
    double r;
    {
        const double mine = b3 ? k2[1] : k2[0], send = b3 ? k2[0] : k2[1];
        const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(send), 0x128, 0xf, 0xf, false); // row_ror:8 = lane ^ 8
        const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(send), 0x128, 0xf, 0xf, false);
        r = mine + __hiloint2double(hi, lo);
    }
#define SF_DPP_ADD(ctrl)                                                                                  \
    {                                                                                                     \
        const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(r), ctrl, 0xf, 0xf, false);          \
        const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(r), ctrl, 0xf, 0xf, false);          \
        r += __hiloint2double(hi, lo);                                                                    \
    }
    SF_DPP_ADD(0xB1)  // quad_perm [1,0,3,2]
    SF_DPP_ADD(0x4E)  // quad_perm [2,3,0,1]
    SF_DPP_ADD(0x141) // row_half_mirror: the other quad of the eight
#undef SF_DPP_ADD
    return r;
}

// Wave-wide sums of FOUR values at once, transposing as it goes: afterwards every lane of the 16-lane row i (lanes 16 i ..
// 16 i + 15) holds the sum of v[i] over the 64 lanes.  Two exchanges across rows and one four-step DPP row reduction -- half the
// instructions of two sf_wave_sum calls, a third of three.
__device__ inline double sf_wave_sum4(const double (&v)[4])
{
    const int lane = sf_lane();
    const bool b5 = lane & 32, b4 = lane & 16;
    double k2[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const double mine = b5 ? v[2 + i] : v[i], send = b5 ? v[i] : v[2 + i];
        k2[i] = mine + __shfl_xor(send, 32);
    }
    const double mine = b4 ? k2[1] : k2[0], send = b4 ? k2[0] : k2[1];
    double r = mine + __shfl_xor(send, 16);
#define SF_DPP_ADD(ctrl)                                                                                  \
    {                                                                                                     \
        const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(r), ctrl, 0xf, 0xf, false);          \
        const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(r), ctrl, 0xf, 0xf, false);          \
        r += __hiloint2double(hi, lo);                                                                    \
    }
    SF_DPP_ADD(0xB1)  // quad_perm [1,0,3,2]
    SF_DPP_ADD(0x4E)  // quad_perm [2,3,0,1]
    SF_DPP_ADD(0x141) // row_half_mirror
    SF_DPP_ADD(0x140) // row_mirror
#undef SF_DPP_ADD
    return r;
}
__device__ inline double sf_read_lane(double v, int lane) // (lane: wave-uniform)
{
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane), __builtin_amdgcn_readlane(__double2loint(v), lane));
}

// Sum over each 16-lane DPP row, result in every lane of the row (the first four steps of sf_wave_sum).
__device__ inline double sf_row16_sum(double v)
{
#define SF_DPP_ADD(ctrl)                                                                                  \
    {                                                                                                     \
        const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), ctrl, 0xf, 0xf, false);          \
        const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), ctrl, 0xf, 0xf, false);          \
        v += __hiloint2double(hi, lo);                                                                    \
    }
    SF_DPP_ADD(0xB1)  // quad_perm [1,0,3,2]
    SF_DPP_ADD(0x4E)  // quad_perm [2,3,0,1]
    SF_DPP_ADD(0x141) // row_half_mirror
    SF_DPP_ADD(0x140) // row_mirror
#undef SF_DPP_ADD
    return v;
}

__device__ inline int sf_row16_sum(int v)
{
    v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x141, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x140, 0xf, 0xf, false);
    return v;
}

// Stores of the descriptor rows (2.8 KB per SHOT row, 1 KB per FPFH row: 3.8 GB per pass at C3, never read again by the pass):
// non-temporal, so that they stream through the L2 instead of evicting the records and table rows the gathers of the same
// and of the next kernel live on (K5 1.52 -> 1.47 ms, the K6 that follows 0.81 -> 0.79: same-box A/B, tools/ab_libs.sh).
__device__ inline void sf_store_stream(double *p, double v) { __builtin_nontemporal_store(v, p); }
__device__ inline void sf_store_stream2(double *p, double a, double b)
{
    typedef double sf_d2 __attribute__((ext_vector_type(2)));
    const sf_d2 v = {a, b};
    __builtin_nontemporal_store(v, reinterpret_cast<sf_d2 *>(p));
}

// A neighbour list is written once (K2) and read exactly once by every kernel that walks it (0.44 GB per kernel at C3): streamed
// past the L2 both ways, like the descriptor rows.  Same-box A/B (tools/ab_libs.sh): the loads K5 -1 %, K7 -1.6 %; the stores
// K2 -2.8 %, the K6 behind it -1 %.
#define SF_LIST_LOAD(p) __builtin_nontemporal_load(p)
#define SF_LIST_STORE(p, v) __builtin_nontemporal_store((int)(v), p)

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

// gathers from the AoS records {x, y, z, nx, ny, nz}
__device__ inline void sf_load_xyz(const double *__restrict__ rec, int j, double &x, double &y, double &z)
{
    const double2 *p = reinterpret_cast<const double2 *>(rec + 6 * (size_t)j);
    const double2 a = p[0], b = p[1];
    x = a.x; y = a.y; z = b.x;
}
__device__ inline void sf_load_pn(const double *__restrict__ rec, int j, double &x, double &y, double &z, double &nx,
                                  double &ny, double &nz)
{
    const double2 *p = reinterpret_cast<const double2 *>(rec + 6 * (size_t)j);
    const double2 a = p[0], b = p[1], c = p[2];
    x = a.x; y = a.y; z = b.x; nx = b.y; ny = c.x; nz = c.y;
}

// 8-instruction v_rsq_f64 + Newton square root and a ~1 ulp reciprocal, for continuous quantities only (weights;
// no bin or sign decision hangs on their last bit)
__device__ inline double sf_sqrt_fast(double x) // x >= 0, normal range
{
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y, h = 0.5 * y;
    const double r = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, r, g);
    h = __builtin_fma(h, r, h);
    const double d = __builtin_fma(-g, g, x);
    g = __builtin_fma(d, h, g);
    return x > 0.0 ? g : 0.0;
}

__device__ inline double sf_rcp_fast(double d) // 1/d for normal d, ~1 ulp
{
    double r = __builtin_amdgcn_rcp(d);
    double e = __builtin_fma(-d, r, 1.0);
    r = __builtin_fma(r, e, r);
    e = __builtin_fma(-d, r, 1.0);
    return __builtin_fma(r, e, r);
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
int sf_cloud_ensure_inv_perm(struct sf_ctx *ctx, sf_cloud *c);
int sf_cloud_normals_max2(struct sf_ctx *ctx, sf_cloud *c, double *out); // max |n|^2 over the cloud's normals, cached
