// grid.hip -- K1: resident cloud + uniform grid (the MI355X stand-in for sklearn's KDTree(X)).
//
// Replaces: KDTree(cloud_points) at fpfh.py:26, shot_parallelization.py:167/220/229/283,
//           pca_based_descriptors.py:45-49.
// Layout in HBM: the caller's AoS xyz / normals are kept as uploaded; the grid build writes
// cell-sorted SoA positions (xs, ys, zs) so that a wave scanning a run of cells reads consecutive doubles,
// cell-sorted AoS records {x,y,z,nx,ny,nz} for the list-driven gathers, plus perm / inv_perm (sorted position <-> original index) and
// cell_start (first sorted position of every cell, x fastest).
// Roofline: HBM; ~ n * (24 read + 4+4 id/idx + sort passes + 48 gather + 48 write) bytes, one-off.
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_reduce.hpp>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/iterator/transform_iterator.hpp>

#include <algorithm>
#include <cmath>

#include <cstring>

#include "common.h"
#include "device_util.h"

// rocPRIM's radix sort falls back to a merge sort up to 2^20 items (a 1M-point cloud: block sort + 10 merge passes
// x 2 kernels = 21 launches, 0.16 ms); Onesweep above 64k items does the 16-17 cell-id bits in a few launches.
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

// The cell-id sort of a grid build: (key, value) pairs of int32, keys of `bits` bits.  (Smaller blocks / fewer items per thread
// for small inputs were measured on a 190k-point slab: 45-56 us against 50 -- the sort's time there is its nine dependent
// launches, not its bytes; such builds take the counting path below.)
static hipError_t sort_cells(void *tmp, size_t &tmp_bytes, int32_t *key_in, int32_t *key_out, int32_t *val_in, int32_t *val_out, size_t n, int bits,
                             hipStream_t stream)
{
    return rocprim::radix_sort_pairs<sf_sort_config>(tmp, tmp_bytes, key_in, key_out, val_in, val_out, n, 0, bits, stream);
}

namespace {

__global__ void k_bbox_partial(const double *__restrict__ xyz, int64_t n, double *__restrict__ partial)
{
    // partial[block][6] = min xyz, max xyz
    double mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        for (int a = 0; a < 3; ++a) {
            double v = xyz[3 * i + a];
            if (!(fabs(v) < INFINITY)) v = INFINITY; // NaN / inf poison the maximum (fmin/fmax would drop a NaN)
            mn[a] = fmin(mn[a], v);
            mx[a] = fmax(mx[a], v);
        }
    __shared__ double s[6][4];
    int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int a = 0; a < 3; ++a) {
        double v0 = mn[a], v1 = mx[a];
        for (int off = 32; off > 0; off >>= 1) {
            v0 = fmin(v0, __shfl_xor(v0, off));
            v1 = fmax(v1, __shfl_xor(v1, off));
        }
        if (lane == 0) {
            s[a][wave] = v0;
            s[3 + a][wave] = v1;
        }
    }
    __syncthreads();
    if (threadIdx.x < 6) {
        int a = threadIdx.x;
        double v = s[a][0];
        for (int w = 1; w < (int)(blockDim.x >> 6); ++w) v = a < 3 ? fmin(v, s[a][w]) : fmax(v, s[a][w]);
        partial[blockIdx.x * 6 + a] = v;
    }
}

// second stage of the bounding box: one block, wave a folds component a of the per-block partials
__global__ __launch_bounds__(384) void k_bbox_final(const double *__restrict__ partial, int nb, double *__restrict__ out6)
{
    const int a = threadIdx.x >> 6, lane = threadIdx.x & 63; // a = 0..5
    double v = a < 3 ? INFINITY : -INFINITY;
    for (int b = lane; b < nb; b += 64) v = a < 3 ? fmin(v, partial[b * 6 + a]) : fmax(v, partial[b * 6 + a]);
    for (int off = 32; off > 0; off >>= 1) {
        const double o = __shfl_xor(v, off);
        v = a < 3 ? fmin(v, o) : fmax(v, o);
    }
    if (lane == 0) out6[a] = v;
}

// cell ids of the internal points [first, first + n) (the whole cloud, or the slab of a block build -- a contiguous run of
// the z-sorted internal order), relative to cid_base, with their internal indices as the values to be sorted along
__global__ void k_cell_ids(const double *__restrict__ xyz, int64_t first, int64_t n, sf_grid_desc g, int32_t *__restrict__ cid,
                           int32_t *__restrict__ val, int cid_base)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int64_t o = first + i;
    int cx = sf_cell_coord(xyz[3 * o + 0], g.lo[0], g.inv_cell_x, g.dim[0]);
    int cy = sf_cell_coord(xyz[3 * o + 1], g.lo[1], g.inv_cell, g.dim[1]);
    int cz = sf_cell_coord(xyz[3 * o + 2], g.lo[2], g.inv_cell, g.dim[2]);
    cid[i] = (cz * g.dim[1] + cy) * g.dim[0] + cx - cid_base;
    val[i] = (int32_t)o;
}

// upload: the z coordinates as sort keys with the caller's indices as values; the gather into internal order afterwards
__global__ void k_upload_keys(const double *__restrict__ xyz, int64_t n, double *__restrict__ z, int32_t *__restrict__ idx)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    z[i] = xyz[3 * i + 2];
    idx[i] = (int32_t)i;
}

__global__ void k_upload_gather(const double *__restrict__ src, const int32_t *__restrict__ zperm, int64_t n, double *__restrict__ dst)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int64_t o = zperm[i];
    dst[3 * i + 0] = src[3 * o + 0];
    dst[3 * i + 1] = src[3 * o + 1];
    dst[3 * i + 2] = src[3 * o + 2];
}

// first internal point of every z-layer of cells (= the layer's first cell-sorted position: layers are the slowest axis of
// the cell numbering): lower bound of the layer index in the ascending z array.  One WAVE per layer, 64 probes per round
// (four dependent rounds for 8M points instead of the 23 of a binary search: the build waits for this table on the host).
__global__ __launch_bounds__(256) void k_layer_bounds(const double *__restrict__ z, int64_t n, double lo, double inv_cell, int nlayers,
                                                      int64_t *__restrict__ first)
{
    const int lane = threadIdx.x & 63;
    const int l = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (l > nlayers) return;
    int64_t a = 0, b = n; // the answer (first i with layer(z[i]) >= l; the layer of a coordinate is monotone in it) lies in [a, b]
    while (b - a > 64) {
        const int64_t step = (b - a + 63) / 64;
        const int64_t i = a + (int64_t)lane * step; // probes a, a + step, ...: `below` is true for a prefix of the lanes
        const bool below = i < b && sf_cell_coord(z[i], lo, inv_cell, nlayers) < l;
        const int k = __popcll(__ballot(below));
        if (k == 0) { b = a; break; }
        const int64_t na = a + (int64_t)(k - 1) * step + 1; // probe k - 1 is below: the answer is past it
        const int64_t nb = k < 64 ? a + (int64_t)k * step : b;  // probe k (if any) is not below: the answer is at or before it
        a = na;
        b = nb < b ? nb : b;
    }
    if (b > a) {
        const int64_t i = a + lane;
        const bool below = i < b && sf_cell_coord(z[i], lo, inv_cell, nlayers) < l;
        a += __popcll(__ballot(below));
    }
    if (lane == 0) first[l] = l == nlayers ? n : a;
}

// perm_int holds the sorted values (internal indices); perm gets the caller's numbering through zperm
__global__ void k_gather_sorted(const double *__restrict__ xyz, const double *__restrict__ nrm, const int32_t *__restrict__ perm_int,
                                const int32_t *__restrict__ zperm, int32_t *__restrict__ perm,
                                int64_t base, int64_t n, double *__restrict__ xs, double *__restrict__ ys,
                                double *__restrict__ zs, double *__restrict__ rec, unsigned *__restrict__ zero_word)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0 && zero_word) *zero_word = 0u; // (the long-gap counter of the cell-table kernel that follows: no memset launch)
    if (i >= n) return;
    i += base;
    int64_t o = perm_int[i];
    perm[i] = zperm[o];
    const double x = xyz[3 * o + 0], y = xyz[3 * o + 1], z = xyz[3 * o + 2];
    xs[i] = x; ys[i] = y; zs[i] = z;
    rec[6 * i + 0] = x; rec[6 * i + 1] = y; rec[6 * i + 2] = z;
    if (nrm) {
        rec[6 * i + 3] = nrm[3 * o + 0];
        rec[6 * i + 4] = nrm[3 * o + 1];
        rec[6 * i + 5] = nrm[3 * o + 2];
    }
}

// inv_perm[perm[i]] = i: a scattered 4-byte write per point (a 64-byte sector each), so only made when somebody asks for
// positions by original index (keypoint subsets, k-NN lists): sf_cloud_ensure_inv_perm
__global__ void k_inv_perm(const int32_t *__restrict__ perm, int64_t base, int64_t n, int32_t *__restrict__ inv_perm)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    i += base;
    inv_perm[perm[i]] = (int32_t)i;
}

// normals: slots 3..5 of the AoS records
__global__ void k_gather_normals(const double *__restrict__ nrm, const int32_t *__restrict__ perm, int64_t base, int64_t n,
                                 double *__restrict__ rec)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    i += base;
    int64_t o = perm[i];
    rec[6 * i + 3] = nrm[3 * o + 0];
    rec[6 * i + 4] = nrm[3 * o + 1];
    rec[6 * i + 5] = nrm[3 * o + 2];
}

// cell_start[c] = first sorted position whose cell id >= c (lower bound over the n populated positions that start at
// `base`; cells before / after the populated slab come out empty).  One thread per sorted POSITION i (and one past the
// last): the cells (id[i - 1], id[i]] all start at i.  (Until round 3: one thread per CELL with a binary search over the
// ids -- 20 dependent loads per cell, 0.45 ms for the 1.8e8 mostly empty fine cells of a clustered cloud.  The table is now
// written once, at streaming speed: a gap of more than a few cells is filled by the whole wave, 64 cells per store.)
// (sorted_cid holds cell id - cid_base: a block build sorts ids relative to its slab's first cell -- fewer key bits)
#define SF_LONG_GAP 65536 // cells: a longer run of empty cells is filled by the whole grid (k_cell_fill_long), not by one wave
struct sf_gap { int64_t lo, len; int32_t val, pad; };
__global__ __launch_bounds__(256) void k_cell_start(const int32_t *__restrict__ sorted_cid, int64_t cid_base, int64_t base, int64_t n,
                                                    int64_t ncell, int32_t *__restrict__ cell_start, sf_gap *__restrict__ gaps,
                                                    unsigned *__restrict__ n_gaps)
{
    const int lane = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t lo = 0, len = 0;
    if (i <= n) {
        lo = i == 0 ? 0 : (int64_t)sorted_cid[i - 1] + cid_base + 1;
        const int64_t hi = i == n ? ncell : (int64_t)sorted_cid[i] + cid_base; // inclusive
        len = hi - lo + 1;
        if (len < 0) len = 0;
    }
    const int32_t val = (int32_t)(base + i);
    if (len > SF_LONG_GAP) { // (the cells outside a rank's slab, the void between two far-apart parts of a cloud)
        const unsigned e = atomicAdd(n_gaps, 1u);
        gaps[e] = sf_gap{lo, len, val, 0};
        len = 0;
    }
    const bool wide = len > 8;
    if (!wide)
        for (int64_t t = 0; t < len; ++t) cell_start[lo + t] = val;
    unsigned long long todo = __ballot(wide);
    while (todo) {
        const int src = __builtin_ctzll(todo);
        todo &= todo - 1;
        const int64_t l0 = __shfl(lo, src), ln = __shfl(len, src);
        const int32_t v = __shfl(val, src);
        for (int64_t t = lane; t < ln; t += 64) cell_start[l0 + t] = v;
    }
}

__global__ __launch_bounds__(256) void k_cell_fill_long(const sf_gap *__restrict__ gaps, const unsigned *__restrict__ n_gaps,
                                                        int32_t *__restrict__ cell_start)
{
    const unsigned ng = *n_gaps;
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, nt = (int64_t)gridDim.x * blockDim.x;
    for (unsigned e = 0; e < ng; ++e) {
        const sf_gap g = gaps[e];
        for (int64_t t = tid; t < g.len; t += nt) cell_start[g.lo + t] = g.val;
    }
}

// ---- the counting build ------------------------------------------------------------------------------------------------
// When the grid is not much sparser than the points (at most SF_COUNT_CELLS_PER_POINT cells of the populated slab per point:
// every uniform or surface cloud at its own radius) the cell order comes from the cell table itself instead of a radix sort:
//   count   every point adds one to its cell's counter; what the atomic returns is its (arbitrary) place among the cell's points
//   scan    exclusive prefix sum of the counters = the cell table (one pass, decoupled look-back); cells outside the slab filled;
//           the counters go back to zero (they belong to the cloud and are zero between builds: no fill launch)
//   place   every point drops (its index, its cell) at cell start + that place
//   settle  every position ranks its index among the cell's (a cell holds ~1 point; k points cost k compares each, which is
//           k / 67 of what the radius search spends on the same point) and gathers its point to cell start + rank
// = the stable order the radix sort gives (ascending internal index inside a cell), bit for bit, in four dependent GPU
// operations instead of thirteen (rocPRIM's Onesweep alone is two kernels + two passes + five fills): a build of a 190k-point
// slab is bound by that count, not by bytes.
#define SF_COUNT_CELLS_PER_POINT 8
#define SF_Z_HOST_MAX 67108864LL // points: up to here a block build keeps the sorted z coordinates on the host too (512 MB)
#define SF_Z_HOST_LAYERS 4096    // ... and looks the layer bounds up there if the grid has at most this many z-layers
#define SF_SCAN_TPB 256
#define SF_SCAN_ITEMS 16
#define SF_SCAN_TILE (SF_SCAN_TPB * SF_SCAN_ITEMS)
#define SF_SCAN_AGGREGATE (1ull << 62)
#define SF_SCAN_PREFIX (2ull << 62)

__global__ void k_cell_count(const double *__restrict__ xyz, int64_t first, int64_t n, sf_grid_desc g, int cid_base,
                             int32_t *__restrict__ count, int32_t *__restrict__ cid, int32_t *__restrict__ place,
                             unsigned long long *__restrict__ status, int64_t n_status)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_status) status[i] = 0ull; // (the scan's ticket + tile states: no fill launch of their own)
    if (i >= n) return;
    const int64_t o = first + i;
    const int cx = sf_cell_coord(xyz[3 * o + 0], g.lo[0], g.inv_cell_x, g.dim[0]);
    const int cy = sf_cell_coord(xyz[3 * o + 1], g.lo[1], g.inv_cell, g.dim[1]);
    const int cz = sf_cell_coord(xyz[3 * o + 2], g.lo[2], g.inv_cell, g.dim[2]);
    const int c = (cz * g.dim[1] + cy) * g.dim[0] + cx - cid_base;
    cid[i] = c;
    place[i] = atomicAdd(&count[c], 1);
}

// cell_start[cid_base + c] = base + (number of points in the slab's cells before c) from the counters count[c]; the cells
// in front of the slab get base, those behind it (and the one-past-the-end entry) base + ns.  Blocks [0, slab_tiles) take the
// slab's tiles in the order they START (a ticket), so a tile's predecessors are always resident: the look-back cannot starve.
__global__ __launch_bounds__(SF_SCAN_TPB) void k_cell_scan(int32_t *__restrict__ count, int32_t *__restrict__ cell_start, int64_t ncell, int64_t cid_base,
                                                           int64_t slab_cells, int64_t slab_tiles, int32_t base, int32_t ns,
                                                           unsigned long long *__restrict__ status)
{
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    if ((int64_t)blockIdx.x >= slab_tiles) {
        const int64_t b = (int64_t)blockIdx.x - slab_tiles, outside = ncell + 1 - slab_cells;
        for (int k = 0; k < SF_SCAN_ITEMS; ++k) {
            const int64_t j = b * SF_SCAN_TILE + (int64_t)k * SF_SCAN_TPB + tid;
            if (j < outside) cell_start[j < cid_base ? j : j + slab_cells] = j < cid_base ? base : base + ns;
        }
        return;
    }
    __shared__ int64_t s_tile;
    __shared__ int s_wave[SF_SCAN_TPB / 64];
    __shared__ int s_prefix;
    if (tid == 0) s_tile = (int64_t)atomicAdd(&status[0], 1ull);
    __syncthreads();
    const int64_t t = s_tile;
    unsigned long long *st = status + 1;
    const int64_t c0 = t * SF_SCAN_TILE + (int64_t)tid * SF_SCAN_ITEMS;
    int32_t *p = cell_start + cid_base + c0, *q = count + c0;
    const int64_t left = slab_cells - c0;
    int v[SF_SCAN_ITEMS];
    int sum = 0;
#pragma unroll
    for (int k = 0; k < SF_SCAN_ITEMS; ++k) {
        const int x = k < left ? q[k] : 0;
        v[k] = sum;
        sum += x;
    }
#pragma unroll
    for (int k = 0; k < SF_SCAN_ITEMS; ++k)
        if (k < left) q[k] = 0; // (the counters are all zero again when the build is over: the next one starts without a fill)
    int inc = sum; // inclusive scan of the threads' sums over the wave
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int o = __shfl_up(inc, off);
        if (lane >= off) inc += o;
    }
    if (lane == 63) s_wave[w] = inc;
    __syncthreads();
    int before = inc - sum, total = 0;
#pragma unroll
    for (int k = 0; k < SF_SCAN_TPB / 64; ++k) {
        if (k < w) before += s_wave[k];
        total += s_wave[k];
    }
    if (w == 0) {
        int excl = 0;
        if (t == 0) {
            if (lane == 0) __hip_atomic_store(&st[0], SF_SCAN_PREFIX | (unsigned long long)(unsigned)total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            if (lane == 0) __hip_atomic_store(&st[t], SF_SCAN_AGGREGATE | (unsigned long long)(unsigned)total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            for (int64_t pos = t - 1;; pos -= 64) {
                const int64_t j = pos - lane;
                unsigned long long s;
                do {
                    s = j >= 0 ? __hip_atomic_load(&st[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : SF_SCAN_PREFIX;
                } while (__ballot((s >> 62) == 0ull));
                const unsigned long long have = __ballot((s >> 62) == 2ull);
                const int stop = have ? __builtin_ctzll(have) : 64; // nearest predecessor that already knows its inclusive prefix
                int val = lane <= stop ? (int)(unsigned)(s & 0xffffffffull) : 0;
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) val += __shfl_xor(val, off);
                excl += val;
                if (have) break;
            }
            if (lane == 0) __hip_atomic_store(&st[t], SF_SCAN_PREFIX | (unsigned long long)(unsigned)(excl + total), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (lane == 0) s_prefix = excl;
    }
    __syncthreads();
    const int add = base + s_prefix + before;
#pragma unroll
    for (int k = 0; k < SF_SCAN_ITEMS; ++k)
        if (k < left) p[k] = add + v[k];
}

// slot[cell start + place - base] = (internal index, cell)
__global__ void k_cell_place(const int32_t *__restrict__ cid, const int32_t *__restrict__ place, const int32_t *__restrict__ start_rel,
                             int64_t first, int64_t n, int2 *__restrict__ slot)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int c = cid[i];
    slot[(int64_t)start_rel[c] + place[i] - first] = make_int2((int)(first + i), c);
}

// position p of the slab: the point that landed there goes to its cell's start + (number of the cell's points with a smaller
// internal index), and is gathered there (what k_gather_sorted does after a radix sort)
__global__ void k_cell_settle(const int2 *__restrict__ slot, const int32_t *__restrict__ start_rel, const double *__restrict__ xyz,
                              const double *__restrict__ nrm, const int32_t *__restrict__ zperm, int32_t *__restrict__ perm,
                              int32_t *__restrict__ perm_int, int64_t base, int64_t n, double *__restrict__ xs, double *__restrict__ ys,
                              double *__restrict__ zs, double *__restrict__ rec)
{
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const int2 me = slot[p];
    const int64_t cs = (int64_t)start_rel[me.y] - base, ce = (int64_t)start_rel[me.y + 1] - base;
    int r = 0;
    for (int64_t j = cs; j < ce; ++j) r += slot[j].x < me.x ? 1 : 0;
    const int64_t i = base + cs + r, o = me.x;
    perm_int[i] = me.x;
    perm[i] = zperm[o];
    const double x = xyz[3 * o + 0], y = xyz[3 * o + 1], z = xyz[3 * o + 2];
    xs[i] = x; ys[i] = y; zs[i] = z;
    rec[6 * i + 0] = x; rec[6 * i + 1] = y; rec[6 * i + 2] = z;
    if (nrm) {
        rec[6 * i + 3] = nrm[3 * o + 0];
        rec[6 * i + 4] = nrm[3 * o + 1];
        rec[6 * i + 5] = nrm[3 * o + 2];
    }
}

// first cell-sorted position of every z-layer, read off the cell table (after a whole-cloud build)
__global__ void k_layer_first(const int32_t *__restrict__ cell_start, int64_t layer_cells, int nl, int64_t *__restrict__ out)
{
    const int z = blockIdx.x * blockDim.x + threadIdx.x;
    if (z <= nl) out[z] = cell_start[(int64_t)z * layer_cells];
}

} // namespace

static void cloud_release_grid(sf_ctx *ctx, sf_cloud *c)
{
    void *ptrs[] = {c->cell_start, c->perm, c->perm_int, c->inv_perm, c->xs, c->ys, c->zs, c->rec};
    for (void *p : ptrs)
        if (p) {
            if (ctx) sf_pool_release(ctx, p); else (void)hipFree(p);
        }
    c->cell_start = c->perm = c->perm_int = c->inv_perm = nullptr;
    c->inv_perm_valid = false;
    c->xs = c->ys = c->zs = c->rec = nullptr;
    c->normals_sorted = false;
    c->cell = 0.0;
    c->pop_begin = c->pop_end = 0;
    c->layer_first.clear();
}

// internal (z-sorted) copy of a caller-ordered n x 3 array: through a pooled staging buffer unless the source is on the device
static int upload_in_zorder(sf_ctx *ctx, sf_cloud *c, const double *src, int flags, double *dst)
{
    const int64_t n = c->n;
    if (!n) return SF_OK;
    sf_pool_guard tmp(ctx);
    const double *dsrc = src;
    if (!(flags & SF_IN_DEVICE)) {
        double *stage = nullptr;
        SF_CHECK(tmp.alloc(&stage, (size_t)n * 3));
        SF_HIP(hipMemcpyAsync(stage, src, (size_t)n * 24, hipMemcpyHostToDevice, ctx->stream));
        dsrc = stage;
    }
    SF_LAUNCH(ctx, "k0_upload", k_upload_gather, dim3((unsigned)sf_div_up(n, 256)), dim3(256), dsrc, (const int32_t *)c->zperm, n, dst);
    SF_HIP(hipStreamSynchronize(ctx->stream)); // (the caller's buffer / the staging block are free again)
    return SF_OK;
}

static int cloud_upload(sf_ctx *ctx, sf_cloud *c, const double *xyz, const double *normals, int flags)
{
    const int64_t n = c->n;
    const size_t bytes = (size_t)(n ? n : 1) * 3 * sizeof(double);
    // (from the context's stream-ordered pool, like every other block of a cloud: a drop-in call uploads a cloud and frees it
    // again, and hipMalloc / hipFree -- the latter a device-wide synchronisation -- cost it about a millisecond each way)
    SF_CHECK(sf_pool_alloc(ctx, bytes, (void **)&c->xyz_orig));
    SF_CHECK(sf_pool_alloc(ctx, (size_t)(n ? n : 1) * sizeof(int32_t), (void **)&c->zperm));
    SF_CHECK(sf_pool_alloc(ctx, (size_t)(n + 2) * sizeof(double), (void **)&c->z_orig));
    if (n) {
        // ---- the z-sorted internal order: sort (z, caller index) once, stable ----
        sf_pool_guard tmp(ctx);
        const double *dsrc = xyz;
        if (!(flags & SF_IN_DEVICE)) {
            double *stage = nullptr;
            SF_CHECK(tmp.alloc(&stage, (size_t)n * 3));
            SF_HIP(hipMemcpyAsync(stage, xyz, (size_t)n * 24, hipMemcpyHostToDevice, ctx->stream));
            dsrc = stage;
        }
        double *zkey = nullptr;
        int32_t *idx = nullptr;
        SF_CHECK(tmp.alloc(&zkey, (size_t)n));
        SF_CHECK(tmp.alloc(&idx, (size_t)n));
        SF_LAUNCH(ctx, "k0_upload", k_upload_keys, dim3((unsigned)sf_div_up(n, 256)), dim3(256), dsrc, n, zkey, idx);
        size_t tb = 0;
        SF_HIP(rocprim::radix_sort_pairs<sf_sort_config>(nullptr, tb, zkey, c->z_orig, idx, c->zperm, (size_t)n, 0, 64, ctx->stream));
        char *stmp = nullptr;
        SF_CHECK(tmp.alloc(&stmp, tb ? tb : 8));
        {
            sf_launch_timer t_(ctx, "k0_upload");
            SF_HIP(rocprim::radix_sort_pairs<sf_sort_config>(stmp, tb, zkey, c->z_orig, idx, c->zperm, (size_t)n, 0, 64, ctx->stream));
        }
        SF_LAUNCH(ctx, "k0_upload", k_upload_gather, dim3((unsigned)sf_div_up(n, 256)), dim3(256), dsrc, (const int32_t *)c->zperm, n,
                  c->xyz_orig);
        SF_HIP(hipStreamSynchronize(ctx->stream));
    }
    if (normals) {
        SF_CHECK(sf_pool_alloc(ctx, bytes, (void **)&c->nrm_orig));
        SF_CHECK(upload_in_zorder(ctx, c, normals, flags, c->nrm_orig));
    }
    return SF_OK;
}

extern "C" sf_cloud *sf_cloud_upload(sf_ctx *ctx, const double *xyz, const double *normals, int64_t n, int flags)
{
    if (!ctx || (!xyz && n > 0) || n < 0 || n > 2147483000LL) {
        sf_set_error("sf_cloud_upload: bad arguments (n=%lld)", (long long)n);
        return nullptr;
    }
    SF_HIP_NULL(hipSetDevice(ctx->device));
    sf_cloud *c = new sf_cloud();
    c->n = n;
    if (cloud_upload(ctx, c, xyz, normals, flags) != SF_OK) {
        sf_cloud_free(ctx, c);
        return nullptr;
    }
    return c;
}

extern "C" int sf_cloud_set_normals(sf_ctx *ctx, sf_cloud *c, const double *normals, int flags)
{
    if (!ctx || !c || !normals) { sf_set_error("sf_cloud_set_normals: null argument"); return SF_ERR_ARG; }
    SF_HIP(hipSetDevice(ctx->device));
    size_t bytes = (size_t)(c->n ? c->n : 1) * 24;
    if (!c->nrm_orig) SF_CHECK(sf_pool_alloc(ctx, bytes, (void **)&c->nrm_orig));
    SF_CHECK(upload_in_zorder(ctx, c, normals, flags, c->nrm_orig));
    c->normals_sorted = false;
    c->nrm_max2 = -1.0;
    return SF_OK;
}

namespace {
// max over the normals of |n|^2 (non-finite components count as +inf): non-negative doubles order like their bit patterns
__global__ void k_normals_max2(const double *__restrict__ nrm, int64_t n, unsigned long long *__restrict__ out)
{
    double mx = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const double a = nrm[3 * i], b = nrm[3 * i + 1], c = nrm[3 * i + 2];
        const double v = (a * a + b * b) + c * c;
        mx = v <= 1.7976931348623157e308 ? fmax(mx, v) : INFINITY; // NaN / inf -> inf
    }
    for (int off = 32; off > 0; off >>= 1) mx = fmax(mx, __shfl_xor(mx, off));
    if ((threadIdx.x & 63) == 0) atomicMax(out, (unsigned long long)__double_as_longlong(mx));
}
} // namespace

// Largest squared norm among the cloud's normals, computed once per set of normals (one small kernel + read-back).
// K6 uses it: with unit normals |alpha| <= |p_j - p_i| <= radius, which pins alpha's bin when the radius is small.
int sf_cloud_normals_max2(sf_ctx *ctx, sf_cloud *c, double *out)
{
    if (!c->nrm_orig) { sf_set_error("this operation needs normals, but the cloud has none"); return SF_ERR_STATE; }
    if (c->nrm_max2 < 0.0) {
        sf_pool_guard tmp(ctx); // (the block goes back to the pool on every exit)
        unsigned long long *d = nullptr;
        void *pin = nullptr;
        SF_CHECK(tmp.alloc(&d, (size_t)1));
        SF_CHECK(sf_ctx_pinned(ctx, &pin));
        SF_HIP(hipMemsetAsync(d, 0, sizeof(unsigned long long), ctx->stream));
        if (c->n) SF_LAUNCH(ctx, "k1_normals_max", k_normals_max2, dim3(512), dim3(256), (const double *)c->nrm_orig, c->n, d);
        SF_HIP(hipMemcpyAsync(pin, d, sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream));
        SF_HIP(hipStreamSynchronize(ctx->stream));
        long long bits = *(const long long *)pin;
        double v;
        static_assert(sizeof(v) == sizeof(bits), "");
        memcpy(&v, &bits, sizeof(v));
        c->nrm_max2 = v;
    }
    *out = c->nrm_max2;
    return SF_OK;
}

// inv_perm on first use after a grid build (see k_inv_perm)
int sf_cloud_ensure_inv_perm(sf_ctx *ctx, sf_cloud *c)
{
    if (c->inv_perm_valid) return SF_OK;
    if (!c->perm || !c->inv_perm) { sf_set_error("grid not built"); return SF_ERR_STATE; }
    const int64_t np = c->pop_end - c->pop_begin;
    if (np > 0)
        SF_LAUNCH(ctx, "k1_inv_perm", k_inv_perm, dim3((unsigned)sf_div_up(np, 256)), dim3(256), (const int32_t *)c->perm,
                  c->pop_begin, np, c->inv_perm);
    c->inv_perm_valid = true;
    return SF_OK;
}

// Sort the normals into cell order on first use (lazy: a normals-less cloud never pays for it).
int sf_cloud_ensure_sorted_normals(sf_ctx *ctx, sf_cloud *c)
{
    if (c->normals_sorted) return SF_OK;
    if (!c->nrm_orig) { sf_set_error("this operation needs normals, but the cloud has none"); return SF_ERR_STATE; }
    if (!c->perm) { sf_set_error("grid not built"); return SF_ERR_STATE; }
    const int64_t np = c->pop_end - c->pop_begin;
    if (np > 0) {
        SF_LAUNCH(ctx, "k1_gather_normals", k_gather_normals, dim3((unsigned)sf_div_up(np, 256)), dim3(256),
                  c->nrm_orig, c->perm_int, c->pop_begin, np, c->rec);
        // The flag below is per cloud, not per stream: while the context is forked (sf_fork) a consumer on the OTHER
        // stream would see it set and read rec[3..5] with nothing ordering that read after this gather.  Make the
        // other stream wait for it.
        hipStream_t other = ctx->stream == ctx->streams[0] ? ctx->streams[1] : ctx->streams[0];
        SF_HIP(hipEventRecord(ctx->join_event, ctx->stream));
        SF_HIP(hipStreamWaitEvent(other, ctx->join_event, 0));
    }
    c->normals_sorted = true;
    return SF_OK;
}

// min / max of n x 3 coordinates on the device (block partials, folded by one more block; 6 values read back through
// page-locked memory)
int sf_cloud_bbox_raw(sf_ctx *ctx, const double *xyz_dev, int64_t n, double lo[3], double hi[3])
{
    for (int a = 0; a < 3; ++a) lo[a] = hi[a] = 0.0;
    if (!n) return SF_OK;
    const int nb = 1024;
    void *scr = nullptr;
    SF_CHECK(sf_ctx_scratch(ctx, ((size_t)nb * 6 + 8) * sizeof(double), &scr));
    SF_LAUNCH(ctx, "k1_bbox", k_bbox_partial, dim3(nb), dim3(256), xyz_dev, n, (double *)scr);
    double *fin = (double *)scr + (size_t)nb * 6;
    SF_LAUNCH(ctx, "k1_bbox", k_bbox_final, dim3(1), dim3(384), (const double *)scr, nb, fin);
    void *pin = nullptr;
    SF_CHECK(sf_ctx_pinned(ctx, &pin));
    double *part = (double *)pin;
    SF_HIP(hipMemcpyAsync(part, fin, 6 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    SF_HIP(hipStreamSynchronize(ctx->stream));
    for (int a = 0; a < 3; ++a) { lo[a] = part[a]; hi[a] = part[3 + a]; }
    for (int a = 0; a < 3; ++a)
        if (!std::isfinite(lo[a]) || !std::isfinite(hi[a])) {
            sf_set_error("cloud has non-finite coordinates");
            return SF_ERR_ARG;
        }
    return SF_OK;
}

int sf_cloud_bbox(sf_ctx *ctx, sf_cloud *c, double lo[3], double hi[3])
{
    if (!c->bbox_known) {
        SF_CHECK(sf_cloud_bbox_raw(ctx, c->xyz_orig, c->n, c->bbox_lo, c->bbox_hi));
        c->bbox_known = true; // (sf_cloud_upload is the only writer of xyz_orig)
    }
    for (int a = 0; a < 3; ++a) { lo[a] = c->bbox_lo[a]; hi[a] = c->bbox_hi[a]; }
    return SF_OK;
}

// Common grid build.  block_end < 0: the whole cloud.  Otherwise only the z-layers of cells that the queries at
// cell-sorted positions [block_begin, block_end) can reach within `reach` cells are sorted and gathered -- the
// positions, perm and cell_start keep their GLOBAL numbering (a layer's first position is the number of points in
// the layers below it, known from a per-layer histogram), so everything downstream is unchanged and the result of
// any query is bit-identical to a whole-cloud build.  A rank of an N-GPU job pays for ~1/N of the sort and gather
// instead of all of it; only the streaming passes (bounding box, cell ids, layer histogram, selection) see every point.
static int build_grid(sf_ctx *ctx, sf_cloud *c, double cell, int64_t block_begin, int64_t block_end, int reach)
{
    SF_HIP(hipSetDevice(ctx->device));
    cloud_release_grid(ctx, c);
    ++c->grid_gen; // (list sets made on the previous build are refused from here on: sf_nbrs_on_grid)
    const int64_t n = c->n;
    // ---- bounding box -------------------------------------------------------------------------
    double lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};
    SF_CHECK(sf_cloud_bbox(ctx, c, lo, hi));
    // ---- grid geometry: edge slightly above `cell` so that |a-b| <= cell never spans 2 cells ----
    double edge = cell * (1.0 + 9.5367431640625e-07); // 1 + 2^-20
    int64_t ncell;
    for (;;) {
        double tot = 1.0;
        bool too_long = false;
        for (int a = 0; a < 3; ++a) {
            double d = std::floor((hi[a] - lo[a]) / edge) + 1.0;
            if (!(d >= 1.0)) d = 1.0;
            if (d > 2097152.0) too_long = true;
            c->dim[a] = (int)std::min(d, 2097152.0);
            tot *= d;
        }
        if (!too_long && tot <= 67108864.0) { ncell = (int64_t)tot; break; }
        edge *= 2.0; // coarser cells stay correct (edge >= radius), only slower
    }
    // Along x the cells are cut xsub times finer (up to 16, SF_XSUB overrides; a power of two, so a fine cell index / xsub is exactly the index of
    // the edge-sized cell): the radius search clips each of its nine runs to the x range its ball can reach in that
    // row of cells (search.hip), which needs cell boundaries every edge / xsub.  Rows stay contiguous runs of positions
    // and z-layers stay slabs, so nothing else changes but the row stride.
    c->xsub = 1;
    static const int xsub_max = [] { const char *e = getenv("SF_XSUB"); const int v = e ? atoi(e) : 16; return v >= 1 && v <= 16 ? v : 16; }();
    while (c->xsub < xsub_max && (double)ncell * 2.0 <= 67108864.0 && c->dim[0] * 2 <= 2097152) {
        c->xsub *= 2;
        c->dim[0] *= 2;
        ncell *= 2;
    }
    c->cell = edge;
    c->inv_cell = 1.0 / edge;
    c->ncell = ncell;
    for (int a = 0; a < 3; ++a) c->lo[a] = lo[a];

    size_t nn = (size_t)(n ? n : 1);
    SF_CHECK(sf_palloc(ctx, &c->cell_start, (size_t)(ncell + 1)));
    SF_CHECK(sf_palloc(ctx, &c->perm, nn));
    SF_CHECK(sf_palloc(ctx, &c->perm_int, nn));
    SF_CHECK(sf_palloc(ctx, &c->inv_perm, nn));
    SF_CHECK(sf_palloc(ctx, &c->xs, nn + 2)); // +2: K2 reads candidates in pairs (one element past the end)
    SF_CHECK(sf_palloc(ctx, &c->ys, nn + 2));
    SF_CHECK(sf_palloc(ctx, &c->zs, nn + 2));
    SF_CHECK(sf_palloc(ctx, &c->rec, nn * 6 + 2));
    if (!n) {
        SF_HIP(hipMemsetAsync(c->cell_start, 0, (size_t)(ncell + 1) * sizeof(int32_t), ctx->stream));
        SF_HIP(hipStreamSynchronize(ctx->stream));
        return SF_OK;
    }
    // ---- cell ids, stable radix sort (ties keep ascending original index), SoA gather ---------
    sf_pool_guard tmp(ctx);
    sf_grid_desc g = sf_make_grid_desc(c);
    int bits = 1;
    while (((int64_t)1 << bits) < ncell) ++bits;
    int64_t cid_base = 0;                      // the sorted keys are cell id - cid_base

    int64_t base = 0, ns = n;                  // populated slice [base, base + ns) of the global order
    int64_t slab_cells = ncell;                // ... and the cells it can fall into: [cid_base, cid_base + slab_cells)
    const bool whole = block_end < 0;
    if (!whole) {
        // ---- which z-layers does the block need?  The internal order is ascending in z (sf_cloud), so every layer is a run of
        //      it and a layer's first point is a binary search away: nl + 1 searches, one small read-back. ----
        const int nl = c->dim[2];
        std::vector<int64_t> &first = c->layer_first; // global position (= internal index) of each layer's first point
        first.assign((size_t)nl + 1, 0);
        const bool on_host = n <= SF_Z_HOST_MAX && nl <= SF_Z_HOST_LAYERS && !getenv("SF_K1_DEVICE_BOUNDS");
        if (on_host) {
            if ((int64_t)c->z_host.size() != n) { // (once per cloud)
                c->z_host.resize((size_t)n);
                SF_HIP(hipMemcpyAsync(c->z_host.data(), c->z_orig, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
                SF_HIP(hipStreamSynchronize(ctx->stream));
            }
            // the layer of a coordinate is monotone in it: lower bounds, each searched from the previous one (a galloping step
            // first: consecutive layers are ~n / nl points apart, so the probes stay close together)
            const double *z = c->z_host.data();
            const double lo_z = c->lo[2], inv = c->inv_cell;
            int64_t a = 0;
            for (int l = 1; l <= nl; ++l) {
                int64_t step = 1, hi = a; // first i >= a with layer(z[i]) >= l
                while (hi < n && sf_cell_coord(z[hi], lo_z, inv, nl) < l) { a = hi + 1; hi += step; step *= 2; }
                if (hi > n) hi = n;
                while (a < hi) {
                    const int64_t mid = a + (hi - a) / 2;
                    if (sf_cell_coord(z[mid], lo_z, inv, nl) < l) a = mid + 1; else hi = mid;
                }
                first[(size_t)l] = l == nl ? n : a;
            }
        } else {
            int64_t *dfirst = nullptr;
            SF_CHECK(tmp.alloc(&dfirst, (size_t)nl + 1));
            SF_LAUNCH(ctx, "k1_layer_bounds", k_layer_bounds, dim3((unsigned)sf_div_up(nl + 1, 4)), dim3(256), (const double *)c->z_orig, n,
                      c->lo[2], c->inv_cell, nl, dfirst);
            void *pin = nullptr;
            SF_CHECK(sf_ctx_pinned(ctx, &pin));
            int64_t *dst = ((size_t)nl + 1) * sizeof(int64_t) <= SF_PINNED_BYTES ? (int64_t *)pin : first.data();
            SF_HIP(hipMemcpyAsync(dst, dfirst, ((size_t)nl + 1) * sizeof(int64_t), hipMemcpyDeviceToHost, ctx->stream));
            SF_HIP(hipStreamSynchronize(ctx->stream));
            if (dst != first.data()) std::copy(dst, dst + nl + 1, first.begin());
        }
        int zb = 0, ze = nl - 1;
        if (block_begin < block_end) {
            while (zb + 1 < nl && first[(size_t)zb + 1] <= block_begin) ++zb;          // layer holding position block_begin
            ze = zb;
            while (ze + 1 < nl && first[(size_t)ze + 1] <= block_end - 1) ++ze;        // layer holding position block_end - 1
        } else {
            zb = 0; ze = -1; // empty block: populate nothing
        }
        const int zlo = std::max(zb - reach, 0), zhi = block_begin < block_end ? std::min(ze + reach, nl - 1) : -1;
        base = zhi >= zlo ? first[(size_t)zlo] : 0;
        ns = zhi >= zlo ? first[(size_t)zhi + 1] - base : 0;
        // ---- the slab's points are the internal points [base, base + ns): their cell ids, nothing else of the cloud is read ----
        if (ns > 0) {
            // keys relative to the slab's first cell: 19 bits instead of 23 at config 5 -- two sort passes instead of three
            const int64_t layer_cells = (int64_t)c->dim[0] * c->dim[1];
            cid_base = (int64_t)zlo * layer_cells;
            slab_cells = (int64_t)(zhi - zlo + 1) * layer_cells;
            bits = 1;
            while (((int64_t)1 << bits) < slab_cells) ++bits;
        }
    }
    c->pop_begin = base;
    c->pop_end = base + ns;
    // (the grid is built on the context's current stream, and a fork (sf_fork) orders the side stream after everything
    // issued before it, so this flag needs no event of its own -- unlike the lazy gather of sf_cloud_ensure_sorted_normals)
    c->normals_sorted = c->nrm_orig != nullptr;
    const bool force_radix = getenv("SF_K1_RADIX") != nullptr; // (read at every build: the tests compare the two paths in one process)
    // (the counting build ranks every point among the points of its cell, pop compares per point: grids with a handful of points per
    // cell only -- a mean population above 64 goes to the sort, whose cost does not depend on how the points pile up)
    if (ns > 0 && !force_radix && slab_cells <= SF_COUNT_CELLS_PER_POINT * ns && ns <= 64 * slab_cells) {
        // ---- the counting build (see k_cell_count): fill, count, scan, place, settle ----
        const int64_t slab_tiles = sf_div_up(slab_cells, (int64_t)SF_SCAN_TILE), n_status = slab_tiles + 1;
        const int64_t fill_blocks = sf_div_up(ncell + 1 - slab_cells, (int64_t)SF_SCAN_TILE);
        int32_t *cid = nullptr, *place = nullptr;
        int2 *slot = nullptr;
        unsigned long long *status = nullptr;
        SF_CHECK(tmp.alloc(&cid, (size_t)ns));
        SF_CHECK(tmp.alloc(&place, (size_t)ns));
        SF_CHECK(tmp.alloc(&slot, (size_t)ns));
        SF_CHECK(tmp.alloc(&status, (size_t)n_status));
        int32_t *start_rel = c->cell_start + cid_base;
        // (the counters are zero between builds: k_cell_scan writes them back.  A build that did not get as far as its scan --
        // a launch that failed, an error left on the device by whatever ran before -- leaves them as they are: zeroed again)
        if (hipPeekAtLastError() != hipSuccess) c->count_dirty = true;
        if (c->cell_count_cap < slab_cells || c->count_dirty) { // (first build of this cloud with a slab this large)
            if (c->cell_count_cap < slab_cells) {
                if (c->cell_count) sf_pool_release(ctx, c->cell_count);
                c->cell_count = nullptr;
                c->cell_count_cap = 0;
                SF_CHECK(sf_pool_alloc(ctx, (size_t)slab_cells * sizeof(int32_t), (void **)&c->cell_count));
                c->cell_count_cap = slab_cells;
            }
            sf_launch_timer t_(ctx, "k1_cell_zero");
            SF_HIP(hipMemsetAsync(c->cell_count, 0, (size_t)c->cell_count_cap * sizeof(int32_t), ctx->stream));
        }
        c->count_dirty = true;
        SF_LAUNCH(ctx, "k1_cell_count", k_cell_count, dim3((unsigned)sf_div_up(std::max(ns, n_status), 256)), dim3(256), c->xyz_orig, base, ns, g,
                  (int)cid_base, c->cell_count, cid, place, status, n_status);
        SF_LAUNCH(ctx, "k1_cell_scan", k_cell_scan, dim3((unsigned)(slab_tiles + fill_blocks)), dim3(SF_SCAN_TPB), c->cell_count, c->cell_start, ncell,
                  cid_base, slab_cells, slab_tiles, (int32_t)base, (int32_t)ns, status);
        c->count_dirty = false;
        SF_LAUNCH(ctx, "k1_cell_place", k_cell_place, dim3((unsigned)sf_div_up(ns, 256)), dim3(256), (const int32_t *)cid,
                  (const int32_t *)place, (const int32_t *)start_rel, base, ns, slot);
        SF_LAUNCH(ctx, "k1_cell_settle", k_cell_settle, dim3((unsigned)sf_div_up(ns, 256)), dim3(256), (const int2 *)slot,
                  (const int32_t *)start_rel, c->xyz_orig, (const double *)c->nrm_orig, (const int32_t *)c->zperm, c->perm, c->perm_int, base, ns,
                  c->xs, c->ys, c->zs, c->rec);
        return SF_OK;
    }
    // ---- sparse grids: cell ids, stable radix sort, gather, cell table from the sorted ids ----
    int32_t *key_in = nullptr, *val_in = nullptr, *cid_sorted = nullptr; // what gets sorted; the sorted ids
    SF_CHECK(tmp.alloc(&cid_sorted, (size_t)std::max<int64_t>(ns, 1)));
    if (ns > 0) {
        SF_CHECK(tmp.alloc(&key_in, (size_t)ns + 1));
        SF_CHECK(tmp.alloc(&val_in, (size_t)ns + 1));
        SF_LAUNCH(ctx, "k1_cell_ids", k_cell_ids, dim3((unsigned)sf_div_up(ns, 256)), dim3(256), c->xyz_orig, base, ns, g, key_in, val_in,
                  (int)cid_base);
    }
    sf_gap *gaps = nullptr; // (the cell-table kernel's list of long runs of empty cells)
    unsigned *n_gaps = nullptr;
    SF_CHECK(tmp.alloc(&gaps, (size_t)(ncell / SF_LONG_GAP + 2)));
    SF_CHECK(tmp.alloc(&n_gaps, 1));
    if (ns > 0) {
        size_t tmp_bytes = 0;
        SF_HIP(sort_cells(nullptr, tmp_bytes, key_in, cid_sorted, val_in, c->perm_int + base, (size_t)ns, bits, ctx->stream));
        char *stmp = nullptr;
        SF_CHECK(tmp.alloc(&stmp, tmp_bytes ? tmp_bytes : 8));
        {
            sf_launch_timer t_(ctx, "k1_radix_sort");
            SF_HIP(sort_cells(stmp, tmp_bytes, key_in, cid_sorted, val_in, c->perm_int + base, (size_t)ns, bits, ctx->stream));
        }
        SF_LAUNCH(ctx, "k1_gather_sorted", k_gather_sorted, dim3((unsigned)sf_div_up(ns, 256)), dim3(256), c->xyz_orig,
                  (const double *)c->nrm_orig, (const int32_t *)c->perm_int, (const int32_t *)c->zperm, c->perm, base, ns, c->xs, c->ys, c->zs,
                  c->rec, n_gaps);
    }
    {
        if (ns <= 0) SF_HIP(hipMemsetAsync(n_gaps, 0, sizeof(unsigned), ctx->stream)); // (else zeroed by the gather kernel)
        SF_LAUNCH(ctx, "k1_cell_start", k_cell_start, dim3((unsigned)sf_div_up(ns + 1, 256)), dim3(256), cid_sorted, cid_base, base, ns,
                  ncell, c->cell_start, gaps, n_gaps);
        SF_LAUNCH(ctx, "k1_cell_fill_long", k_cell_fill_long, dim3(1024), dim3(256), (const sf_gap *)gaps, (const unsigned *)n_gaps,
                  c->cell_start);
    }
    return SF_OK;
}

// First cell-sorted position of every z-layer of cells, dim[2] + 1 entries: kept by the block build, read off the cell
// table (one small kernel + copy) after a whole-cloud build.
static int cloud_layer_first(sf_ctx *ctx, sf_cloud *c)
{
    if (!c->cell_start) { sf_set_error("grid not built"); return SF_ERR_STATE; }
    const int nl = c->dim[2];
    if ((int64_t)c->layer_first.size() == (int64_t)nl + 1) return SF_OK;
    sf_pool_guard tmp(ctx);
    int64_t *d = nullptr;
    SF_CHECK(tmp.alloc(&d, (size_t)nl + 1));
    SF_LAUNCH(ctx, "k1_layer_first", k_layer_first, dim3((unsigned)sf_div_up(nl + 1, 256)), dim3(256), (const int32_t *)c->cell_start,
              (int64_t)c->dim[0] * c->dim[1], nl, d);
    c->layer_first.assign((size_t)nl + 1, 0);
    SF_HIP(hipMemcpyAsync(c->layer_first.data(), d, ((size_t)nl + 1) * sizeof(int64_t), hipMemcpyDeviceToHost, ctx->stream));
    SF_HIP(hipStreamSynchronize(ctx->stream));
    return SF_OK;
}

extern "C" int sf_cloud_layer_table(sf_ctx *ctx, sf_cloud *c, int64_t *first, int64_t cap, int64_t *n_layers)
{
    if (!ctx || !c || !n_layers) { sf_set_error("sf_cloud_layer_table: null argument"); return SF_ERR_ARG; }
    SF_CHECK(cloud_layer_first(ctx, c));
    *n_layers = c->dim[2];
    if (first) {
        if (cap < (int64_t)c->layer_first.size()) { sf_set_error("sf_cloud_layer_table: room for %lld entries, %zu needed", (long long)cap, c->layer_first.size()); return SF_ERR_ARG; }
        std::copy(c->layer_first.begin(), c->layer_first.end(), first);
    }
    return SF_OK;
}

extern "C" int sf_cloud_build_grid(sf_ctx *ctx, sf_cloud *c, double cell)
{
    if (!ctx || !c || !(cell > 0.0) || !std::isfinite(cell)) {
        sf_set_error("sf_cloud_build_grid: bad arguments (cell=%g)", cell);
        return SF_ERR_ARG;
    }
    return build_grid(ctx, c, cell, 0, -1, 0);
}

extern "C" int sf_cloud_build_grid_block(sf_ctx *ctx, sf_cloud *c, double cell, int64_t begin, int64_t end, int reach,
                                         int64_t *pop_begin, int64_t *pop_end)
{
    if (!ctx || !c || !(cell > 0.0) || !std::isfinite(cell) || begin < 0 || end > c->n || begin > end || reach < 0) {
        sf_set_error("sf_cloud_build_grid_block: bad arguments (cell=%g, block [%lld, %lld), reach %d)", cell,
                     (long long)begin, (long long)end, reach);
        return SF_ERR_ARG;
    }
    SF_CHECK(build_grid(ctx, c, cell, begin, end, reach));
    if (pop_begin) *pop_begin = c->pop_begin;
    if (pop_end) *pop_end = c->pop_end;
    return SF_OK;
}

extern "C" int sf_cloud_perm(sf_ctx *ctx, sf_cloud *c, int32_t *perm)
{
    if (!ctx || !c || !perm) { sf_set_error("sf_cloud_perm: null argument"); return SF_ERR_ARG; }
    if (!c->perm) { sf_set_error("sf_cloud_perm: grid not built"); return SF_ERR_STATE; }
    if (c->n) SF_HIP(hipMemcpyAsync(perm, c->perm, (size_t)c->n * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    SF_HIP(hipStreamSynchronize(ctx->stream));
    return SF_OK;
}

extern "C" int sf_cloud_halo_range(sf_ctx *ctx, sf_cloud *c, int64_t begin, int64_t end, int64_t *hb, int64_t *he)
{
    if (!ctx || !c || !hb || !he || begin < 0 || end > c->n || begin > end) {
        sf_set_error("sf_cloud_halo_range: bad arguments");
        return SF_ERR_ARG;
    }
    if (!c->cell_start) { sf_set_error("sf_cloud_halo_range: grid not built"); return SF_ERR_STATE; }
    if (begin == end) { *hb = begin; *he = end; return SF_OK; }
    // the z-layers of cells holding the block's first and last position, one more layer on either side
    SF_CHECK(cloud_layer_first(ctx, c));
    const std::vector<int64_t> &first = c->layer_first;
    const int nl = c->dim[2];
    int zb = 0;
    while (zb + 1 < nl && first[(size_t)zb + 1] <= begin) ++zb;
    int ze = zb;
    while (ze + 1 < nl && first[(size_t)ze + 1] <= end - 1) ++ze;
    *hb = std::min<int64_t>(first[(size_t)std::max(zb - 1, 0)], begin);
    *he = std::max<int64_t>(first[(size_t)std::min(ze + 2, nl)], end);
    return SF_OK;
}

extern "C" int64_t sf_cloud_size(const sf_cloud *c) { return c ? c->n : -1; }

extern "C" void sf_cloud_free(sf_ctx *ctx, sf_cloud *c)
{
    if (!c) return;
    if (ctx) (void)hipStreamSynchronize(ctx->stream);
    cloud_release_grid(ctx, c);
    for (void *p : {(void *)c->xyz_orig, (void *)c->nrm_orig, (void *)c->z_orig, (void *)c->cell_count, (void *)c->zperm})
        if (p) { if (ctx) sf_pool_release(ctx, p); else (void)hipFree(p); }
    delete c;
}
