// voxel.hip -- voxel-grid subsampling on the device.
//
// Replaces: grid_subsampling                            core/subsampling.py:5-39
//           the voxel loop of select_keypoints_with_density_threshold   keypoint_selection.py:80-101
// (the step in front of the default SHOT configuration: pipeline.py:293, shot_parallelization.py:157-161).
//
// The reference: keys = ((p - min(p)) // voxel).astype(int); np.unique(keys, axis=0, return_inverse, return_counts);
// the points of a voxel in the order np.argsort(inverse) yields; per voxel the point closest to the voxel's barycentre
// (first minimum).  Voxels come out in np.unique's lexicographic key order.
//
// Here: V1 keys (NumPy's floor_divide evaluated step for step, so every key equals the reference's) packed
// kx:ky:kz into one 64-bit word whose integer order IS the lexicographic order; V2 stable radix sort (rocPRIM) --
// within a voxel the sorted order is ascending point index; V3 run heads -> voxel rank of every point (`inverse`) and the
// run starts; V4 one thread per voxel: sequential barycentre (np.mean over axis 0 adds the rows one after the other),
// distances, first minimum.  The ONLY thing the reference leaves undefined is the order of a voxel's points: its
// np.argsort(inverse) is an unstable sort, and two-point voxels tie exactly (both points are equally far from their
// midpoint), so which point it returns there depends on NumPy's sort implementation.  V4 therefore takes the visiting
// order as an input: ascending index (device only, platform independent), or any permutation the caller supplies
// (the Python layer passes np.argsort(inverse) -- the reference's own call on the reference's own array -- which makes
// the result identical to the reference's on the same NumPy build, ties included).
// HBM roofline: n x (24 B in + 8 B key + 4 B index, sorted once) + 24 B gather per point in V4; a few hundred us at 1M.
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include <cmath>

#include "common.h"
#include "device_util.h"

struct sf_voxels {
    int64_t n = 0, nvox = 0;
    double voxel = 0.0;
    double lo[3] = {0, 0, 0};
    const double *xyz = nullptr; // device, n x 3 (owned unless borrowed)
    bool owns_xyz = false;
    int32_t *perm = nullptr;     // n: points grouped by voxel (key order), ascending index inside a voxel
    int32_t *rank = nullptr;     // n: voxel rank of sorted element i
    int32_t *start = nullptr;    // nvox + 1: first sorted element of every voxel
    int32_t max_pop = 0;         // points of the fullest voxel (decides whether the wave-per-voxel pass has anything to do)
};

// (no tuned Onesweep configuration for gfx950 in rocPRIM 4.2: see grid.hip)
using sf_vox_onesweep = rocprim::radix_sort_onesweep_config<rocprim::kernel_config<1024, 4>, rocprim::kernel_config<1024, 4>, 10,
                                                            rocprim::block_radix_rank_algorithm::match>;
using sf_vox_sort_config = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config, sf_vox_onesweep, 65536>;

namespace {

// numpy's floor_divide for float64 (npy_floor_divide / npy_divmod), a >= 0 or not, b != 0
__device__ inline double np_floor_divide(double a, double b)
{
    double mod = fmod(a, b);
    double div = (a - mod) / b;
    if (mod != 0.0) {
        if ((b < 0.0) != (mod < 0.0)) {
            mod += b;
            div -= 1.0;
        }
    }
    double fd;
    if (div != 0.0) {
        fd = floor(div);
        if (div - fd > 0.5) fd += 1.0;
    } else {
        fd = copysign(0.0, a / b);
    }
    return fd;
}

__global__ void k_voxel_keys(const double *__restrict__ xyz, int64_t n, double lx, double ly, double lz, double voxel,
                             int by, int bz, unsigned long long *__restrict__ key, int32_t *__restrict__ val,
                             int *__restrict__ bad)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double kx = np_floor_divide(xyz[3 * i + 0] - lx, voxel);
    const double ky = np_floor_divide(xyz[3 * i + 1] - ly, voxel);
    const double kz = np_floor_divide(xyz[3 * i + 2] - lz, voxel);
    // the host sized the three fields from the extent; anything outside (NaN, overflow) is reported, not wrapped
    const double mx = (double)(1ull << (63 - by - bz)), my = (double)(1ull << by), mz = (double)(1ull << bz);
    if (!(kx >= 0.0 && kx < mx && ky >= 0.0 && ky < my && kz >= 0.0 && kz < mz)) {
        atomicOr(bad, 1);
        key[i] = ~0ull;
    } else {
        key[i] = ((unsigned long long)kx << (by + bz)) | ((unsigned long long)ky << bz) | (unsigned long long)kz;
    }
    val[i] = (int32_t)i;
}

__global__ void k_voxel_heads(const unsigned long long *__restrict__ skey, int64_t n, int32_t *__restrict__ head)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) head[i] = (i == 0 || skey[i] != skey[i - 1]) ? 1 : 0;
}

// rank[i] = (inclusive scan of head)[i] - 1; start[rank] = i at run heads; start[nvox] = n
__global__ void k_voxel_starts(const int32_t *__restrict__ head, const int32_t *__restrict__ scan, int64_t n,
                               int32_t *__restrict__ rank, int32_t *__restrict__ start)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int32_t r = scan[i] - 1;
    rank[i] = r;
    if (head[i]) start[r] = (int32_t)i;
    if (i == n - 1) start[r + 1] = (int32_t)n;
}

// points of the fullest voxel: bad[1] = max over run heads of (next run's first element - this one's)
__global__ void k_voxel_maxpop(const int32_t *__restrict__ head, const int32_t *__restrict__ rank, const int32_t *__restrict__ start,
                               int64_t n, int *__restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int len = 0;
    if (i < n && head[i]) len = start[rank[i] + 1] - (int32_t)i;
    for (int off = 32; off > 0; off >>= 1) len = max(len, __shfl_xor(len, off));
    if ((threadIdx.x & 63) == 0 && len > 0) atomicMax(out, len);
}

__global__ void k_voxel_inverse(const int32_t *__restrict__ perm, const int32_t *__restrict__ rank, int64_t n,
                                int64_t *__restrict__ inverse)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) inverse[perm[i]] = rank[i];
}

// order (nullable): a permutation of the points grouped by voxel in voxel order -- the elements [start[v], start[v+1])
// of it are voxel v's points in the caller's visiting order; null = this library's own (perm).  An element outside
// [0, n) is never dereferenced: the voxel's answer becomes -1 and the context's flag is raised (common.h).
// Small voxels (the usual case: a handful of points, millions of voxels): one THREAD per voxel, k_voxel_select.
// Voxels of more than SF_VOXEL_WAVE points (a voxel size chosen large against the cloud's extent: a few voxels hold
// everything): one WAVE per voxel, k_voxel_select_wave -- a lone thread walking 10^6 dependent, scattered loads twice takes
// seconds.  The barycentre must still be the sequential row-by-row sum np.mean(axis=0) forms (subsampling.py:31): the
// wave stages 64 points at a time in LDS with coalesced gathers and lanes 0..2 add the x / y / z columns in order; the
// distances and the first minimum in visiting order are lane-parallel.
#define SF_VOXEL_WAVE 32
__global__ void k_voxel_select(const double *__restrict__ xyz, int64_t n, const int32_t *__restrict__ start, int64_t nvox,
                               const int32_t *__restrict__ perm, const int64_t *__restrict__ order,
                               int64_t *__restrict__ selected, int64_t *__restrict__ counts, volatile int *__restrict__ flag)
{
    const int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= nvox) return;
    const int32_t s = start[v], e = start[v + 1];
    if (counts) counts[v] = e - s;
    if (e - s > SF_VOXEL_WAVE) return; // k_voxel_select_wave's
    double sx = 0.0, sy = 0.0, sz = 0.0;
    bool bad = false;
    for (int32_t t = s; t < e; ++t) { // np.mean(axis=0): rows added one after the other (subsampling.py:31)
        const int64_t j = order ? order[t] : (int64_t)perm[t];
        if (j < 0 || j >= n) { bad = true; continue; }
        sx += xyz[3 * j + 0];
        sy += xyz[3 * j + 1];
        sz += xyz[3 * j + 2];
    }
    if (bad) {
        *flag = SF_FLAG_VOXEL_ORDER;
        selected[v] = -1;
        return;
    }
    const double k = (double)(e - s);
    const double mx = sx / k, my = sy / k, mz = sz / k;
    double best = INFINITY;
    int64_t arg = -1;
    for (int32_t t = s; t < e; ++t) {
        const int64_t j = order ? order[t] : (int64_t)perm[t];
        const double dx = xyz[3 * j + 0] - mx, dy = xyz[3 * j + 1] - my, dz = xyz[3 * j + 2] - mz;
        const double d = sqrt((dx * dx + dy * dy) + dz * dz); // np.linalg.norm(axis=1)
        if (arg < 0 || d < best) { best = d; arg = j; } // argmin: the first minimum
    }
    selected[v] = arg;
}

__global__ __launch_bounds__(64) void k_voxel_select_wave(const double *__restrict__ xyz, int64_t n, const int32_t *__restrict__ start,
                                                          int64_t nvox, const int32_t *__restrict__ perm,
                                                          const int64_t *__restrict__ order, int64_t *__restrict__ selected,
                                                          volatile int *__restrict__ flag)
{
    __shared__ double stage[3][64];
    const int64_t v = blockIdx.x;
    if (v >= nvox) return;
    const int32_t s = start[v], e = start[v + 1];
    if (e - s <= SF_VOXEL_WAVE) return; // k_voxel_select's
    const int lane = threadIdx.x;
    double acc = 0.0; // lanes 0, 1, 2: the running sums of x, y, z
    bool bad = false;
    for (int32_t t0 = s; t0 < e; t0 += 64) {
        const int32_t t = t0 + lane;
        double x = 0.0, y = 0.0, z = 0.0;
        if (t < e) {
            const int64_t j = order ? order[t] : (int64_t)perm[t];
            if (j < 0 || j >= n) bad = true;
            else { x = xyz[3 * j + 0]; y = xyz[3 * j + 1]; z = xyz[3 * j + 2]; }
        }
        stage[0][lane] = x; stage[1][lane] = y; stage[2][lane] = z;
        __builtin_amdgcn_wave_barrier();
        __threadfence_block();
        if (lane < 3) {
            const int cnt = e - t0 < 64 ? e - t0 : 64;
            for (int u = 0; u < cnt; ++u) acc += stage[lane][u]; // one column, row after row
        }
        __builtin_amdgcn_wave_barrier();
        __threadfence_block();
    }
    if (__ballot(bad)) {
        if (lane == 0) { *flag = SF_FLAG_VOXEL_ORDER; selected[v] = -1; }
        return;
    }
    const double k = (double)(e - s);
    const double mx = __shfl(acc, 0) / k, my = __shfl(acc, 1) / k, mz = __shfl(acc, 2) / k;
    double best = INFINITY;
    int32_t best_t = 0x7fffffff; // visiting position of this lane's first minimum
    for (int32_t t = s + lane; t < e; t += 64) {
        const int64_t j = order ? order[t] : (int64_t)perm[t];
        const double dx = xyz[3 * j + 0] - mx, dy = xyz[3 * j + 1] - my, dz = xyz[3 * j + 2] - mz;
        const double d = sqrt((dx * dx + dy * dy) + dz * dz);
        if (d < best) { best = d; best_t = t; } // (ascending t within a lane: the first of equal distances stays)
    }
    for (int off = 32; off > 0; off >>= 1) { // smallest distance, earliest visiting position among equals
        const double od = __shfl_xor(best, off);
        const int32_t ot = __shfl_xor(best_t, off);
        if (od < best || (od == best && ot < best_t)) { best = od; best_t = ot; }
    }
    // (a voxel whose distances are all NaN: np.argmin returns the first NaN, i.e. the first visited point)
    if (lane == 0) {
        const int32_t t = best_t == 0x7fffffff ? s : best_t;
        selected[v] = order ? order[t] : (int64_t)perm[t];
    }
}

int bits_for(double cells) // bits needed to hold values 0 .. cells
{
    int b = 1;
    while (b < 62 && (double)(1ull << b) <= cells) ++b;
    return b;
}

} // namespace

int sf_cloud_bbox_raw(sf_ctx *ctx, const double *xyz_dev, int64_t n, double lo[3], double hi[3]); // grid.hip

extern "C" void sf_voxels_free(sf_ctx *ctx, sf_voxels *v)
{
    if (!v) return;
    if (ctx) {
        sf_pool_release(ctx, v->perm);
        sf_pool_release(ctx, v->rank);
        sf_pool_release(ctx, v->start);
        if (v->owns_xyz) sf_pool_release(ctx, const_cast<double *>(v->xyz));
    }
    delete v;
}

extern "C" sf_voxels *sf_voxels_build(sf_ctx *ctx, const double *xyz, int64_t n, double voxel, int flags)
{
    if (!ctx || (!xyz && n > 0) || n < 0 || n > 2147483000LL) { sf_set_error("sf_voxels_build: bad argument"); return nullptr; }
    if (!(voxel > 0.0) || !std::isfinite(voxel)) { sf_set_error("sf_voxels_build: voxel size must be positive and finite"); return nullptr; }
    if (hipSetDevice(ctx->device) != hipSuccess) { sf_set_error("hipSetDevice failed"); return nullptr; }
    sf_voxels *v = new sf_voxels();
    v->n = n;
    v->voxel = voxel;
    auto fail = [&]() { sf_voxels_free(ctx, v); return (sf_voxels *)nullptr; };
    if (flags & SF_IN_DEVICE) {
        v->xyz = xyz;
    } else {
        double *d = nullptr;
        if (sf_palloc(ctx, &d, (size_t)n * 3) != SF_OK) return fail();
        v->xyz = d;
        v->owns_xyz = true;
        if (n && hipMemcpyAsync(d, xyz, (size_t)n * 24, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) {
            sf_set_error("sf_voxels_build: upload failed");
            return fail();
        }
    }
    if (sf_palloc(ctx, &v->start, (size_t)n + 2) != SF_OK) return fail();
    if (!n) return v;
    double hi[3];
    if (sf_cloud_bbox_raw(ctx, v->xyz, n, v->lo, hi) != SF_OK) return fail(); // np.min(points, axis=0), exact
    // field widths of the packed key from the extent (+1 cell of slack for the rounding of the division)
    const int bx = bits_for((hi[0] - v->lo[0]) / voxel + 1.0), by = bits_for((hi[1] - v->lo[1]) / voxel + 1.0),
              bz = bits_for((hi[2] - v->lo[2]) / voxel + 1.0);
    if (bx + by + bz > 63) {
        sf_set_error("sf_voxels_build: %d + %d + %d key bits: a voxel of %g is too fine for this extent", bx, by, bz, voxel);
        return fail();
    }
    sf_pool_guard tmp(ctx);
    unsigned long long *key = nullptr, *skey = nullptr;
    int32_t *val = nullptr, *head = nullptr, *scan = nullptr;
    int *bad = nullptr;
    if (tmp.alloc(&key, (size_t)n) != SF_OK || tmp.alloc(&skey, (size_t)n) != SF_OK || tmp.alloc(&val, (size_t)n) != SF_OK ||
        tmp.alloc(&head, (size_t)n) != SF_OK || tmp.alloc(&scan, (size_t)n) != SF_OK || tmp.alloc(&bad, 2) != SF_OK ||
        sf_palloc(ctx, &v->perm, (size_t)n) != SF_OK || sf_palloc(ctx, &v->rank, (size_t)n) != SF_OK)
        return fail();
    const dim3 grid((unsigned)sf_div_up(n, 256)), block(256);
    bool ok = hipMemsetAsync(bad, 0, 2 * sizeof(int), ctx->stream) == hipSuccess; // [0]: non-finite input, [1]: fullest voxel
    {
        sf_launch_timer t_(ctx, "v1_voxel_keys");
        hipLaunchKernelGGL(k_voxel_keys, grid, block, 0, ctx->stream, v->xyz, n, v->lo[0], v->lo[1], v->lo[2], voxel, by, bz, key,
                           val, bad);
    }
    {
        sf_launch_timer t_(ctx, "v2_voxel_sort");
        size_t tb = 0;
        ok = ok && rocprim::radix_sort_pairs<sf_vox_sort_config>(nullptr, tb, key, skey, val, v->perm, (size_t)n, 0, bx + by + bz, ctx->stream) == hipSuccess;
        void *ts = nullptr;
        ok = ok && sf_ctx_scratch(ctx, tb, &ts) == SF_OK;
        ok = ok && rocprim::radix_sort_pairs<sf_vox_sort_config>(ts, tb, key, skey, val, v->perm, (size_t)n, 0, bx + by + bz, ctx->stream) == hipSuccess;
    }
    {
        sf_launch_timer t_(ctx, "v3_voxel_runs");
        hipLaunchKernelGGL(k_voxel_heads, grid, block, 0, ctx->stream, skey, n, head);
        size_t tb = 0;
        ok = ok && rocprim::inclusive_scan(nullptr, tb, head, scan, (size_t)n, rocprim::plus<int32_t>(), ctx->stream) == hipSuccess;
        void *ts = nullptr;
        ok = ok && sf_ctx_scratch(ctx, tb, &ts) == SF_OK;
        ok = ok && rocprim::inclusive_scan(ts, tb, head, scan, (size_t)n, rocprim::plus<int32_t>(), ctx->stream) == hipSuccess;
        hipLaunchKernelGGL(k_voxel_starts, grid, block, 0, ctx->stream, head, scan, n, v->rank, v->start);
        hipLaunchKernelGGL(k_voxel_maxpop, grid, block, 0, ctx->stream, head, v->rank, v->start, n, bad + 1);
    }
    int32_t last = 0;
    int hbad2[2] = {0, 0};
    ok = ok && hipMemcpyAsync(&last, scan + (n - 1), sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream) == hipSuccess;
    ok = ok && hipMemcpyAsync(hbad2, bad, 2 * sizeof(int), hipMemcpyDeviceToHost, ctx->stream) == hipSuccess;
    ok = ok && hipStreamSynchronize(ctx->stream) == hipSuccess && hipGetLastError() == hipSuccess;
    if (!ok) { sf_set_error("sf_voxels_build: device error"); return fail(); }
    if (hbad2[0]) { sf_set_error("sf_voxels_build: non-finite coordinates"); return fail(); }
    v->nvox = last;
    v->max_pop = hbad2[1];
    return v;
}

extern "C" int64_t sf_voxels_count(const sf_voxels *v) { return v ? v->nvox : -1; }

extern "C" int sf_voxels_inverse(sf_ctx *ctx, sf_voxels *v, int64_t *inverse)
{
    if (!ctx || !v || !inverse) { sf_set_error("sf_voxels_inverse: null argument"); return SF_ERR_ARG; }
    SF_HIP(hipSetDevice(ctx->device));
    if (!v->n) return SF_OK;
    sf_pool_guard tmp(ctx);
    int64_t *d = nullptr;
    SF_CHECK(tmp.alloc(&d, (size_t)v->n));
    SF_LAUNCH(ctx, "v3_voxel_inverse", k_voxel_inverse, dim3((unsigned)sf_div_up(v->n, 256)), dim3(256), v->perm, v->rank, v->n, d);
    SF_HIP(hipMemcpyAsync(inverse, d, (size_t)v->n * sizeof(int64_t), hipMemcpyDeviceToHost, ctx->stream));
    SF_HIP(hipStreamSynchronize(ctx->stream));
    return SF_OK;
}

extern "C" int sf_voxels_select(sf_ctx *ctx, sf_voxels *v, const int64_t *order, int64_t *selected, int64_t *counts)
{
    if (!ctx || !v || !selected) { sf_set_error("sf_voxels_select: null argument"); return SF_ERR_ARG; }
    SF_HIP(hipSetDevice(ctx->device));
    if (!v->nvox) return SF_OK;
    sf_pool_guard tmp(ctx);
    int64_t *dorder = nullptr, *dsel = nullptr, *dcnt = nullptr;
    if (order) {
        SF_CHECK(tmp.alloc(&dorder, (size_t)v->n));
        SF_HIP(hipMemcpyAsync(dorder, order, (size_t)v->n * sizeof(int64_t), hipMemcpyHostToDevice, ctx->stream));
    }
    SF_CHECK(tmp.alloc(&dsel, (size_t)v->nvox));
    if (counts) SF_CHECK(tmp.alloc(&dcnt, (size_t)v->nvox));
    SF_LAUNCH(ctx, "v4_voxel_select", k_voxel_select, dim3((unsigned)sf_div_up(v->nvox, 128)), dim3(128), v->xyz, v->n, v->start,
              v->nvox, v->perm, (const int64_t *)dorder, dsel, dcnt, ctx->dev_flag);
    // (a wave per voxel for the voxels above SF_VOXEL_WAVE points -- launched only when the fullest voxel is one: with
    // typical voxel sizes none is, and the launch was millions of workgroups that returned at once)
    if (v->max_pop > (int32_t)SF_VOXEL_WAVE)
        SF_LAUNCH(ctx, "v4_voxel_select", k_voxel_select_wave, dim3((unsigned)v->nvox), dim3(64), v->xyz, v->n, v->start, v->nvox,
                  v->perm, (const int64_t *)dorder, dsel, ctx->dev_flag);
    SF_HIP(hipMemcpyAsync(selected, dsel, (size_t)v->nvox * sizeof(int64_t), hipMemcpyDeviceToHost, ctx->stream));
    if (counts) SF_HIP(hipMemcpyAsync(counts, dcnt, (size_t)v->nvox * sizeof(int64_t), hipMemcpyDeviceToHost, ctx->stream));
    SF_HIP(hipStreamSynchronize(ctx->stream));
    return sf_ctx_check_flag(ctx);
}
