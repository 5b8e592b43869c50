// icp.hip -- one ICP iteration on the device: transform, nearest neighbour, inlier filter and the normal-equation sums.
//
// Replaces, per iteration of shot_fpfh/icp.py (:64-72, :108-124, :160-183):
//   transformation[points]                          core/rigid_transform.py:81-88
//   kdtree.query(points_aligned)                    the grid k-NN kernel with k = 1 (search.hip)
//   the inlier filter  distances <= d_max
//   solver_point_to_point's centroids and 3x3 cross-covariance      core/solvers.py:17-18
//   solver_point_to_plane's G^T G (6x6) and G^T h (6)                core/solvers.py:38-46
//   the residuals the reference reports as "rms"                     icp.py:66-69, 176-182
// The reference forms these with NumPy over (n_inliers, 3) arrays on the host, after shipping distances and indices
// back from the tree.  Here the scan subset stays in HBM; per iteration ~30 doubles come back, and the host does what is
// left: a 3x3 SVD or a 6x6 solve, and the composition of the transform.
//
// Sums are accumulated per thread, folded per block (shuffles + LDS) and then over the block partials by ONE block in a
// fixed order, so a run is reproducible bit for bit.  Point-to-point is centred in a second pass (centroids first),
// which keeps the cross-covariance free of cancellation, like the reference's explicit centring.
#include "common.h"
#include "device_util.h"

extern "C" sf_nbrs *sf_knn_search(sf_ctx *ctx, sf_cloud *c, const double *queries, int64_t m, int k, int flags);
extern "C" void sf_nbrs_free(sf_ctx *ctx, sf_nbrs *nb);
int sf_cloud_ensure_sorted_normals(sf_ctx *ctx, sf_cloud *c);

namespace {

constexpr int ICP_BLOCKS = 256;
constexpr int ICP_NV = 32; // values per partial row (largest mode: 21 + 6 + 2 = 29)

// p <- p R^T + t, rows (optionally selected by `sel`) written to out (may alias pts when sel == null)
__global__ void k_transform(const double *__restrict__ pts, const int64_t *__restrict__ sel, int64_t m,
                            const double *__restrict__ Rt, double *__restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    const int64_t j = sel ? sel[i] : i;
    const double x = pts[3 * j], y = pts[3 * j + 1], z = pts[3 * j + 2];
    double ox = x, oy = y, oz = z;
    if (Rt) {
        ox = ((x * Rt[0] + y * Rt[1]) + z * Rt[2]) + Rt[9];
        oy = ((x * Rt[3] + y * Rt[4]) + z * Rt[5]) + Rt[10];
        oz = ((x * Rt[6] + y * Rt[7]) + z * Rt[8]) + Rt[11];
    }
    out[3 * i] = ox; out[3 * i + 1] = oy; out[3 * i + 2] = oz;
}

template <int NV>
__device__ inline void block_fold(double (&acc)[NV], double *__restrict__ partial_row)
{
    __shared__ double sh[4][ICP_NV];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        double a = acc[v];
        for (int off = 32; off > 0; off >>= 1) a += __shfl_xor(a, off);
        if (lane == 0) sh[wave][v] = a;
    }
    __syncthreads();
    if (threadIdx.x < NV) partial_row[threadIdx.x] = ((sh[0][threadIdx.x] + sh[1][threadIdx.x]) + sh[2][threadIdx.x]) + sh[3][threadIdx.x];
}

// pass A (both modes): inlier count, sum of inlier points, sum of their neighbours
// pass B, MODE 0 (point to point): centred cross-covariance H = sum (p - pbar)(q - qbar)^T (9) and sum |p - q|^2
// pass B, MODE 1 (point to plane): upper triangle of G^T G (21), G^T h (6), sum |h| with g = [p x n, n], h = (q - p) . n
template <int PASS, int MODE>
__global__ __launch_bounds__(256) void k_icp_sums(const double *__restrict__ qx, const double *__restrict__ qy,
                                                  const double *__restrict__ qz, const int32_t *__restrict__ idx,
                                                  const double *__restrict__ rec, int64_t m, double d_max,
                                                  const double *__restrict__ mean /* 6, pass B mode 0 */,
                                                  double *__restrict__ partial)
{
    constexpr int NV = PASS == 0 ? 7 : (MODE == 0 ? 10 : 28);
    double acc[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) acc[v] = 0.0;
    double pm[3] = {0, 0, 0}, qm[3] = {0, 0, 0};
    if (PASS == 1 && MODE == 0) {
        pm[0] = mean[0]; pm[1] = mean[1]; pm[2] = mean[2];
        qm[0] = mean[3]; qm[1] = mean[4]; qm[2] = mean[5];
    }
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < m; i += (int64_t)gridDim.x * blockDim.x) {
        const double px = qx[i], py = qy[i], pz = qz[i];
        double x, y, z, nx = 0, ny = 0, nz = 0;
        if (PASS == 1 && MODE == 1) sf_load_pn(rec, idx[i], x, y, z, nx, ny, nz);
        else sf_load_xyz(rec, idx[i], x, y, z);
        const double dx = x - px, dy = y - py, dz = z - pz;
        const double d2 = (dx * dx + dy * dy) + dz * dz;
        if (!(sqrt(d2) <= d_max)) continue; // KDTree.query returns sqrt(d2); `distances <= d_max` (icp.py:64, 112, 164)
        if (PASS == 0) {
            acc[0] += 1.0;
            acc[1] += px; acc[2] += py; acc[3] += pz;
            acc[4] += x; acc[5] += y; acc[6] += z;
        } else if (MODE == 0) {
            const double ax = px - pm[0], ay = py - pm[1], az = pz - pm[2];
            const double bx = x - qm[0], by = y - qm[1], bz = z - qm[2];
            acc[0] += ax * bx; acc[1] += ax * by; acc[2] += ax * bz;
            acc[3] += ay * bx; acc[4] += ay * by; acc[5] += ay * bz;
            acc[6] += az * bx; acc[7] += az * by; acc[8] += az * bz;
            acc[9] += d2;
        } else {
            const double g[6] = {py * nz - pz * ny, pz * nx - px * nz, px * ny - py * nx, nx, ny, nz}; // cross(p, n), n
            const double h = (dx * nx + dy * ny) + dz * nz;
            int t = 0;
#pragma unroll
            for (int a = 0; a < 6; ++a)
#pragma unroll
                for (int b = a; b < 6; ++b) acc[t++] += g[a] * g[b];
#pragma unroll
            for (int a = 0; a < 6; ++a) acc[21 + a] += g[a] * h;
            acc[27] += fabs(h);
        }
    }
    block_fold<NV>(acc, partial + (size_t)blockIdx.x * ICP_NV);
}

// fold the block partials in block order; pass A also leaves the centroids (means) for pass B
__global__ __launch_bounds__(64) void k_icp_final(const double *__restrict__ partial, int nblocks, int nv, double *__restrict__ out,
                                                  double *__restrict__ mean)
{
    const int v = threadIdx.x;
    double a = 0.0;
    if (v < nv)
        for (int b = 0; b < nblocks; ++b) a += partial[(size_t)b * ICP_NV + v];
    if (v < nv) out[v] = a;
    if (mean) {
        const double cnt = __shfl(a, 0);
        if (v >= 1 && v <= 6) mean[v - 1] = cnt > 0.0 ? a / cnt : 0.0;
    }
}

} // namespace

extern "C" int sf_transform_points(sf_ctx *ctx, double *pts_dev, int64_t n, const double *Rt)
{
    if (!ctx || !pts_dev || !Rt || n < 0) { sf_set_error("sf_transform_points: bad argument"); return SF_ERR_ARG; }
    SF_HIP(hipSetDevice(ctx->device));
    if (!n) return SF_OK;
    sf_pool_guard tmp(ctx);
    double *dRt = nullptr;
    SF_CHECK(tmp.alloc(&dRt, 12));
    SF_HIP(hipMemcpyAsync(dRt, Rt, 12 * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    SF_LAUNCH(ctx, "i0_transform", k_transform, dim3((unsigned)sf_div_up(n, 256)), dim3(256), (const double *)pts_dev,
              (const int64_t *)nullptr, n, (const double *)dRt, pts_dev);
    SF_HIP(hipStreamSynchronize(ctx->stream)); // Rt is a host buffer
    return SF_OK;
}

extern "C" int sf_icp_accumulate(sf_ctx *ctx, sf_cloud *ref, const double *pts_dev, const int64_t *sel_dev, int64_t m,
                                 const double *Rt, double d_max, int mode, double *sums)
{
    if (!ctx || !ref || !pts_dev || !sums || m < 0 || (mode != 0 && mode != 1)) { sf_set_error("sf_icp_accumulate: bad argument"); return SF_ERR_ARG; }
    SF_HIP(hipSetDevice(ctx->device));
    for (int i = 0; i < 40; ++i) sums[i] = 0.0;
    if (!m) return SF_OK;
    if (ref->n < 1) { sf_set_error("sf_icp_accumulate: empty reference cloud"); return SF_ERR_ARG; }
    sf_pool_guard tmp(ctx);
    double *dRt = nullptr, *moved = nullptr, *partial = nullptr, *dout = nullptr, *dmean = nullptr;
    if (Rt) {
        SF_CHECK(tmp.alloc(&dRt, 12));
        SF_HIP(hipMemcpyAsync(dRt, Rt, 12 * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    }
    SF_CHECK(tmp.alloc(&moved, (size_t)m * 3));
    SF_CHECK(tmp.alloc(&partial, (size_t)ICP_BLOCKS * ICP_NV));
    SF_CHECK(tmp.alloc(&dout, 80));
    SF_CHECK(tmp.alloc(&dmean, 8));
    SF_LAUNCH(ctx, "i0_transform", k_transform, dim3((unsigned)sf_div_up(m, 256)), dim3(256), pts_dev, sel_dev, m,
              (const double *)dRt, moved);
    sf_nbrs *nb = sf_knn_search(ctx, ref, moved, m, 1, SF_IN_DEVICE); // kdtree.query(points_aligned)
    if (!nb) return SF_ERR_HIP;
    struct nb_guard { sf_ctx *c; sf_nbrs *n; ~nb_guard() { sf_nbrs_free(c, n); } } nbg{ctx, nb};
    if (mode == 1) SF_CHECK(sf_cloud_ensure_sorted_normals(ctx, ref));
    // the k-NN lists are in PROCESSING order (queries sorted by cell): sums do not care, and qx / qy / qz follow it
    const dim3 grid(ICP_BLOCKS), block(256);
    SF_LAUNCH(ctx, "i1_icp_sums", (k_icp_sums<0, 0>), grid, block, nb->qx, nb->qy, nb->qz, nb->idx, ref->rec, m, d_max,
              (const double *)nullptr, partial);
    SF_LAUNCH(ctx, "i1_icp_final", k_icp_final, dim3(1), dim3(64), (const double *)partial, ICP_BLOCKS, 7, dout, dmean);
    if (mode == 0) {
        SF_LAUNCH(ctx, "i1_icp_sums", (k_icp_sums<1, 0>), grid, block, nb->qx, nb->qy, nb->qz, nb->idx, ref->rec, m, d_max,
                  (const double *)dmean, partial);
        SF_LAUNCH(ctx, "i1_icp_final", k_icp_final, dim3(1), dim3(64), (const double *)partial, ICP_BLOCKS, 10, dout + 8, (double *)nullptr);
    } else {
        SF_LAUNCH(ctx, "i1_icp_sums", (k_icp_sums<1, 1>), grid, block, nb->qx, nb->qy, nb->qz, nb->idx, ref->rec, m, d_max,
                  (const double *)nullptr, partial);
        SF_LAUNCH(ctx, "i1_icp_final", k_icp_final, dim3(1), dim3(64), (const double *)partial, ICP_BLOCKS, 28, dout + 8, (double *)nullptr);
    }
    SF_HIP(hipMemcpyAsync(sums, dout, 40 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    SF_HIP(hipStreamSynchronize(ctx->stream));
    return SF_OK;
}
