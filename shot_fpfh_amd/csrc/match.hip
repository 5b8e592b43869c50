// match.hip -- K8 (brute-force L2 arg-min matching) and K9 (RANSAC inlier scoring).
//
// K8 replaces scipy.spatial.distance.cdist + argmin at matching.py:47-52, 63-65, 164-168.
//   dist(i, j) = sqrt(sum_t (a[i,t] - b[j,t])^2), the sum taken left to right in float64 without FMA
//   (scipy's euclidean loop), argmin = FIRST minimum.  The M1 x M2 matrix is never materialised: a
//   256-thread workgroup owns a 64 x 64 tile of it (4 x 4 per thread), streams the descriptor
//   dimension through LDS in slices of 16, reduces the tile to per-row minima with wave shuffles and
//   walks the reference rows in ascending order so "first minimum" is preserved.  Small M1 is split
//   over several workgroups along M2 and merged by k_match_merge.
//   Roofline: FP64 vector ALU (3 flop per pair-dimension), bytes are negligible.
// K9 replaces the inlier count of ransac.py:60-67 (RigidTransform.__getitem__, rigid_transform.py:81-88).
#include <cmath>
#include <cstdlib>

#include "common.h"
#include "device_util.h"

namespace {

constexpr int TM = 64, TN = 64, TK = 16;

// n_scales == 1, masks null: plain cdist + argmin.  Otherwise the "minimum over scales" distance of
// matching.py:77-136: dist(i,j) = min over scales s of (a_ok[s][i] && b_ok[s][j] ? ||a_s[i] - b_s[j]|| : max_val),
// a, b being (n_scales, m, d) stacks; a row that is empty at every scale keeps max_val everywhere.
__global__ __launch_bounds__(256) void k_match_tile(const double *__restrict__ a, int64_t m1,
                                                    const double *__restrict__ b, int64_t m2, int64_t d,
                                                    int n_scales, const unsigned char *__restrict__ a_ok,
                                                    const unsigned char *__restrict__ b_ok, double max_val,
                                                    int64_t tiles_per_split, double *__restrict__ pdist,
                                                    int64_t *__restrict__ pidx)
{
    __shared__ double As[TK][TM + 1];
    __shared__ double Bs[TK][TN + 1];
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    const int64_t i0 = (int64_t)blockIdx.x * TM;
    const int split = blockIdx.y;
    const int64_t ntiles = (m2 + TN - 1) / TN;
    const int64_t jt0 = (int64_t)split * tiles_per_split;
    const int64_t jt1 = jt0 + tiles_per_split < ntiles ? jt0 + tiles_per_split : ntiles;
    double best[4];
    int64_t bidx[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) { best[u] = INFINITY; bidx[u] = 0; }

    for (int64_t jt = jt0; jt < jt1; ++jt) {
        const int64_t j0 = jt * TN;
        double dmin[4][4];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int v = 0; v < 4; ++v) dmin[u][v] = a_ok ? max_val : INFINITY;
        for (int sc = 0; sc < n_scales; ++sc) {
            const double *as = a + (int64_t)sc * m1 * d, *bs = b + (int64_t)sc * m2 * d;
            double acc[4][4];
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int v = 0; v < 4; ++v) acc[u][v] = 0.0;
            for (int64_t t0 = 0; t0 < d; t0 += TK) {
                // stage TM x TK of a and TN x TK of b (zero padded); 1024 elements each, 4 per thread
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int lin = tid + 256 * e; // 0..1023
                    const int r = lin >> 4, cc = lin & 15;
                    const int64_t t = t0 + cc;
                    As[cc][r] = (i0 + r < m1 && t < d) ? as[(i0 + r) * d + t] : 0.0;
                    Bs[cc][r] = (j0 + r < m2 && t < d) ? bs[(j0 + r) * d + t] : 0.0;
                }
                __syncthreads();
#pragma unroll
                for (int t = 0; t < TK; ++t) {
                    double av[4], bv[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) { av[u] = As[t][ty * 4 + u]; bv[u] = Bs[t][tx * 4 + u]; }
#pragma unroll
                    for (int u = 0; u < 4; ++u)
#pragma unroll
                        for (int v = 0; v < 4; ++v) {
                            const double df = av[u] - bv[v];
                            acc[u][v] += df * df;
                        }
                }
                __syncthreads();
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    double dist = sqrt(acc[u][v]);
                    if (a_ok) {
                        const int64_t i = i0 + ty * 4 + u, j = j0 + tx * 4 + v;
                        const bool ok = i < m1 && j < m2 && a_ok[(int64_t)sc * m1 + i] && b_ok[(int64_t)sc * m2 + j];
                        dist = ok ? dist : max_val;
                    }
                    dmin[u][v] = fmin(dmin[u][v], dist);
                }
        }
        // per-row minimum over this tile's 64 columns: 4 local columns, then the 16 tx lanes of the row group
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            double bd = INFINITY;
            int64_t bj = 0x7fffffffffffffffLL;
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int64_t j = j0 + tx * 4 + v;
                const double dist = j < m2 ? dmin[u][v] : INFINITY;
                if (dist < bd) { bd = dist; bj = j; } // ascending j: first minimum kept
            }
#pragma unroll
            for (int off = 1; off < 16; off <<= 1) {
                const double od = __shfl_xor(bd, off, 16);
                const int64_t oj = __shfl_xor(bj, off, 16);
                if (od < bd || (od == bd && oj < bj)) { bd = od; bj = oj; }
            }
            if (bd < best[u]) { best[u] = bd; bidx[u] = bj; }
        }
    }
    if (tx == 0) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t i = i0 + ty * 4 + u;
            if (i < m1) {
                pdist[(int64_t)split * m1 + i] = best[u];
                pidx[(int64_t)split * m1 + i] = bidx[u];
            }
        }
    }
}

// mask[i] = 1 when row i of a (m x d) has a non-zero entry: np.any(desc, axis=1) of matching.py:43-44/162-163
__global__ __launch_bounds__(256) void k_rows_nonzero(const double *__restrict__ a, int64_t m, int64_t d,
                                                      unsigned char *__restrict__ mask)
{
    const int lane = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= m) return;
    bool nz = false;
    for (int64_t t = lane; t < d; t += 64) nz |= a[i * d + t] != 0.0;
    const unsigned long long any = __ballot(nz);
    if (lane == 0) mask[i] = any != 0ull;
}

// dst row i = src row sel[i], or a zero row when sel[i] < 0 (the padding of an equal-sized per-rank block): picks the
// descriptors of a keypoint subset out of a resident descriptor matrix before the exchange that precedes K8
// (a selection >= n_rows is the caller's mistake: the row comes out zero and the context's flag is raised, see common.h)
__global__ __launch_bounds__(256) void k_rows_gather(const double *__restrict__ src, int64_t n_rows,
                                                     const int64_t *__restrict__ sel, int64_t m, int64_t d,
                                                     double *__restrict__ dst, volatile int *__restrict__ flag)
{
    const int lane = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= m) return;
    int64_t r = sel[i];
    if (r >= n_rows) {
        if (lane == 0) *flag = SF_FLAG_ROWS_GATHER;
        r = -1;
    }
    for (int64_t t = lane; t < d; t += 64) dst[i * d + t] = r < 0 ? 0.0 : src[r * d + t];
}

__global__ void k_match_merge(const double *__restrict__ pdist, const int64_t *__restrict__ pidx, int64_t m1, int nsplit,
                              int64_t *__restrict__ idx, double *__restrict__ dist)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m1) return;
    double bd = pdist[i];
    int64_t bj = pidx[i];
    for (int s = 1; s < nsplit; ++s) {
        const double od = pdist[(int64_t)s * m1 + i];
        if (od < bd) { bd = od; bj = pidx[(int64_t)s * m1 + i]; }
    }
    idx[i] = bj;
    if (dist) dist[i] = bd;
}

// "Minimum over scales" (matching.py:77-136) from the per-scale row minima: min_j min_s d_s(i, j) = min_s min_j d_s(i, j), and the
// first column that attains it is the lowest of the per-scale arg-mins at that distance.  The per-scale results come from the
// matrix-core matcher with empty descriptors masked at +inf; the reference counts a pair with an empty side as `max_val`:
//   * best == +inf: every pair of the row has an empty side at every scale -> a row of max_val, arg-min 0;
//   * best <  max_val: the reference's answer (pairs at max_val lose against it);
//   * max_val <= best < +inf (descriptors so far apart that "empty" pairs would win): counted in *n_hard -- the caller then
//     runs the exact tile kernel on the whole problem (never seen on normalised rows, whose distances are at most 2).
__global__ void k_match_scale_fold(const int64_t *__restrict__ idx_s, const double *__restrict__ dist_s, int n_scales, int64_t m1,
                                   double max_val, int64_t *__restrict__ idx, double *__restrict__ dist, unsigned *__restrict__ n_hard)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m1) return;
    double bd = INFINITY;
    int64_t bj = 0;
    for (int sc = 0; sc < n_scales; ++sc) {
        const double d = dist_s[(int64_t)sc * m1 + i];
        const int64_t j = idx_s[(int64_t)sc * m1 + i];
        if (d < bd || (d == bd && d < INFINITY && j < bj)) { bd = d; bj = j; }
    }
    if (!(bd < INFINITY)) { bj = 0; bd = max_val; }
    else if (!(bd < max_val)) atomicAdd(n_hard, 1u);
    idx[i] = bj;
    if (dist) dist[i] = bd;
}

// K9: inlier counts of all candidate transforms (ransac.py:60-67).  A workgroup keeps a tile of 256 x 8 matched
// pairs in registers and walks over its range of draws; the 12 coefficients of a draw are wave-uniform (scalar
// loads), the votes of a wave are a ballot + popcount, one LDS add per wave and draw, one global add per workgroup
// and draw at the end.  (A workgroup per draw re-reads all pairs from L2 for every draw: 48 B per pair-draw.)
// The residual is formed exactly like the reference's (a @ R.T + t) - b, left to right, without FMA.  The
// reference tests sqrt(s) <= thr; s <= lo and s >= hi (the squares of thr and of the next double, rounded
// outwards) decide that without the square root for all but a 2^-51-wide band, where the square root is taken.
constexpr int K9_MPT = 8;         // pairs per thread
constexpr int K9_TILE = 256 * K9_MPT;
constexpr int K9_MAX_DRAWS = 8192; // draws per workgroup (LDS counters)

__global__ __launch_bounds__(256) void k_ransac_score(const double *__restrict__ a, const double *__restrict__ b,
                                                      int64_t m, const double *__restrict__ Rt, int64_t n_draws,
                                                      int64_t draws_per_split, double thr, double lo, double hi,
                                                      unsigned long long *__restrict__ inliers)
{
    __shared__ int cnt[K9_MAX_DRAWS];
    const int tid = threadIdx.x, lane = tid & 63;
    const int64_t d0 = (int64_t)blockIdx.y * draws_per_split;
    const int nd = (int)((n_draws - d0 < draws_per_split) ? n_draws - d0 : draws_per_split);
    for (int d = tid; d < nd; d += 256) cnt[d] = 0;
    double px[K9_MPT], py[K9_MPT], pz[K9_MPT], qx[K9_MPT], qy[K9_MPT], qz[K9_MPT];
    unsigned valid = 0;
#pragma unroll
    for (int u = 0; u < K9_MPT; ++u) {
        const int64_t i = (int64_t)blockIdx.x * K9_TILE + u * 256 + tid;
        const bool ok = i < m;
        const int64_t ii = ok ? i : 0;
        px[u] = a[3 * ii]; py[u] = a[3 * ii + 1]; pz[u] = a[3 * ii + 2];
        qx[u] = b[3 * ii]; qy[u] = b[3 * ii + 1]; qz[u] = b[3 * ii + 2];
        valid |= (ok ? 1u : 0u) << u;
    }
    __syncthreads();
    for (int d = 0; d < nd; ++d) {
        const double *R = Rt + 12 * (d0 + d); // uniform address: scalar loads
        const double r0 = R[0], r1 = R[1], r2 = R[2], r3 = R[3], r4 = R[4], r5 = R[5], r6 = R[6], r7 = R[7], r8 = R[8];
        const double t0 = R[9], t1 = R[10], t2 = R[11];
        int votes = 0;
#pragma unroll
        for (int u = 0; u < K9_MPT; ++u) {
            const double e0 = ((px[u] * r0 + py[u] * r1) + pz[u] * r2) + t0 - qx[u];
            const double e1 = ((px[u] * r3 + py[u] * r4) + pz[u] * r5) + t1 - qy[u];
            const double e2 = ((px[u] * r6 + py[u] * r7) + pz[u] * r8) + t2 - qz[u];
            const double s = (e0 * e0 + e1 * e1) + e2 * e2;
            bool in = s <= lo;
            if (__ballot(!in & (s < hi))) { // the band between the two squares (NaN: neither); practically never
                double sb = s;
                asm volatile("" : "+v"(sb)); // keeps the square root inside the branch
                in |= sqrt(sb) <= thr;
            }
            votes += __popcll(__ballot(in & ((valid >> u) & 1u)));
        }
        if (lane == 0 && votes) atomicAdd(&cnt[d], votes);
    }
    __syncthreads();
    for (int d = tid; d < nd; d += 256)
        if (cnt[d]) atomicAdd(&inliers[d0 + d], (unsigned long long)cnt[d]);
}

} // namespace

static int match_one_way(sf_ctx *ctx, const double *da, int64_t m1, const double *db, int64_t m2, int64_t d,
                         int64_t *didx, double *ddist, const char *name, int n_scales = 1,
                         const unsigned char *a_ok = nullptr, const unsigned char *b_ok = nullptr, double max_val = 0.0)
{
    const int64_t row_tiles = sf_div_up(m1, TM), col_tiles = sf_div_up(m2, TN);
    int64_t nsplit = 1;
    if (row_tiles < 2048) nsplit = std::min<int64_t>(col_tiles, sf_div_up(2048, row_tiles));
    if (nsplit > 65535) nsplit = 65535;
    const int64_t tiles_per_split = sf_div_up(col_tiles, nsplit);
    nsplit = sf_div_up(col_tiles, tiles_per_split);
    sf_pool_guard tmp(ctx); // stream-ordered release: safe to reuse by later launches on this stream
    double *pdist = nullptr;
    int64_t *pidx = nullptr;
    SF_CHECK(tmp.alloc(&pdist, (size_t)(nsplit * m1)));
    SF_CHECK(tmp.alloc(&pidx, (size_t)(nsplit * m1)));
    SF_LAUNCH(ctx, name, k_match_tile, dim3((unsigned)row_tiles, (unsigned)nsplit), dim3(256), da, m1, db, m2, d,
              n_scales, a_ok, b_ok, max_val, tiles_per_split, pdist, pidx);
    SF_LAUNCH(ctx, "k8_match_merge", k_match_merge, dim3((unsigned)sf_div_up(m1, 256)), dim3(256), pdist, pidx, m1,
              (int)nsplit, didx, ddist);
    return SF_OK;
}

// exact kernel, callable from the GEMM fast path (match_gemm.hip) for its undecided rows
int sf_match_exact(sf_ctx *ctx, const double *da, int64_t m1, const double *db, int64_t m2, int64_t d, int64_t *didx,
                   double *ddist, const char *name, const unsigned char *a_ok, const unsigned char *b_ok)
{
    if (!a_ok && !b_ok) return match_one_way(ctx, da, m1, db, m2, d, didx, ddist, name);
    // masked form: a row whose mask is 0 is at distance +inf from everything (the scan side only if given)
    sf_pool_guard tmp(ctx);
    unsigned char *ones = nullptr;
    if (!a_ok) {
        SF_CHECK(tmp.alloc(&ones, (size_t)m1));
        SF_HIP(hipMemsetAsync(ones, 1, (size_t)(m1 ? m1 : 1), ctx->stream));
    }
    return match_one_way(ctx, da, m1, db, m2, d, didx, ddist, name, 1, a_ok ? a_ok : ones, b_ok, INFINITY);
}

int sf_match_gemm(sf_ctx *ctx, const double *da, int64_t m1, const double *db, int64_t m2, int64_t d, int64_t *didx,
                  double *ddist, const char *name, int64_t *n_slow, const unsigned char *a_ok,
                  const unsigned char *b_ok); // match_gemm.hip

// Large problems go through the FP64 matrix cores (same result, see match_gemm.hip); small ones, where the
// fixed costs of the fast path dominate, straight through the exact kernel.  SF_MATCH_EXACT=1 forces the latter.
static int match_dispatch(sf_ctx *ctx, const double *da, int64_t m1, const double *db, int64_t m2, int64_t d,
                          int64_t *didx, double *ddist, const char *name_exact, const char *name_gemm)
{
    static const bool force_exact = getenv("SF_MATCH_EXACT") && getenv("SF_MATCH_EXACT")[0] == '1';
    const double work = (double)m1 * (double)m2 * (double)d;
    if (force_exact || work < 5e8 || m2 < 256) return match_one_way(ctx, da, m1, db, m2, d, didx, ddist, name_exact);
    return sf_match_gemm(ctx, da, m1, db, m2, d, didx, ddist, name_gemm, nullptr, nullptr, nullptr);
}

extern "C" int sf_match_argmin(sf_ctx *ctx, const double *a, int64_t m1, const double *b, int64_t m2, int64_t d,
                               int64_t *idx, double *dist, int64_t *col_idx, int flags)
{
    if (!ctx || !a || !b || !idx || m1 < 0 || m2 < 0 || d <= 0) { sf_set_error("sf_match_argmin: bad argument"); return SF_ERR_ARG; }
    if (m2 == 0 && m1 > 0) { sf_set_error("sf_match_argmin: empty reference set (argmin of an empty sequence)"); return SF_ERR_ARG; }
    SF_HIP(hipSetDevice(ctx->device));
    const bool in_dev = flags & SF_IN_DEVICE, out_dev = flags & SF_OUT_DEVICE;
    sf_pool_guard tmp(ctx); // staging buffers: back in the pool on every return path
    double *da = const_cast<double *>(a), *db = const_cast<double *>(b);
    if (!in_dev) {
        SF_CHECK(tmp.alloc(&da, (size_t)std::max<int64_t>(m1 * d, 1)));
        SF_CHECK(tmp.alloc(&db, (size_t)std::max<int64_t>(m2 * d, 1)));
        if (m1) SF_HIP(hipMemcpyAsync(da, a, (size_t)(m1 * d) * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
        if (m2) SF_HIP(hipMemcpyAsync(db, b, (size_t)(m2 * d) * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    }
    int64_t *didx = idx, *dcol = col_idx;
    double *ddist = dist;
    if (!out_dev) {
        SF_CHECK(tmp.alloc(&didx, (size_t)std::max<int64_t>(m1, 1)));
        if (dist) SF_CHECK(tmp.alloc(&ddist, (size_t)std::max<int64_t>(m1, 1)));
        if (col_idx) SF_CHECK(tmp.alloc(&dcol, (size_t)std::max<int64_t>(m2, 1)));
    }
    if (m1) SF_CHECK(match_dispatch(ctx, da, m1, db, m2, d, didx, ddist, "k8_match_tile", "k8_match_gemm"));
    if (col_idx && m2 && m1)
        SF_CHECK(match_dispatch(ctx, db, m2, da, m1, d, dcol, nullptr, "k8_match_tile_cols", "k8_match_gemm_cols"));
    if (!out_dev) {
        if (m1) SF_HIP(hipMemcpyAsync(idx, didx, (size_t)m1 * sizeof(int64_t), hipMemcpyDeviceToHost, ctx->stream));
        if (dist && m1) SF_HIP(hipMemcpyAsync(dist, ddist, (size_t)m1 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        if (col_idx && m2) SF_HIP(hipMemcpyAsync(col_idx, dcol, (size_t)m2 * sizeof(int64_t), hipMemcpyDeviceToHost, ctx->stream));
    }
    if (!out_dev || !in_dev) SF_HIP(hipStreamSynchronize(ctx->stream)); // host buffers are the caller's again
    return SF_OK;
}

extern "C" int sf_rows_nonzero(sf_ctx *ctx, const double *rows_dev, int64_t m, int64_t d, unsigned char *mask_dev)
{
    if (!ctx || !rows_dev || !mask_dev || m < 0 || d <= 0) { sf_set_error("sf_rows_nonzero: bad argument"); return SF_ERR_ARG; }
    SF_HIP(hipSetDevice(ctx->device));
    if (m) {
        SF_LAUNCH(ctx, "k8_rows_nonzero", k_rows_nonzero, dim3((unsigned)sf_div_up(m, 4)), dim3(256), rows_dev, m, d,
                  mask_dev);
    }
    return SF_OK;
}

extern "C" int sf_rows_gather(sf_ctx *ctx, const double *rows_dev, int64_t n_rows, const int64_t *sel_dev, int64_t m, int64_t d,
                              double *out_dev)
{
    if (!ctx || !rows_dev || !sel_dev || !out_dev || m < 0 || d <= 0 || n_rows < 0) { sf_set_error("sf_rows_gather: bad argument"); return SF_ERR_ARG; }
    SF_HIP(hipSetDevice(ctx->device));
    if (m) {
        SF_LAUNCH(ctx, "k8_rows_gather", k_rows_gather, dim3((unsigned)sf_div_up(m, 4)), dim3(256), rows_dev, n_rows, sel_dev, m, d,
                  out_dev, ctx->dev_flag);
    }
    return SF_OK;
}

namespace {
// second half of a sharded column arg-min (see sf_match_col_candidates)
__global__ void k_col_candidates(const double *__restrict__ local_dist, const double *__restrict__ global_dist,
                                 const int64_t *__restrict__ local_idx, int64_t row_offset, int64_t m,
                                 unsigned long long *__restrict__ cand)
{
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= m) return;
    const double l = local_dist[j];
    // (+inf: no scan row of this rank reaches the column at all -- a masked or empty block must not claim it)
    cand[j] = (l == global_dist[j] && l < INFINITY) ? (unsigned long long)(row_offset + local_idx[j]) : ~0ull;
}
} // namespace

// Reciprocity over sharded scan rows (matching.py:63-65: distance_matrix.argmin(axis=0) over ALL scan rows).  Every rank
// has the arg-min of each reference column over its own scan block (local_idx, local_dist: sf_match_argmin_multiscale with
// the operands swapped).  The column minimum over all ranks is an all-reduce(min) of the distances -- non-negative
// doubles order like their bit patterns, sf_comm_allreduce_min_u64 -- and the winner is the LOWEST scan row that
// attains it: this call turns (local, global) distances into candidates `row_offset + local_idx` where the rank attains
// the global minimum and ~0 where it does not, and a second all-reduce(min) picks the first minimum, as NumPy does.
extern "C" int sf_match_col_candidates(sf_ctx *ctx, const double *local_dist_dev, const double *global_dist_dev,
                                       const int64_t *local_idx_dev, int64_t row_offset, int64_t m, void *cand_dev)
{
    if (!ctx || !local_dist_dev || !global_dist_dev || !local_idx_dev || !cand_dev || m < 0 || row_offset < 0) {
        sf_set_error("sf_match_col_candidates: bad argument");
        return SF_ERR_ARG;
    }
    SF_HIP(hipSetDevice(ctx->device));
    if (m)
        SF_LAUNCH(ctx, "k8_col_candidates", k_col_candidates, dim3((unsigned)sf_div_up(m, 256)), dim3(256), local_dist_dev,
                  global_dist_dev, local_idx_dev, row_offset, m, (unsigned long long *)cand_dev);
    return SF_OK;
}

// The 3-D branch through the FP16 pre-filter + exact float64 re-rank of the 2-D matcher, one scale after the other, and the fold
// above (round 5; the exact tile kernel alone ran 2 x 10^4 x 10^4 x 352 in 12.6 ms = 11 TFLOP/s).  *done = false: not applicable
// (small problem, SF_MATCH_EXACT=1) or a row needs the exact kernel -- the caller runs it.  Device pointers.
static int match_multiscale_fast(sf_ctx *ctx, const double *a, const double *b, int n_scales, int64_t m1, int64_t m2, int64_t d,
                                 const unsigned char *a_ok, const unsigned char *b_ok, double max_val, int64_t *idx, double *dist,
                                 bool *done)
{
    *done = false;
    const char *fe = getenv("SF_MATCH_EXACT");
    const double work = (double)m1 * (double)m2 * (double)d;
    if ((fe && fe[0] == '1') || n_scales < 2 || work < 5e8 || m2 < 256 || !(max_val > 0.0) || std::isinf(max_val)) return SF_OK;
    sf_pool_guard tmp(ctx);
    int64_t *idx_s = nullptr;
    double *dist_s = nullptr;
    unsigned *n_hard = nullptr;
    SF_CHECK(tmp.alloc(&idx_s, (size_t)n_scales * m1));
    SF_CHECK(tmp.alloc(&dist_s, (size_t)n_scales * m1));
    SF_CHECK(tmp.alloc(&n_hard, 1));
    SF_HIP(hipMemsetAsync(n_hard, 0, sizeof(unsigned), ctx->stream));
    for (int sc = 0; sc < n_scales; ++sc)
        SF_CHECK(sf_match_gemm(ctx, a + (size_t)sc * m1 * d, m1, b + (size_t)sc * m2 * d, m2, d, idx_s + (size_t)sc * m1, dist_s + (size_t)sc * m1,
                               "k8_match_gemm", nullptr, a_ok + (size_t)sc * m1, b_ok + (size_t)sc * m2));
    SF_LAUNCH(ctx, "k8_match_merge", k_match_scale_fold, dim3((unsigned)sf_div_up(m1, 256)), dim3(256), (const int64_t *)idx_s,
              (const double *)dist_s, n_scales, m1, max_val, idx, dist, n_hard);
    void *pin = nullptr;
    SF_CHECK(sf_ctx_pinned(ctx, &pin));
    unsigned *hw = (unsigned *)((char *)pin + SF_PINNED_BYTES - 32);
    SF_HIP(hipMemcpyAsync(hw, n_hard, sizeof(unsigned), hipMemcpyDeviceToHost, ctx->stream));
    SF_HIP(hipStreamSynchronize(ctx->stream));
    *done = *hw == 0u;
    return SF_OK;
}

extern "C" int sf_match_argmin_multiscale(sf_ctx *ctx, const double *a, const double *b, int n_scales, int64_t m1,
                                          int64_t m2, int64_t d, const unsigned char *a_ok, const unsigned char *b_ok,
                                          double max_val, int64_t *idx, double *dist, int flags)
{
    if (!ctx || !a || !b || !a_ok || !b_ok || !idx || n_scales < 1 || m1 < 0 || m2 <= 0 || d <= 0) {
        sf_set_error("sf_match_argmin_multiscale: bad argument");
        return SF_ERR_ARG;
    }
    SF_HIP(hipSetDevice(ctx->device));
    if (flags == (SF_IN_DEVICE | SF_OUT_DEVICE)) { // everything resident: no copies
        if (!m1) return SF_OK;
        static const bool force_exact = getenv("SF_MATCH_EXACT") && getenv("SF_MATCH_EXACT")[0] == '1';
        const double work = (double)m1 * (double)m2 * (double)d;
        if (n_scales == 1 && std::isinf(max_val) && max_val > 0 && !force_exact && work >= 5e8 && m2 >= 256)
            // single scale, masked rows at +inf: the matrix-core path with ||b_j||^2 = +inf for masked reference rows
            return sf_match_gemm(ctx, a, m1, b, m2, d, idx, dist, "k8_match_gemm", nullptr, a_ok, b_ok);
        bool done = false;
        SF_CHECK(match_multiscale_fast(ctx, a, b, n_scales, m1, m2, d, a_ok, b_ok, max_val, idx, dist, &done));
        if (!done) SF_CHECK(match_one_way(ctx, a, m1, b, m2, d, idx, dist, "k8_match_tile_masked", n_scales, a_ok, b_ok, max_val));
        return SF_OK;
    }
    if (flags != SF_HOST) { sf_set_error("sf_match_argmin_multiscale: flags must be SF_HOST or SF_IN_DEVICE|SF_OUT_DEVICE"); return SF_ERR_UNSUPPORTED; }
    const size_t na = (size_t)n_scales * m1 * d, nbv = (size_t)n_scales * m2 * d;
    double *da = nullptr, *db = nullptr, *ddist = nullptr;
    unsigned char *dao = nullptr, *dbo = nullptr;
    int64_t *didx = nullptr;
    sf_pool_guard tmp(ctx);
    SF_CHECK(tmp.alloc(&da, na));
    SF_CHECK(tmp.alloc(&db, nbv));
    SF_CHECK(tmp.alloc(&dao, (size_t)n_scales * m1));
    SF_CHECK(tmp.alloc(&dbo, (size_t)n_scales * m2));
    SF_CHECK(tmp.alloc(&didx, (size_t)m1));
    SF_CHECK(tmp.alloc(&ddist, (size_t)m1));
    if (na) SF_HIP(hipMemcpyAsync(da, a, na * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    SF_HIP(hipMemcpyAsync(db, b, nbv * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    if (m1) SF_HIP(hipMemcpyAsync(dao, a_ok, (size_t)n_scales * m1, hipMemcpyHostToDevice, ctx->stream));
    SF_HIP(hipMemcpyAsync(dbo, b_ok, (size_t)n_scales * m2, hipMemcpyHostToDevice, ctx->stream));
    if (m1) {
        bool done = false;
        SF_CHECK(match_multiscale_fast(ctx, da, db, n_scales, m1, m2, d, dao, dbo, max_val, didx, ddist, &done));
        if (!done) SF_CHECK(match_one_way(ctx, da, m1, db, m2, d, didx, ddist, "k8_match_tile_multiscale", n_scales, dao, dbo, max_val));
        SF_HIP(hipMemcpyAsync(idx, didx, (size_t)m1 * sizeof(int64_t), hipMemcpyDeviceToHost, ctx->stream));
        if (dist) SF_HIP(hipMemcpyAsync(dist, ddist, (size_t)m1 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    }
    SF_HIP(hipStreamSynchronize(ctx->stream));
    return SF_OK;
}

extern "C" int sf_ransac_score(sf_ctx *ctx, const double *a, const double *b, int64_t m, const double *Rt,
                               int64_t n_draws, double thr, int64_t *inliers, int flags)
{
    if (!ctx || !a || !b || !Rt || !inliers || m < 0 || n_draws < 0) { sf_set_error("sf_ransac_score: bad argument"); return SF_ERR_ARG; }
    SF_HIP(hipSetDevice(ctx->device));
    const bool in_dev = flags & SF_IN_DEVICE, out_dev = flags & SF_OUT_DEVICE;
    sf_pool_guard tmp(ctx);
    double *da = const_cast<double *>(a), *db = const_cast<double *>(b), *dR = const_cast<double *>(Rt);
    if (!in_dev) {
        SF_CHECK(tmp.alloc(&da, (size_t)std::max<int64_t>(m * 3, 1)));
        SF_CHECK(tmp.alloc(&db, (size_t)std::max<int64_t>(m * 3, 1)));
        SF_CHECK(tmp.alloc(&dR, (size_t)std::max<int64_t>(n_draws * 12, 1)));
        if (m) {
            SF_HIP(hipMemcpyAsync(da, a, (size_t)m * 24, hipMemcpyHostToDevice, ctx->stream));
            SF_HIP(hipMemcpyAsync(db, b, (size_t)m * 24, hipMemcpyHostToDevice, ctx->stream));
        }
        if (n_draws) SF_HIP(hipMemcpyAsync(dR, Rt, (size_t)n_draws * 96, hipMemcpyHostToDevice, ctx->stream));
    }
    int64_t *dinl = inliers;
    if (!out_dev) SF_CHECK(tmp.alloc(&dinl, (size_t)std::max<int64_t>(n_draws, 1)));
    if (n_draws) {
        SF_HIP(hipMemsetAsync(dinl, 0, (size_t)n_draws * sizeof(int64_t), ctx->stream));
        if (m) {
            // sqrt(s) <= thr is certain for s <= lo = thr^2 rounded down, impossible for s >= hi = (next double)^2
            // rounded up (sqrt is monotone and correctly rounded); a negative or NaN threshold admits nothing
            double lo = -1.0, hi = -1.0;
            if (thr >= 0.0) {
                const double up = std::nextafter(thr, INFINITY);
                lo = std::nextafter(thr * thr, -INFINITY);
                hi = std::isinf(up) ? INFINITY : std::nextafter(up * up, INFINITY);
                if (std::isinf(thr)) lo = hi = INFINITY; // everything finite is an inlier; s = inf: sqrt path
            }
            const int64_t tiles = sf_div_up(m, K9_TILE);
            int64_t splits = sf_div_up(n_draws, K9_MAX_DRAWS);
            if (tiles * splits < 1024) splits = std::min<int64_t>(n_draws, sf_div_up(1024, tiles));
            if (splits > 65535) splits = 65535;
            const int64_t dps = sf_div_up(n_draws, splits);
            if (dps > K9_MAX_DRAWS) { sf_set_error("sf_ransac_score: more than 65535 x 8192 draws"); return SF_ERR_UNSUPPORTED; }
            splits = sf_div_up(n_draws, dps);
            SF_LAUNCH(ctx, "k9_ransac_score", k_ransac_score, dim3((unsigned)tiles, (unsigned)splits), dim3(256), da, db, m, dR,
                      n_draws, dps, thr, lo, hi, reinterpret_cast<unsigned long long *>(dinl));
        }
    }
    if (!out_dev && n_draws)
        SF_HIP(hipMemcpyAsync(inliers, dinl, (size_t)n_draws * sizeof(int64_t), hipMemcpyDeviceToHost, ctx->stream));
    if (!out_dev || !in_dev) SF_HIP(hipStreamSynchronize(ctx->stream));
    return SF_OK;
}


// ---- helpers of the streamed K8 (match_i8.hip: sf_match_stream_*) --------------------------------------------------------------------
namespace {
__global__ void k_rows_abs_max(const double *__restrict__ v, int64_t n, double *__restrict__ partial)
{
    double mx = 0.0;
    bool bad = false;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const double x = fabs(v[i]);
        bad |= !(x <= 1.7976931348623157e308);
        mx = fmax(mx, x);
    }
    if (bad) mx = INFINITY;
    for (int off = 32; off > 0; off >>= 1) mx = fmax(mx, __shfl_xor(mx, off));
    __shared__ double s[4];
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = fmax(fmax(s[0], s[1]), fmax(s[2], s[3]));
}
} // namespace

extern "C" int sf_rows_abs_max(sf_ctx *ctx, const double *rows_dev, int64_t m, int64_t d, double *out_host)
{
    if (!ctx || !out_host || m < 0 || d <= 0 || (m && !rows_dev)) { sf_set_error("sf_rows_abs_max: bad argument"); return SF_ERR_ARG; }
    SF_HIP(hipSetDevice(ctx->device));
    *out_host = 0.0;
    if (!m) return SF_OK;
    sf_pool_guard tmp(ctx);
    double *part = nullptr;
    SF_CHECK(tmp.alloc(&part, 1024));
    SF_LAUNCH(ctx, "k8_rows_abs_max", k_rows_abs_max, dim3(1024), dim3(256), rows_dev, m * d, part);
    std::vector<double> h(1024);
    SF_HIP(hipMemcpyAsync(h.data(), part, 1024 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    SF_HIP(hipStreamSynchronize(ctx->stream));
    double mx = 0.0;
    for (double x : h) mx = std::max(mx, x);
    *out_host = mx;
    return SF_OK;
}

// (what sf_match_stream_end falls back to: the resident, masked one-shot arg-min)
int sf_match_argmin_masked_generic(sf_ctx *ctx, const double *a, const double *b, int64_t m1, int64_t m2, int64_t d,
                                   const unsigned char *a_ok, const unsigned char *b_ok, int64_t *idx, double *dist)
{
    return sf_match_argmin_multiscale(ctx, a, b, 1, m1, m2, d, a_ok, b_ok, INFINITY, idx, dist, SF_IN_DEVICE | SF_OUT_DEVICE);
}
