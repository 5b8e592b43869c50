// match_gemm.hip -- K8 fast path: the arg-min of cdist(a, b) through the FP64 matrix cores, with the
// reference's exact result.
//
// Replaces scipy cdist + argmin (matching.py:47-52, 164-168) for large problems.  The exact kernel
// (match.hip) spends three float64 vector instructions per pair-dimension; here the ranking key
//     s(i, j) = ||b_j||^2 - 2 a_i . b_j            (= ||a_i - b_j||^2 - ||a_i||^2)
// is a GEMM: v_mfma_f64_16x16x4_f64 tiles, 128 x 128 output tile per 256-thread workgroup, LDS
// double-buffered in slices of 16 along the descriptor dimension, each wave owning a 64 x 64 sub-tile
// (16 accumulators).  Per row the best key, its column and the SECOND best key are tracked.
//
// Exactness.  The key carries a rounding error of at most delta = d * eps * (||a||^2 + 2 max||b||^2);
// the reference's own sequentially rounded distance differs from the true one by less than that too.
// If the second best key exceeds the best by more than tol = 8 d eps (||a_i||^2 + max_j ||b_j||^2),
// the best column is the reference's arg-min whatever the rounding (and the distances cannot collide
// after the square root, their relative gap being >= 4 d eps).  Such rows get their distance from a
// sequential float64 sum for that single pair (bit-identical to scipy).  Rows that fail the gap test
// (exact or near ties, e.g. duplicated descriptors) are re-run through the exact kernel, which also
// reproduces the first-minimum rule.  So the result equals the exact kernel's for every input; only
// the amount of work sent to the slow path depends on the data.
// Roofline: FP64 matrix cores (78.6 TFLOP/s), 2 m1 m2 d flop.
#include <algorithm>
#include <cmath>
#include <vector>

#include "common.h"
#include "device_util.h"

int sf_match_exact(sf_ctx *ctx, const double *da, int64_t m1, const double *db, int64_t m2, int64_t d, int64_t *didx,
                   double *ddist, const char *name, const unsigned char *a_ok, const unsigned char *b_ok); // match.hip

namespace {

typedef double d4 __attribute__((ext_vector_type(4)));
constexpr int GM = 128, GN = 128, GK = 16;
// LDS tiles are row-major [row][k], 16 doubles per row, columns swizzled by sw() (see there).
constexpr int LDS_P = 16; // no padding: bank conflicts are avoided by the XOR swizzle sw() below

// ||row||^2; rows whose mask is 0 (all-zero descriptors that must never be matched) get +inf, which keeps them
// out of every arg-min without touching the GEMM
__global__ __launch_bounds__(256) void k_row_sqnorm(const double *__restrict__ a, int64_t m, int64_t d,
                                                    const unsigned char *__restrict__ ok, double *__restrict__ out)
{
    const int lane = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= m) return;
    double s = 0.0;
    for (int64_t t = lane; t < d; t += 64) s += a[i * d + t] * a[i * d + t];
    s = sf_wave_sum(s);
    if (lane == 0) out[i] = (ok && !ok[i]) ? INFINITY : s;
}

__global__ void k_max_partial(const double *__restrict__ v, int64_t n, double *__restrict__ partial)
{
    double mx = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        mx = fmax(mx, isinf(v[i]) ? 0.0 : v[i]); // masked rows do not enter the error bound
    for (int off = 32; off > 0; off >>= 1) mx = fmax(mx, __shfl_xor(mx, off));
    __shared__ double s[4];
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = fmax(fmax(s[0], s[1]), fmax(s[2], s[3]));
}

template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v)
{
    return __hiloint2double(__builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, false),
                            __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, false));
}

// best / second-best bookkeeping for one row; ties count as "second best equal to best" (-> slow path)
struct top2 {
    double m1, m2;
    int64_t j1;
};
__device__ inline void top2_insert(top2 &t, double s, int64_t j)
{
    if (s < t.m1) { t.m2 = t.m1; t.m1 = s; t.j1 = j; }
    else if (s < t.m2) t.m2 = s;
}
__device__ inline void top2_merge(top2 &t, double om1, double om2, int64_t oj1)
{
    if (om1 < t.m1) { t.m2 = fmin(t.m1, om2); t.m1 = om1; t.j1 = oj1; }
    else t.m2 = fmin(t.m2, om1); // om1 >= t.m1 (ties land here: m2 = m1)
}

// VEC: the descriptor length is even and both matrices are 16-byte aligned -> a stage is fetched with 16-byte
// loads, eight lanes per 128-byte row segment (8 cache lines per wave instruction instead of 64).
// LDS column swizzle: a tile row holds GK = 16 doubles = four 32-byte groups; row r stores group g at slot
// g ^ (r & 3).  An MFMA fragment read (16 consecutive rows x the 4 k of one group, 8 bytes per lane) then touches
// every bank exactly once per 16 lanes -- the minimum of four passes per wave read -- where the padded layout
// (pitch 18) had rows r and r + 8 and neighbouring k colliding; and without padding the tile pair is 64 KB, so
// two workgroups fit a CU's LDS.
__device__ __forceinline__ int sw(int row, int c) { return (((c >> 2) ^ (row & 3)) << 2) | (c & 3); }

template <bool VEC>
__global__ __launch_bounds__(256, 2) void k_match_gemm(const double *__restrict__ a, int64_t m1,
                                                    const double *__restrict__ b, int64_t m2, int64_t d,
                                                    const double *__restrict__ nb, int64_t tiles_per_split,
                                                    double *__restrict__ pm1, int64_t *__restrict__ pj1,
                                                    double *__restrict__ pm2)
{
    __shared__ __attribute__((aligned(16))) double As[2][GM][LDS_P];
    __shared__ __attribute__((aligned(16))) double Bs[2][GN][LDS_P];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int64_t i0 = (int64_t)blockIdx.x * GM;
    const int split = blockIdx.y;
    const int64_t ntiles = (m2 + GN - 1) / GN;
    const int64_t jt0 = (int64_t)split * tiles_per_split;
    const int64_t jt1 = jt0 + tiles_per_split < ntiles ? jt0 + tiles_per_split : ntiles;
    const int srow = tid & 127, skh = tid >> 7; // staging: this thread's tile row and its half of the 16 k
    const int nk = (int)((d + GK - 1) / GK);
    const int l15 = lane & 15, l4 = lane >> 4;

    top2 run; // running result of the row this lane OWNS within its 16-lane DPP row: row 16*ti + l4 + 4*r
    run.m1 = INFINITY; run.m2 = INFINITY; run.j1 = 0; //   with 4*ti + r == l15

    for (int64_t jt = jt0; jt < jt1; ++jt) {
        const int64_t j0 = jt * GN;
        d4 acc[4][4];
#pragma unroll
        for (int ti = 0; ti < 4; ++ti)
#pragma unroll
            for (int tj = 0; tj < 4; ++tj) acc[ti][tj] = d4{0.0, 0.0, 0.0, 0.0};
        double ra[8], rb[8];
        auto fetch = [&](int kt) {
            if (VEC) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int p = tid + 256 * u, row = p >> 3, kp = p & 7;
                    const int64_t k = (int64_t)kt * GK + 2 * kp, ar = i0 + row, br = j0 + row;
                    double2 va = make_double2(0.0, 0.0), vb = make_double2(0.0, 0.0);
                    if (ar < m1 && k < d) va = *reinterpret_cast<const double2 *>(a + ar * d + k);
                    if (br < m2 && k < d) vb = *reinterpret_cast<const double2 *>(b + br * d + k);
                    ra[2 * u] = va.x; ra[2 * u + 1] = va.y;
                    rb[2 * u] = vb.x; rb[2 * u + 1] = vb.y;
                }
            } else {
                const int64_t kbase = (int64_t)kt * GK + skh * 8;
                const int64_t ar = i0 + srow, br = j0 + srow;
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int64_t k = kbase + u;
                    ra[u] = (ar < m1 && k < d) ? a[ar * d + k] : 0.0;
                    rb[u] = (br < m2 && k < d) ? b[br * d + k] : 0.0;
                }
            }
        };
        auto stash = [&](int buf) {
            if (VEC) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int p = tid + 256 * u, row = p >> 3, kp = p & 7;
                    *reinterpret_cast<double2 *>(&As[buf][row][sw(row, 2 * kp)]) = make_double2(ra[2 * u], ra[2 * u + 1]);
                    *reinterpret_cast<double2 *>(&Bs[buf][row][sw(row, 2 * kp)]) = make_double2(rb[2 * u], rb[2 * u + 1]);
                }
            } else {
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    As[buf][srow][sw(srow, skh * 8 + u)] = ra[u];
                    Bs[buf][srow][sw(srow, skh * 8 + u)] = rb[u];
                }
            }
        };
        fetch(0);
        stash(0);
        __syncthreads();
        for (int kt = 0; kt < nk; ++kt) {
            const int buf = kt & 1;
            if (kt + 1 < nk) fetch(kt + 1);
#pragma unroll
            for (int kk = 0; kk < GK / 4; ++kk) {
                double af[4], bf[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    af[t] = As[buf][64 * wr + 16 * t + l15][sw(l15, kk * 4 + l4)]; // (row & 3) == (l15 & 3)
                    bf[t] = Bs[buf][64 * wc + 16 * t + l15][sw(l15, kk * 4 + l4)];
                }
#pragma unroll
                for (int ti = 0; ti < 4; ++ti)
#pragma unroll
                    for (int tj = 0; tj < 4; ++tj)
                        acc[ti][tj] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[ti], bf[tj], acc[ti][tj], 0, 0, 0);
            }
            if (kt + 1 < nk) stash(buf ^ 1);
            __syncthreads();
        }
        // epilogue of this column tile: keys s = nb[j] - 2 a.b, top-2 per row over the wave's 64 columns
        double nbv[4];
        int64_t jcol[4];
#pragma unroll
        for (int tj = 0; tj < 4; ++tj) {
            jcol[tj] = j0 + 64 * wc + 16 * tj + l15;
            nbv[tj] = jcol[tj] < m2 ? nb[jcol[tj]] : INFINITY;
        }
        // Per row (TI, R) of the wave's 64 x 64 block: the four keys each lane holds are first tested against the
        // row's running SECOND-best (owned by lane 4 TI + R of the DPP row, broadcast with row_newbcast).  Only a
        // key below it can change the top-2, and after t column tiles that happens with probability ~2/t, so the
        // insert + butterfly + merge below is skipped (wave-uniformly) for almost every row of almost every tile.
#define SF_TOP2_STEP(CTRL)                                                                                          \
    {                                                                                                               \
        const double om1 = dpp_f64<CTRL>(t.m1), om2 = dpp_f64<CTRL>(t.m2);                                          \
        const int oj = __builtin_amdgcn_update_dpp(0, jloc, CTRL, 0xf, 0xf, false);                                 \
        if (om1 < t.m1 || (om1 == t.m1 && oj < jloc)) {                                                             \
            const double keep = t.m1;                                                                               \
            t.m2 = fmin(om2, keep);                                                                                 \
            t.m1 = om1;                                                                                             \
            jloc = oj;                                                                                              \
        } else {                                                                                                    \
            t.m2 = fmin(t.m2, om1);                                                                                 \
        }                                                                                                           \
    }
#define SF_EPI_ROW(TI, R)                                                                                           \
    {                                                                                                               \
        const double thr = dpp_f64<0x150 + 4 * (TI) + (R)>(run.m2);                                                 \
        double key[4];                                                                                              \
        bool below = false;                                                                                         \
        _Pragma("unroll") for (int tj = 0; tj < 4; ++tj) {                                                          \
            key[tj] = jcol[tj] < m2 ? nbv[tj] - 2.0 * acc[TI][tj][R] : INFINITY;                                    \
            below |= key[tj] < thr;                                                                                 \
        }                                                                                                           \
        if (__ballot(below)) {                                                                                      \
            top2 t;                                                                                                 \
            t.m1 = INFINITY; t.m2 = INFINITY; t.j1 = 0;                                                             \
            _Pragma("unroll") for (int tj = 0; tj < 4; ++tj) top2_insert(t, key[tj], jcol[tj]);                     \
            int jloc = (int)(t.j1 - j0); /* column inside the tile: one register through the butterfly */           \
            SF_TOP2_STEP(0xB1)  /* quad_perm [1,0,3,2] */                                                           \
            SF_TOP2_STEP(0x4E)  /* quad_perm [2,3,0,1] */                                                           \
            SF_TOP2_STEP(0x141) /* row_half_mirror */                                                               \
            SF_TOP2_STEP(0x140) /* row_mirror: the merge is symmetric, every lane of the row ends with the same triple */ \
            if (l15 == 4 * (TI) + (R)) top2_merge(run, t.m1, t.m2, j0 + jloc);                                      \
        }                                                                                                           \
    }
#define SF_EPI_TI(TI) SF_EPI_ROW(TI, 0) SF_EPI_ROW(TI, 1) SF_EPI_ROW(TI, 2) SF_EPI_ROW(TI, 3)
        SF_EPI_TI(0) SF_EPI_TI(1) SF_EPI_TI(2) SF_EPI_TI(3)
#undef SF_EPI_TI
#undef SF_EPI_ROW
#undef SF_TOP2_STEP
    }
    // merge the two column halves (waves wc = 0, 1 of the same row half) through LDS and write the partials
    __syncthreads();
    double *sm1 = &As[0][0][0], *sm2 = &As[1][0][0];
    int64_t *sj1 = reinterpret_cast<int64_t *>(&Bs[0][0][0]);
    // the row this lane owns: ti = l15 >> 2, r = l15 & 3 -> 64*wr + 16*ti + l4 + 4*r
    const int own = 64 * wr + 16 * (l15 >> 2) + l4 + 4 * (l15 & 3);
    if (wc == 1) { sm1[own] = run.m1; sm2[own] = run.m2; sj1[own] = run.j1; }
    __syncthreads();
    if (wc == 0) {
        // columns of wc = 1 are all larger than those of wc = 0 inside a tile, but tiles interleave: merge by value
        const double om1 = sm1[own], om2 = sm2[own];
        const int64_t oj1 = sj1[own];
        if (om1 < run.m1 || (om1 == run.m1 && oj1 < run.j1)) {
            const double keep = run.m1;
            run.m2 = fmin(om2, keep);
            run.m1 = om1;
            run.j1 = oj1;
        } else {
            run.m2 = fmin(run.m2, om1);
        }
        const int64_t i = i0 + own;
        if (i < m1) {
            pm1[(int64_t)split * m1 + i] = run.m1;
            pm2[(int64_t)split * m1 + i] = run.m2;
            pj1[(int64_t)split * m1 + i] = run.j1;
        }
    }
}

// Merge the column splits, apply the gap test, and for decided rows compute the reference's distance with a
// sequential float64 sum (one lane per row, exactly scipy's loop).  Undecided rows are flagged.
__global__ void k_match_decide(const double *__restrict__ a, int64_t m1, const double *__restrict__ b, int64_t d,
                               const double *__restrict__ pm1, const int64_t *__restrict__ pj1,
                               const double *__restrict__ pm2, int nsplit, double nb_max,
                               const unsigned char *__restrict__ a_ok, int64_t *__restrict__ idx,
                               double *__restrict__ dist, int *__restrict__ flag, int *__restrict__ n_flagged)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m1) return;
    if (a_ok && !a_ok[i]) { // masked scan row: +inf from everything, first column (the exact kernel's answer)
        idx[i] = 0;
        if (dist) dist[i] = INFINITY;
        flag[i] = 0;
        return;
    }
    double bm1 = pm1[i], bm2 = pm2[i];
    int64_t bj = pj1[i];
    for (int s = 1; s < nsplit; ++s) {
        const double om1 = pm1[(int64_t)s * m1 + i], om2 = pm2[(int64_t)s * m1 + i];
        const int64_t oj = pj1[(int64_t)s * m1 + i];
        if (om1 < bm1) { bm2 = fmin(bm1, om2); bm1 = om1; bj = oj; }
        else bm2 = fmin(bm2, om1);
    }
    double na = 0.0, acc = 0.0;
    const double *ai = a + i * d, *bjp = b + bj * d;
    for (int64_t t = 0; t < d; ++t) {
        const double av = ai[t], df = av - bjp[t];
        na += av * av;
        acc += df * df; // left to right, no FMA: scipy's euclidean loop
    }
    const double tol = 8.0 * (double)d * 1.1102230246251565e-16 * (na + nb_max);
    const bool decided = (bm2 - bm1) > tol; // false for NaN as well
    idx[i] = bj;
    if (dist) dist[i] = sqrt(acc);
    flag[i] = decided ? 0 : 1;
    if (!decided) atomicAdd(n_flagged, 1);
}

__global__ void k_gather_rows(const double *__restrict__ a, int64_t d, const int64_t *__restrict__ rows, int64_t nr,
                              double *__restrict__ out)
{
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= nr * d) return;
    const int64_t r = g / d, t = g - r * d;
    out[g] = a[rows[r] * d + t];
}

__global__ void k_scatter_results(const int64_t *__restrict__ rows, int64_t nr, const int64_t *__restrict__ sidx,
                                  const double *__restrict__ sdist, int64_t *__restrict__ idx, double *__restrict__ dist)
{
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= nr) return;
    idx[rows[g]] = sidx[g];
    if (dist) dist[rows[g]] = sdist[g];
}

} // namespace

// Row arg-min of cdist(a, b) with the exact kernel's result; returns the number of rows that needed the slow path.
int sf_match_gemm_f64(sf_ctx *ctx, const double *da, int64_t m1, const double *db, int64_t m2, int64_t d, int64_t *didx,
                  double *ddist, const char *name, int64_t *n_slow, const unsigned char *a_ok, const unsigned char *b_ok)
{
    if (n_slow) *n_slow = 0;
    if (!m1) return SF_OK;
    sf_pool_guard tmp(ctx);
    double *nb = nullptr, *part = nullptr;
    SF_CHECK(tmp.alloc(&nb, (size_t)m2));
    SF_CHECK(tmp.alloc(&part, (size_t)256));
    SF_LAUNCH(ctx, "k8_row_sqnorm", k_row_sqnorm, dim3((unsigned)sf_div_up(m2, 4)), dim3(256), db, m2, d, b_ok, nb);
    SF_LAUNCH(ctx, "k8_max_partial", k_max_partial, dim3(256), dim3(256), (const double *)nb, m2, part);
    const int64_t row_tiles = sf_div_up(m1, GM), col_tiles = sf_div_up(m2, GN);
    int64_t nsplit = 1;
    if (row_tiles < 1024) nsplit = std::min<int64_t>(col_tiles, sf_div_up(1024, row_tiles));
    if (nsplit > 65535) nsplit = 65535;
    const int64_t tiles_per_split = sf_div_up(col_tiles, nsplit);
    nsplit = sf_div_up(col_tiles, tiles_per_split);
    double *pm1 = nullptr, *pm2 = nullptr;
    int64_t *pj1 = nullptr;
    int *flag = nullptr, *nflag = nullptr;
    SF_CHECK(tmp.alloc(&pm1, (size_t)(nsplit * m1)));
    SF_CHECK(tmp.alloc(&pm2, (size_t)(nsplit * m1)));
    SF_CHECK(tmp.alloc(&pj1, (size_t)(nsplit * m1)));
    SF_CHECK(tmp.alloc(&flag, (size_t)m1));
    SF_CHECK(tmp.alloc(&nflag, (size_t)1));
    SF_HIP(hipMemsetAsync(nflag, 0, sizeof(int), ctx->stream));
    const bool vec = (d % 2 == 0) && ((uintptr_t)da % 16 == 0) && ((uintptr_t)db % 16 == 0);
    if (vec) {
        SF_LAUNCH(ctx, name, k_match_gemm<true>, dim3((unsigned)row_tiles, (unsigned)nsplit), dim3(256), da, m1, db, m2, d,
                  (const double *)nb, tiles_per_split, pm1, pj1, pm2);
    } else {
        SF_LAUNCH(ctx, name, k_match_gemm<false>, dim3((unsigned)row_tiles, (unsigned)nsplit), dim3(256), da, m1, db, m2, d,
                  (const double *)nb, tiles_per_split, pm1, pj1, pm2);
    }
    std::vector<double> hpart(256);
    SF_HIP(hipMemcpyAsync(hpart.data(), part, 256 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    SF_HIP(hipStreamSynchronize(ctx->stream));
    double nb_max = 0.0;
    for (double v : hpart) nb_max = std::max(nb_max, v);
    SF_LAUNCH(ctx, "k8_match_decide", k_match_decide, dim3((unsigned)sf_div_up(m1, 256)), dim3(256), da, m1, db, d,
              (const double *)pm1, (const int64_t *)pj1, (const double *)pm2, (int)nsplit, nb_max, a_ok, didx, ddist,
              flag, nflag);
    int nf = 0;
    SF_HIP(hipMemcpyAsync(&nf, nflag, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    SF_HIP(hipStreamSynchronize(ctx->stream));
    int rc = SF_OK;
    if (nf > 0) {
        // slow path for the undecided rows: exact kernel on the gathered rows, results scattered back
        std::vector<int> hflag((size_t)m1);
        SF_HIP(hipMemcpyAsync(hflag.data(), flag, (size_t)m1 * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
        SF_HIP(hipStreamSynchronize(ctx->stream));
        std::vector<int64_t> rows;
        rows.reserve((size_t)nf);
        for (int64_t i = 0; i < m1; ++i)
            if (hflag[(size_t)i]) rows.push_back(i);
        const int64_t nr = (int64_t)rows.size();
        int64_t *drows = nullptr, *sidx = nullptr;
        double *sub = nullptr, *sdist = nullptr;
        SF_CHECK(tmp.alloc(&drows, (size_t)nr));
        SF_CHECK(tmp.alloc(&sidx, (size_t)nr));
        SF_CHECK(tmp.alloc(&sdist, (size_t)nr));
        SF_CHECK(tmp.alloc(&sub, (size_t)(nr * d)));
        SF_HIP(hipMemcpyAsync(drows, rows.data(), (size_t)nr * sizeof(int64_t), hipMemcpyHostToDevice, ctx->stream));
        SF_LAUNCH(ctx, "k8_gather_rows", k_gather_rows, dim3((unsigned)sf_div_up(nr * d, 256)), dim3(256), da, d,
                  (const int64_t *)drows, nr, sub);
        rc = sf_match_exact(ctx, sub, nr, db, m2, d, sidx, sdist, "k8_match_tile_slowpath", nullptr, b_ok);
        if (rc == SF_OK) {
            SF_LAUNCH(ctx, "k8_scatter_results", k_scatter_results, dim3((unsigned)sf_div_up(nr, 256)), dim3(256),
                      (const int64_t *)drows, nr, (const int64_t *)sidx, (const double *)sdist, didx, ddist);
        }
        SF_HIP(hipStreamSynchronize(ctx->stream)); // rows.data() is a host buffer
        if (n_slow) *n_slow = nr;
    }
    return rc;
}

int sf_match_half_mode(); // match_half.hip
int sf_match_i8_mode();   // match_i8.hip
int sf_match_i8(sf_ctx *ctx, const double *da, int64_t m1, const double *db, int64_t m2, int64_t d, int64_t *didx,
                double *ddist, const char *name, int64_t *n_slow, const unsigned char *a_ok, const unsigned char *b_ok,
                int *used);
int sf_match_half(sf_ctx *ctx, const double *da, int64_t m1, const double *db, int64_t m2, int64_t d, int64_t *didx,
                  double *ddist, const char *name, int64_t *n_slow, const unsigned char *a_ok, const unsigned char *b_ok,
                  int *used);

// The matrix-core paths: FP16 pre-filter + float64 decision (match_half.hip) when the problem is large enough to
// pay for the conversion passes, the FP64 GEMM otherwise.  Same result either way.
int sf_match_gemm(sf_ctx *ctx, const double *da, int64_t m1, const double *db, int64_t m2, int64_t d, int64_t *didx,
                  double *ddist, const char *name, int64_t *n_slow, const unsigned char *a_ok, const unsigned char *b_ok)
{
    const int mode = sf_match_half_mode(), mode8 = sf_match_i8_mode();
    const double work = (double)m1 * (double)m2 * (double)d;
    // the integer pre-filter in front of everything, where its conversion passes and its pilot slab are small change: 2.2 x the
    // FP16 pass's pairs per second on the rows that have a clear nearest descriptor, the FP16 pass for the others (match_i8.hip)
    if (mode8 == 1 || (mode8 < 0 && mode != 0 && work >= 1e13 && m1 >= 65536)) {
        int used = 0;
        const int rc = sf_match_i8(ctx, da, m1, db, m2, d, didx, ddist, "k8_match_i8", n_slow, a_ok, b_ok, &used);
        if (rc != SF_OK || used) return rc;
    }
    if (mode == 1 || (mode < 0 && work >= 2e10 && m1 >= 2048)) {
        int used = 0;
        const int rc = sf_match_half(ctx, da, m1, db, m2, d, didx, ddist, "k8_match_half", n_slow, a_ok, b_ok, &used);
        if (rc != SF_OK || used) return rc;
    }
    return sf_match_gemm_f64(ctx, da, m1, db, m2, d, didx, ddist, name, n_slow, a_ok, b_ok);
}
