// knn.hip -- the k nearest neighbours on the grid: replaces KDTree.query(Q, k) (pca_based_descriptors.py:46, the k-NN branch of
// compute_normals; icp.py's per-iteration query with k = 1).  (Until round 6: part of search.hip.)
#include <rocprim/device/device_segmented_radix_sort.hpp>
#include <rocprim/device/device_select.hpp>
#include <rocprim/iterator/counting_iterator.hpp>

#include <algorithm>
#include <cmath>
#include <numeric>

#include "search_util.h"

namespace {

// --------------------------------------------------------------------------------------------------
// k nearest neighbours on the grid -- replaces KDTree.query(Q, k) (pca_based_descriptors.py:46, the k-NN
// branch of compute_normals).  One wave per query sweeps the 27-cell stencil of a grid whose cell edge is
// >= R, so every point within distance R of the query is seen.  The best k candidates so far live in LDS,
// kept sorted by (d2, position): a chunk's candidates not worse than the current k-th best are appended and
// the buffer is pruned back to k by rank counting whenever it could overflow.  If fewer than k points lie
// within R the query is flagged and the host retries it with a doubled R (coarser grid).
// EPL = buffer entries per lane (capacity 64*EPL >= k + 64).
// --------------------------------------------------------------------------------------------------
template <int EPL>
__global__ __launch_bounds__(64) void k_knn(sf_grid_desc g, const int32_t *__restrict__ cell_start,
                                            const double *__restrict__ xs, const double *__restrict__ ys,
                                            const double *__restrict__ zs, const double *__restrict__ qx,
                                            const double *__restrict__ qy, const double *__restrict__ qz,
                                            const int32_t *__restrict__ qsel, int64_t msel, int k, double R2,
                                            const int32_t *__restrict__ perm, int32_t *__restrict__ idx_out,
                                            int32_t *__restrict__ status)
{
    // idx_out receives ORIGINAL point indices (perm[position]): retries rebuild the grid, which renumbers the
    // cell-sorted positions, so positions of different rounds would not be comparable
    constexpr int CAP = 64 * EPL;
    __shared__ double bd[CAP];
    __shared__ int bj[CAP];
    const int lane = threadIdx.x;
    const int64_t slot = sf_xcd_block();
    if (slot >= msel) return;
    const int64_t q = qsel ? qsel[slot] : slot;
    const double px = qx[q], py = qy[q], pz = qz[q];
    int x0, x1, y0, y1, z0, z1;
    stencil_bounds(px, g.lo[0], g.inv_cell, g.dim[0] / g.xsub, x0, x1); // edge-sized cells along x ...
    x0 *= g.xsub;                                                      // ... as a range of fine ones
    x1 = x1 * g.xsub + (g.xsub - 1);
    stencil_bounds(py, g.lo[1], g.inv_cell, g.dim[1], y0, y1);
    stencil_bounds(pz, g.lo[2], g.inv_cell, g.dim[2], z0, z1);
    x0 = sf_uniform(x0); x1 = sf_uniform(x1);
    y0 = sf_uniform(y0); y1 = sf_uniform(y1);
    z0 = sf_uniform(z0); z1 = sf_uniform(z1);
    int cnt = 0, within = 0;
    double tau = INFINITY; // current k-th best d2 once the buffer holds k entries

    // keep the k best of the first `cnt` entries, sorted by (d2, position)
    auto prune = [&]() {
        double d[EPL];
        int j[EPL], rank[EPL];
#pragma unroll
        for (int u = 0; u < EPL; ++u) {
            const int i = lane + 64 * u;
            d[u] = i < cnt ? bd[i] : INFINITY;
            j[u] = i < cnt ? bj[i] : 0x7fffffff;
            rank[u] = 0;
        }
        for (int l = 0; l < cnt; ++l) {
            const double dl = bd[l];
            const int jl = bj[l];
#pragma unroll
            for (int u = 0; u < EPL; ++u) rank[u] += (dl < d[u]) || (dl == d[u] && jl < j[u]);
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < EPL; ++u)
            if (lane + 64 * u < cnt && rank[u] < k) { bd[rank[u]] = d[u]; bj[rank[u]] = j[u]; }
        __syncthreads();
        if (cnt >= k) { cnt = k; tau = bd[k - 1]; }
    };

    for (int cz = z0; cz <= z1; ++cz)
        for (int cy = y0; cy <= y1; ++cy) {
            const int64_t row = ((int64_t)cz * g.dim[1] + cy) * g.dim[0];
            const int s = cell_start[row + x0], e = cell_start[row + x1 + 1];
            for (int j0 = s; j0 < e; j0 += 64) {
                const int j = j0 + lane;
                bool cand = false, keep = false;
                double d2 = 0.0;
                if (j < e) {
                    const double dx = xs[j] - px, dy = ys[j] - py, dz = zs[j] - pz;
                    d2 = (dx * dx + dy * dy) + dz * dz;
                    cand = d2 <= R2;
                    keep = cand && d2 <= tau;
                }
                within += __popcll(__ballot(cand));
                const unsigned long long mask = __ballot(keep);
                if (keep) {
                    const int pos = cnt + sf_prefix_count(mask);
                    bd[pos] = d2;
                    bj[pos] = j;
                }
                cnt += __popcll(mask);
                __syncthreads();
                if (cnt > CAP - 64) prune();
            }
        }
    if (cnt > 0) prune(); // final order: the k best, nearest first
    if (within >= k) {
        for (int i = lane; i < k; i += 64) idx_out[q * (int64_t)k + i] = perm[bj[i]];
        if (lane == 0) status[q] = 0;
    } else if (lane == 0) {
        status[q] = 1;
    }
}

// --------------------------------------------------------------------------------------------------
// k nearest neighbours for k <= 64 on K2's mapping (round 5; k_knn above -- one wave per query over the whole 27-cell stencil, the
// k best pruned in LDS again and again -- took 3.7 ms per 1M queries at k = 30 against K2's 0.46 ms for lists of 110).
// Set-up and sweep are k_radius's: four queries per wave, the run tables of all four built at once, runs clipped in x to what
// the ball of radius R can reach, two candidates per lane and load.  The hits (d2 <= R^2) of a query go to an LDS list of
// SF_KNN_CAP entries in scan order -- which is ascending POSITION order -- and ONE rank pass then orders them: entry e's rank
// is the number of entries with a smaller d2 (every entry is broadcast from LDS once and compared by all lanes: two vector
// instructions per entry and 64 lanes); entries of rank < k write themselves to slot `rank`.  Ties in d2 (duplicated points;
// exact ties are otherwise one in millions) give equal ranks: a slot then stays empty, or k + 1 entries qualify -- both are
// seen by one ballot, and the pass is repeated with the tie broken by position, KDTree.query's order here as in k_knn.
// Two cuts in front of the rank pass: (a) a list of more than 80 entries is first reduced to the entries at or below an UPPER
// BOUND of the k-th smallest d2 -- the largest of the k smallest of a 64-entry subset (every stride-th entry, ranked among
// themselves): 64 compare steps buy a rank pass over ~ k x length / 64 entries instead of `length` x 2 tiers; (b) a query with
// more than SF_KNN_CAP points within R (a dense spot; a retry at a doubled radius) takes its bound from the part of the list
// that was kept and sweeps ONCE MORE with that bound as its radius.
// status: 0 answered, 1 fewer than k points within R (the host retries with a doubled R), 2 more than SF_KNN_CAP points within
// the bound of (b) (the host hands the query to k_knn at the same R).
// --------------------------------------------------------------------------------------------------
#ifndef SF_KNN_CAP
#define SF_KNN_CAP 256
#endif
template <int NT, bool TIES>
__device__ __forceinline__ void knn_rank_pass(const double *kd, int total, int lane, int (&rank)[4])
{
    double d[NT];
#pragma unroll
    for (int u = 0; u < NT; ++u) {
        const int e = lane + 64 * u;
        d[u] = e < total ? kd[e] : INFINITY;
        rank[u] = 0;
    }
#pragma unroll 4
    for (int l = 0; l < total; ++l) {
        const double b = kd[l]; // (one address for the whole wave: a broadcast read)
#pragma unroll
        for (int u = 0; u < NT; ++u) rank[u] += TIES ? ((b < d[u]) || (b == d[u] && l < lane + 64 * u)) : (b < d[u]);
    }
}

template <bool SEL>
__global__ __launch_bounds__(64 * SF_K2_WPB) void k_knn4(sf_grid_desc g, const int32_t *__restrict__ cell_start,
                                              const double *__restrict__ xs, const double *__restrict__ ys,
                                              const double *__restrict__ zs, const double *__restrict__ qx,
                                              const double *__restrict__ qy, const double *__restrict__ qz, int64_t m, double r2,
                                              int k, const int32_t *__restrict__ perm, int32_t *__restrict__ idx_out,
                                              int32_t *__restrict__ status, const int32_t *__restrict__ qsel)
{
    const int lane = threadIdx.x & 63, sl = lane & 15, rw = lane >> 4;
    const int64_t q0 = sf_uniform64((sf_xcd_block() * SF_K2_WPB + (threadIdx.x >> 6)) * 4);
    if (q0 >= m) return;
    const int nq = (int)(m - q0 < 4 ? m - q0 : 4);
    const int64_t qm0 = q0 + (rw < nq ? rw : 0);
    const int64_t qm = SEL ? (int64_t)qsel[qm0] : qm0;
    const double pxv = qx[qm], pyv = qy[qm], pzv = qz[qm]; // this row's query
    int y0, y1, z0, z1;
    stencil_bounds(pyv, g.lo[1], g.inv_cell, g.dim[1], y0, y1);
    stencil_bounds(pzv, g.lo[2], g.inv_cell, g.dim[2], z0, z1);
    __shared__ int4 runs[SF_K2_WPB][4][12];
    __shared__ double kd_all[SF_K2_WPB][SF_KNN_CAP];
    __shared__ int kj_all[SF_K2_WPB][SF_KNN_CAP];
    __shared__ int ks_all[SF_K2_WPB][64];
    int4(*const tabs)[12] = runs[threadIdx.x >> 6];
    double *const kd = kd_all[threadIdx.x >> 6];
    int *const kj = kj_all[threadIdx.x >> 6];
    int *const ks = ks_all[threadIdx.x >> 6];
    int first_slot = 0;
    { // (the run tables: k_radius, which has the reasoning)
        const int r = sl < 9 ? sl : 8;
        const int cz = z0 + r / 3, cy = y0 + r % 3;
        bool ok = sl < 9 && rw < nq && cz <= z1 && cy <= y1;
        const int64_t row = ((int64_t)(ok ? cz : z0) * g.dim[1] + (ok ? cy : y0)) * g.dim[0];
        const double pxr = pxv - g.lo[0], pyr = pyv - g.lo[1], pzr = pzv - g.lo[2];
        const double by0 = (double)cy * g.cell, bz0 = (double)cz * g.cell;
        const double slack_y = 1e-9 * g.cell + 1e-15 * (fabs(pyr) + by0 + g.cell);
        const double slack_z = 1e-9 * g.cell + 1e-15 * (fabs(pzr) + bz0 + g.cell);
        const double dy = fmax(fmax(by0 - pyr, pyr - (by0 + g.cell)) - slack_y, 0.0);
        const double dz = fmax(fmax(bz0 - pzr, pzr - (bz0 + g.cell)) - slack_z, 0.0);
        const double w2 = (r2 * (1.0 + 1e-9) - dy * dy) - dz * dz;
        ok = ok && w2 >= 0.0;
        const double w = sf_sqrt_fast(fmax(w2, 0.0)) * (1.0 + 1e-9) + 1e-9 * g.cell +
                         1e-15 * (fabs(pxr) + (double)g.dim[0] * (g.cell / (double)g.xsub));
        int s = 0, e = 0;
        if (sl < 9) {
            s = cell_start[row + sf_cell_coord(pxr - w, 0.0, g.inv_cell_x, g.dim[0])];
            e = cell_start[row + sf_cell_coord(pxr + w, 0.0, g.inv_cell_x, g.dim[0]) + 1];
        }
        if (!ok) { s = 0; e = 0; }
        const int base = s & ~1;
        const int npairs = (e - base + 1) >> 1;
        int inc = npairs;
        inc += __builtin_amdgcn_update_dpp(0, inc, 0x111, 0xf, 0xf, false);
        inc += __builtin_amdgcn_update_dpp(0, inc, 0x112, 0xf, 0xf, false);
        inc += __builtin_amdgcn_update_dpp(0, inc, 0x114, 0xf, 0xf, false);
        inc += __builtin_amdgcn_update_dpp(0, inc, 0x118, 0xf, 0xf, false);
        first_slot = inc - npairs;
        if (sl < 12) tabs[rw][sl] = make_int4(base - 2 * first_slot, s, e, first_slot);
    }
    __builtin_amdgcn_wave_barrier();
    for (int qi = 0; qi < nq; ++qi) {
        const int64_t q = SEL ? (int64_t)sf_uniform(__shfl((int)qm, 16 * qi)) : q0 + qi;
        const int4 *const tab = tabs[qi];
        const double px = __shfl(pxv, 16 * qi), py = __shfl(pyv, 16 * qi), pz = __shfl(pzv, 16 * qi);
        const int b4 = __shfl(first_slot, 16 * qi + 4), b8 = __shfl(first_slot, 16 * qi + 8);
        const int nslots = sf_uniform(__shfl(first_slot, 16 * qi + 9));
        // one sweep of the query's candidate pairs: the points with d2 <= thr2 go to the LDS list in scan order (= ascending
        // position); returns how many there are (the list keeps the first SF_KNN_CAP)
        auto sweep = [&](double thr2) -> int {
            int total = 0;
            for (int f0 = 0; f0 < nslots; f0 += 64) {
                const int f = f0 + lane;
                int r = f >= b4 ? 4 : 0;
                r += f >= tab[r + 2].w ? 2 : 0;
                r += f >= tab[r + 1].w ? 1 : 0;
                r = f >= b8 ? 8 : r;
                const int4 t = tab[r];
                const bool live = f < nslots;
                const int j = live ? t.x + 2 * f : 0;
                const bool in0 = live & (j >= t.y), in1 = live & (j + 1 < t.z);
                const double2 X = *reinterpret_cast<const double2 *>(xs + j);
                const double2 Y = *reinterpret_cast<const double2 *>(ys + j);
                const double2 Z = *reinterpret_cast<const double2 *>(zs + j);
                const double dxa = X.x - px, dya = Y.x - py, dza = Z.x - pz;
                const double dxb = X.y - px, dyb = Y.y - py, dzb = Z.y - pz;
                const double d2a = (dxa * dxa + dya * dya) + dza * dza, d2b = (dxb * dxb + dyb * dyb) + dzb * dzb;
                const bool hit0 = in0 & (d2a <= thr2);
                const bool hit1 = in1 & (d2b <= thr2);
                const unsigned long long m0 = __ballot(hit0), m1 = __ballot(hit1);
                const int pos = total + sf_prefix_count(m0) + sf_prefix_count(m1);
                const int pos1 = pos + (hit0 ? 1 : 0);
                if (hit0 && pos < SF_KNN_CAP) { kd[pos] = d2a; kj[pos] = j; }
                if (hit1 && pos1 < SF_KNN_CAP) { kd[pos1] = d2b; kj[pos1] = j + 1; }
                total += __popcll(m0) + __popcll(m1);
            }
            __builtin_amdgcn_wave_barrier(); // (the list is written and read by this wave only; its LDS operations stay in order)
            return total;
        };
        // An upper bound of the k-th smallest d2 from a SUBSET of the list: at most 64 entries, every `stride`-th (scan order walks
        // the stencil layer by layer, so a stride spreads the subset over the ball), ranked among themselves; the largest of
        // the subset's k smallest has at least k list entries at or below it.  +inf when the subset is smaller than k.
        auto bound_from_subset = [&](int have, int stride) -> double {
            const int ns = (have + stride - 1) / stride; // <= 64
            if (ns < k) return INFINITY;
            const double ds = lane < ns ? kd[lane * stride] : INFINITY;
            int rk = 0;
#pragma unroll 4
            for (int l = 0; l < ns; ++l) rk += kd[l * stride] < ds;
            return sf_wave_max_nonneg(lane < ns && rk < k ? ds : 0.0);
        };
        int total = sweep(r2);
        if (total < k) { // (wave-uniform) fewer than k points within R: the host retries with a doubled radius
            if (lane == 0) status[q] = 1;
            continue;
        }
        if (total > SF_KNN_CAP) {
            // more points within R than the list holds (a dense spot, or a retry at a doubled radius): a bound from the part
            // of the list that was kept, and the sweep once more with that bound as its radius
            const double tau = bound_from_subset(SF_KNN_CAP, SF_KNN_CAP / 64);
            __builtin_amdgcn_wave_barrier();
            total = tau < r2 ? sweep(tau) : total;
            if (total > SF_KNN_CAP) { // (still too many: k_knn at the same R, sf_knn_search)
                if (lane == 0) status[q] = 2;
                continue;
            }
        }
        // lists of more than 80 entries: drop what cannot be among the k nearest before the rank pass (a rank pass costs the
        // list's length x its number of 64-entry tiers; the bound costs 64 and the pass after it ~ k x length / 64 entries)
        if (total > 80) {
            const int stride = (total + 63) >> 6;
            const double tau = bound_from_subset(total, stride);
            if (tau < INFINITY) { // (wave-uniform) compaction in place, order kept: every tier is read before any is written
                double dv[4];
                int jv[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int e = lane + 64 * u;
                    dv[u] = e < total ? kd[e] : INFINITY;
                    jv[u] = e < total ? kj[e] : 0;
                }
                __builtin_amdgcn_wave_barrier();
                int kept = 0;
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (64 * u < total) { // (wave-uniform)
                        const bool keep = dv[u] <= tau;
                        const unsigned long long mk = __ballot(keep);
                        if (keep) {
                            const int pos = kept + sf_prefix_count(mk);
                            kd[pos] = dv[u];
                            kj[pos] = jv[u];
                        }
                        kept += __popcll(mk);
                    }
                }
                total = kept; // (>= k: the subset's k smallest are among the kept)
                __builtin_amdgcn_wave_barrier();
            }
        }
        int rank[4] = {0, 0, 0, 0};
        for (int attempt = 0; attempt < 2; ++attempt) { // 0: strict ranks; 1: ties broken by position (only if attempt 0 saw one)
            if (attempt == 0) {
                if (total <= 64) knn_rank_pass<1, false>(kd, total, lane, rank);
                else if (total <= 128) knn_rank_pass<2, false>(kd, total, lane, rank);
                else knn_rank_pass<4, false>(kd, total, lane, rank);
            } else {
                knn_rank_pass<4, true>(kd, total, lane, rank);
            }
            ks[lane] = -1;
            __builtin_amdgcn_wave_barrier();
            int nwin = 0;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int e = lane + 64 * u;
                const bool win = 64 * u < total && e < total && rank[u] < k;
                if (win) ks[rank[u]] = e;
                nwin += __popcll(__ballot(win));
            }
            __builtin_amdgcn_wave_barrier();
            const bool hole = lane < k && ks[lane] < 0;
            if (nwin == k && !__ballot(hole)) break;
        }
        if (lane < k) idx_out[q * (int64_t)k + lane] = perm[kj[ks[lane]]];
        if (lane == 0) status[q] = 0;
        __builtin_amdgcn_wave_barrier(); // (the next query's sweep overwrites the list)
    }
}

// original index -> cell-sorted position of the FINAL grid, for every stored neighbour
__global__ void k_knn_to_positions(int64_t total, const int32_t *__restrict__ inv_perm, int32_t *__restrict__ idx)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < total) idx[i] = inv_perm[idx[i]];
}

__global__ void k_knn_fill_csr(int64_t m, int k, int32_t *__restrict__ count, int64_t *__restrict__ offset)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i > m) return;
    if (i < m) count[i] = k;
    offset[i] = i * (int64_t)k;
}

// ---- k above the LDS buffer of k_knn (k > 1984): count, fill, sort ------------------------------------------------------
// Same stencil sweep.  k_knn_sweep<false> counts the points within R of each selected query; k_knn_sweep<true> writes
// (d2, position) of those points to the query's segment of a global list -- in scan order, which IS ascending position
// (rows of cells are visited in cell order).  A segmented radix sort on d2 and k_knn_take (ties on d2: lower position
// first, whatever the sort did with them) then leave the k nearest, nearest first: KDTree.query's answer for any k <= n.
template <bool FILL>
__global__ __launch_bounds__(64) void k_knn_sweep(sf_grid_desc g, const int32_t *__restrict__ cell_start,
                                                  const double *__restrict__ xs, const double *__restrict__ ys,
                                                  const double *__restrict__ zs, const double *__restrict__ qx,
                                                  const double *__restrict__ qy, const double *__restrict__ qz,
                                                  const int32_t *__restrict__ qsel, int64_t msel, double R2,
                                                  int32_t *__restrict__ count, const int64_t *__restrict__ seg,
                                                  double *__restrict__ d2_out, int32_t *__restrict__ pos_out)
{
    const int lane = threadIdx.x;
    const int64_t slot = sf_xcd_block();
    if (slot >= msel) return;
    const int64_t q = qsel ? qsel[slot] : slot;
    const double px = qx[q], py = qy[q], pz = qz[q];
    int x0, x1, y0, y1, z0, z1;
    stencil_bounds(px, g.lo[0], g.inv_cell, g.dim[0] / g.xsub, x0, x1);
    x0 *= g.xsub;
    x1 = x1 * g.xsub + (g.xsub - 1);
    stencil_bounds(py, g.lo[1], g.inv_cell, g.dim[1], y0, y1);
    stencil_bounds(pz, g.lo[2], g.inv_cell, g.dim[2], z0, z1);
    x0 = sf_uniform(x0); x1 = sf_uniform(x1);
    y0 = sf_uniform(y0); y1 = sf_uniform(y1);
    z0 = sf_uniform(z0); z1 = sf_uniform(z1);
    int64_t at = FILL ? seg[slot] : 0;
    int within = 0;
    for (int cz = z0; cz <= z1; ++cz)
        for (int cy = y0; cy <= y1; ++cy) {
            const int64_t row = ((int64_t)cz * g.dim[1] + cy) * g.dim[0];
            const int s = cell_start[row + x0], e = cell_start[row + x1 + 1];
            for (int j0 = s; j0 < e; j0 += 64) {
                const int j = j0 + lane;
                bool cand = false;
                double d2 = 0.0;
                if (j < e) {
                    const double dx = xs[j] - px, dy = ys[j] - py, dz = zs[j] - pz;
                    d2 = (dx * dx + dy * dy) + dz * dz;
                    cand = d2 <= R2;
                }
                const unsigned long long mask = __ballot(cand);
                if (FILL && cand) {
                    const int64_t o = at + sf_prefix_count(mask);
                    d2_out[o] = d2;
                    pos_out[o] = j;
                }
                const int c = __popcll(mask);
                at += c;
                within += c;
            }
        }
    if (!FILL && lane == 0) count[slot] = within;
}

// sorted-by-d2 segment -> the k nearest in (d2, position) order, as ORIGINAL indices (see k_knn)
__global__ __launch_bounds__(256) void k_knn_take(int64_t nres, int k, const int64_t *__restrict__ seg,
                                                  const double *__restrict__ d2, const int32_t *__restrict__ pos,
                                                  const int32_t *__restrict__ qres, const int32_t *__restrict__ perm,
                                                  int32_t *__restrict__ idx_out)
{
    const int64_t r = blockIdx.x;
    if (r >= nres) return;
    const int64_t b = seg[r], e = seg[r + 1], q = qres[r];
    // an element can only land among the first k if fewer than k precede it; everything at sorted index >= k + (length of
    // the tie run that straddles k) is out, so looking at sorted indices below the end of that run is enough
    for (int64_t i = b + threadIdx.x; i < e; i += blockDim.x) {
        const double d = d2[i];
        int64_t lo = i, hi = i + 1;
        while (lo > b && d2[lo - 1] == d) --lo;
        if (lo - b >= k) continue; // the whole run lies beyond the k-th place
        while (hi < e && d2[hi] == d) ++hi;
        int64_t rank = lo - b;
        const int p = pos[i];
        for (int64_t t = lo; t < hi; ++t) rank += pos[t] < p;
        if (rank < k) idx_out[q * (int64_t)k + rank] = perm[p];
    }
}

} // namespace


int sf_cloud_bbox(sf_ctx *ctx, sf_cloud *c, double lo[3], double hi[3]); // grid.hip

// One round of the large-k path for the selected queries (sel == NULL: all m): count the points within R, and for the
// queries that have at least k of them write, sort and take.  hstatus[q] = 0 answered / 1 retry with a larger R.
// Queries are processed in batches whose candidate lists stay within ~1.5e8 entries (3 GB of scratch).
namespace {
// [0]: queries not answered yet (status != 0), [1]: of those the crowded ones (status == 2)
__global__ __launch_bounds__(256) void k_knn_status_counts(const int32_t *__restrict__ status, int64_t m, unsigned *__restrict__ out)
{
    unsigned a = 0, b = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < m; i += (int64_t)gridDim.x * blockDim.x) {
        const int v = status[i];
        a += v != 0;
        b += v == 2;
    }
    for (int off = 32; off > 0; off >>= 1) { a += __shfl_xor(a, off); b += __shfl_xor(b, off); }
    if ((threadIdx.x & 63) == 0) {
        if (a) atomicAdd(out, a);
        if (b) atomicAdd(out + 1, b);
    }
}
struct knn_status_is {
    const int32_t *status;
    bool crowded_only;
    __host__ __device__ bool operator()(int32_t q) const { return crowded_only ? status[q] == 2 : status[q] != 0; }
};
} // namespace

static int knn_round_large(sf_ctx *ctx, sf_cloud *c, sf_nbrs *nb, const sf_grid_desc &g, const int32_t *sel, int64_t msel, int k,
                           double R2, std::vector<int32_t> &hstatus, const std::vector<int32_t> *sel_host)
{
    sf_pool_guard tmp(ctx);
    int32_t *dcount = nullptr;
    SF_CHECK(tmp.alloc(&dcount, (size_t)msel));
    SF_LAUNCH(ctx, "k2_knn", k_knn_sweep<false>, dim3(sf_xcd_grid(msel)), dim3(64), g, c->cell_start, c->xs, c->ys, c->zs, nb->qx,
              nb->qy, nb->qz, sel, msel, R2, dcount, (const int64_t *)nullptr, (double *)nullptr, (int32_t *)nullptr);
    std::vector<int32_t> hcount((size_t)msel);
    SF_HIP(hipMemcpyAsync(hcount.data(), dcount, (size_t)msel * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    SF_HIP(hipStreamSynchronize(ctx->stream));
    const int64_t budget = 150000000;
    int64_t s0 = 0;
    while (s0 < msel) {
        std::vector<int32_t> qres;
        std::vector<int64_t> seg(1, 0);
        int64_t s1 = s0;
        for (; s1 < msel; ++s1) {
            const int64_t q = sel_host ? (*sel_host)[(size_t)s1] : s1;
            if (hcount[(size_t)s1] < k) { hstatus[(size_t)q] = 1; continue; }
            if (!qres.empty() && seg.back() + hcount[(size_t)s1] > budget) break;
            hstatus[(size_t)q] = 0;
            qres.push_back((int32_t)q);
            seg.push_back(seg.back() + hcount[(size_t)s1]);
        }
        s0 = s1;
        const int64_t nres = (int64_t)qres.size(), total = seg.back();
        if (!nres) continue;
        if (total > 0xfffffff0LL) { sf_set_error("sf_knn_search: %lld candidates of one query exceed a sort", (long long)total); return SF_ERR_UNSUPPORTED; }
        sf_pool_guard bt(ctx);
        int32_t *dq = nullptr, *pin = nullptr, *pout = nullptr;
        int64_t *dseg = nullptr;
        double *din = nullptr, *dout = nullptr;
        SF_CHECK(bt.alloc(&dq, (size_t)nres));
        SF_CHECK(bt.alloc(&dseg, (size_t)nres + 1));
        SF_CHECK(bt.alloc(&din, (size_t)total));
        SF_CHECK(bt.alloc(&dout, (size_t)total));
        SF_CHECK(bt.alloc(&pin, (size_t)total));
        SF_CHECK(bt.alloc(&pout, (size_t)total));
        SF_HIP(hipMemcpyAsync(dq, qres.data(), (size_t)nres * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
        SF_HIP(hipMemcpyAsync(dseg, seg.data(), ((size_t)nres + 1) * sizeof(int64_t), hipMemcpyHostToDevice, ctx->stream));
        SF_LAUNCH(ctx, "k2_knn", k_knn_sweep<true>, dim3(sf_xcd_grid(nres)), dim3(64), g, c->cell_start, c->xs, c->ys, c->zs, nb->qx,
                  nb->qy, nb->qz, (const int32_t *)dq, nres, R2, (int32_t *)nullptr, (const int64_t *)dseg, din, pin);
        size_t tb = 0;
        SF_HIP(rocprim::segmented_radix_sort_pairs(nullptr, tb, din, dout, pin, pout, (unsigned)total, (unsigned)nres, dseg, dseg + 1, 0,
                                                   64, ctx->stream));
        char *stmp = nullptr;
        SF_CHECK(bt.alloc(&stmp, tb ? tb : 8));
        {
            sf_launch_timer t_(ctx, "k2_knn_sort");
            SF_HIP(rocprim::segmented_radix_sort_pairs(stmp, tb, din, dout, pin, pout, (unsigned)total, (unsigned)nres, dseg, dseg + 1, 0,
                                                       64, ctx->stream));
        }
        SF_LAUNCH(ctx, "k2_knn", k_knn_take, dim3((unsigned)nres), dim3(256), nres, k, (const int64_t *)dseg, (const double *)dout,
                  (const int32_t *)pout, (const int32_t *)dq, (const int32_t *)c->perm, nb->idx);
        SF_HIP(hipStreamSynchronize(ctx->stream)); // qres / seg are host buffers of the async copies
    }
    return SF_OK;
}

extern "C" sf_nbrs *sf_knn_search(sf_ctx *ctx, sf_cloud *c, const double *queries, int64_t m, int k, int flags)
{
    if (!ctx || !c || (!queries && m > 0) || m < 0 || m > 2147483000LL) {
        sf_set_error("sf_knn_search: bad arguments (m=%lld)", (long long)m);
        return nullptr;
    }
    if (k < 1 || k > c->n) { // sklearn: "k must be less than or equal to the number of training points"
        sf_set_error("sf_knn_search: k=%d must be in 1..%lld (the number of cloud points)", k, (long long)c->n);
        return nullptr;
    }
    const bool large = k > 1984; // beyond the LDS buffer of k_knn: count / fill / segmented sort (k_knn_sweep, k_knn_take)
    if (hipSetDevice(ctx->device) != hipSuccess) { sf_set_error("hipSetDevice failed"); return nullptr; }
    double lo[3], hi[3];
    if (sf_cloud_bbox(ctx, c, lo, hi) != SF_OK) return nullptr;
    double ext[3], emax = 0.0;
    for (int a = 0; a < 3; ++a) { ext[a] = hi[a] - lo[a]; emax = std::max(emax, ext[a]); }
    if (!(emax > 0.0)) emax = 1.0; // all points coincide: any radius works
    double vol = 1.0;
    for (int a = 0; a < 3; ++a) vol *= std::max(ext[a], 1e-3 * emax);
    // radius expected to hold ~2 k (k <= 64) / ~3.5 k points at the mean density of the bounding box (k_knn4 keeps the points within
    // R in an LDS list of SF_KNN_CAP entries: at most 0.6 of that on average)
    // (k <= 64, k_knn4: 2 k, at least k + 6 -- a retry costs little there: the unanswered queries are counted and selected on the
    // device, and a query crowded at the doubled radius sweeps again inside its bound; `tools/knn_target_ab.py`: 1M queries at
    // k = 30, search + normals, 2.88 ms wall at 3.5 k against 2.46 at 2 k on the uniform cloud, 3.58 against 3.46 on the surface.
    // The one-wave-per-query kernel of larger k keeps 3.5 k.)
    const double per_k = k <= 64 ? 2.0 : 3.5;
    const double within_target = k <= 64 ? std::min(std::max(per_k * (double)k, (double)k + 6.0), 0.6 * SF_KNN_CAP) : per_k * (double)k;
    double R = std::cbrt(within_target * vol / ((double)c->n * 4.18879020478639));
    const double diag = std::sqrt(ext[0] * ext[0] + ext[1] * ext[1] + ext[2] * ext[2]) + emax * 1e-6 + 1e-300;
    sf_nbrs *nb = new sf_nbrs();
    nb->m = m;
    nb->radius = 0.0;
    nb->self = false;
    auto fail = [&]() { sf_nbrs_free(ctx, nb); return (sf_nbrs *)nullptr; };
    if (sf_cloud_build_grid(ctx, c, R) != SF_OK) return fail();
    if (sf_k2_prepare_queries(ctx, c, nb, queries, flags) != SF_OK) return fail();
    // The bounding box's mean density is the density where the points are only for a cloud that fills its box: on a surface
    // scan a ball of this R holds ten times the 3.5 k points asked for, and the stencil sweep pays for all of them (10.6 ms
    // per 1M queries at k = 30 against 3.7 ms on a volume, round 4).  So the radius is checked against the data: the lists of a
    // sample of the queries are COUNTED at R (k_radius<0, true>, as run_search sizes its slots); while their mean is more
    // than 1.6 x the target, R shrinks as if the points lay on a surface (count ~ R^2: never shrinks too far for a volume,
    // where count ~ R^3) and the grid is rebuilt.  Only the speed depends on R: queries that see fewer than k points within
    // it are retried with a doubled radius below.
    if (m >= 2 * SF_K2_SAMPLE && !large) {
        const double target = within_target;
        for (int it = 0; it < 3; ++it) {
            sf_pool_guard stmp(ctx);
            int32_t *sel = nullptr, *cnt = nullptr;
            void *pin = nullptr;
            if (stmp.alloc(&sel, (size_t)SF_K2_SAMPLE) != SF_OK || stmp.alloc(&cnt, (size_t)SF_K2_SAMPLE) != SF_OK ||
                sf_ctx_pinned(ctx, &pin) != SF_OK)
                return fail();
            if (sf_k2_count_sample(ctx, c, nb, R * R, sel, cnt) != SF_OK) return fail();
            if (hipMemcpyAsync(pin, cnt, SF_K2_SAMPLE * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
                hipStreamSynchronize(ctx->stream) != hipSuccess) {
                sf_set_error("sf_knn_search: sample failed");
                return fail();
            }
            const int32_t *h = (const int32_t *)pin;
            double sum = 0.0;
            for (int i = 0; i < SF_K2_SAMPLE; ++i) sum += h[i];
            const double mean = sum / SF_K2_SAMPLE;
            if (!(mean > 1.6 * target)) break;
            R *= std::sqrt(target / mean) * 1.05;
            if (sf_cloud_build_grid(ctx, c, R) != SF_OK) return fail();
        }
    }
    if (sf_palloc(ctx, &nb->count, (size_t)(m + 1)) != SF_OK || sf_palloc(ctx, &nb->offset, (size_t)(m + 1)) != SF_OK ||
        sf_palloc(ctx, &nb->idx, (size_t)(m * k) + 4) != SF_OK)
        return fail();
    nb->total = m * (int64_t)k;
    nb->max_count = k;
    {
        sf_launch_timer t_(ctx, "k2_knn_fill_csr");
        hipLaunchKernelGGL(k_knn_fill_csr, dim3((unsigned)sf_div_up(m + 1, 256)), dim3(256), 0, ctx->stream, m, k, nb->count,
                           nb->offset);
    }
    if (!m) return nb;
    sf_pool_guard ktmp(ctx); // status / qsel go back to the pool on every exit of this function
    int32_t *status = nullptr, *qsel = nullptr, *qsel2 = nullptr;
    unsigned *dcounts = nullptr;
    if (ktmp.alloc(&status, (size_t)m) != SF_OK || ktmp.alloc(&qsel, (size_t)m) != SF_OK || ktmp.alloc(&qsel2, (size_t)m) != SF_OK ||
        ktmp.alloc(&dcounts, 2) != SF_OK)
        return fail();
    std::vector<int32_t> hstatus, pending; // (the count / fill / sort scheme of k > 1984 keeps its bookkeeping on the host)
    if (large) hstatus.assign((size_t)m, 0);
    int64_t msel = m;
    bool subset = false;
    void *pinv = nullptr;
    if (sf_ctx_pinned(ctx, &pinv) != SF_OK) return fail();
    unsigned *hcounts = (unsigned *)((char *)pinv + SF_PINNED_BYTES - 64);
    // how many queries are unanswered (status != 0) and how many of them are crowded (status == 2): 8 bytes read back per
    // round instead of the m status words and a host loop over them (round 5: 3 ms of host time per 1M queries and round)
    auto count_statuses = [&]() -> int {
        SF_HIP(hipMemsetAsync(dcounts, 0, 2 * sizeof(unsigned), ctx->stream));
        SF_LAUNCH(ctx, "k2_knn_status", k_knn_status_counts, dim3(256), dim3(256), (const int32_t *)status, m, dcounts);
        SF_HIP(hipMemcpyAsync(hcounts, dcounts, 2 * sizeof(unsigned), hipMemcpyDeviceToHost, ctx->stream));
        SF_HIP(hipStreamSynchronize(ctx->stream));
        return SF_OK;
    };
    auto select_status = [&](bool crowded_only, int32_t *out) -> int { // the queries with status == 2 / != 0, ascending
        rocprim::counting_iterator<int32_t> first(0);
        size_t tb = 0, *dnum = nullptr;
        sf_pool_guard st(ctx);
        SF_CHECK(st.alloc(&dnum, 1));
        const knn_status_is pred{status, crowded_only};
        SF_HIP(rocprim::select(nullptr, tb, first, out, dnum, (size_t)m, pred, ctx->stream));
        char *scratch = nullptr;
        SF_CHECK(st.alloc(&scratch, tb ? tb : 8));
        sf_launch_timer t_(ctx, "k2_knn_status");
        SF_HIP(rocprim::select(scratch, tb, first, out, dnum, (size_t)m, pred, ctx->stream));
        return SF_OK;
    };
    // Each round answers the queries that have k points within R; the others are retried with R doubled on a coarser
    // grid.  A query far outside the cloud's bounding box (KDTree.query answers those too: ICP feeds it scans that are
    // not yet aligned) stays pending until the grid has shrunk to ONE cell; that round drops the radius test, so every
    // cloud point is a candidate and, k being at most n, every remaining query is answered.
    bool resolved = false;
    int64_t n_pending = 0;
    for (int round = 0; round < 2200 && !resolved; ++round) { // R doubles: a double overflows long before 2200 rounds
        sf_grid_desc g = sf_make_grid_desc(c);
        const bool one_cell = g.dim[0] == g.xsub && g.dim[1] == 1 && g.dim[2] == 1;
        const double R2 = one_cell ? INFINITY : R * R;
        const dim3 grid(sf_xcd_grid(msel)), block(64);
        const int32_t *sel = subset ? qsel : nullptr;
        if (large) {
            if (knn_round_large(ctx, c, nb, g, sel, msel, k, R2, hstatus, subset ? &pending : nullptr) != SF_OK) {
                return fail();
            }
            pending.clear();
            for (int64_t i = 0; i < m; ++i)
                if (hstatus[(size_t)i] != 0) pending.push_back((int32_t)i);
            n_pending = (int64_t)pending.size();
        } else {
            if (k <= 64 && !one_cell) {
                // K2's mapping (k_knn4); the few queries with more than SF_KNN_CAP points within their bound go to k_knn at the same R
                const dim3 grid4(sf_xcd_grid(sf_div_up(msel, 4 * SF_K2_WPB))), block4(64 * SF_K2_WPB);
                {
                    sf_launch_timer t_(ctx, "k2_knn");
                    if (sel) hipLaunchKernelGGL((k_knn4<true>), grid4, block4, 0, ctx->stream, g, c->cell_start, c->xs, c->ys, c->zs, nb->qx, nb->qy, nb->qz, msel, R2, k, c->perm, nb->idx, status, sel);
                    else hipLaunchKernelGGL((k_knn4<false>), grid4, block4, 0, ctx->stream, g, c->cell_start, c->xs, c->ys, c->zs, nb->qx, nb->qy, nb->qz, msel, R2, k, c->perm, nb->idx, status, sel);
                }
                if (hipGetLastError() != hipSuccess || count_statuses() != SF_OK) { sf_set_error("sf_knn_search: launch failed"); return fail(); }
                if (hcounts[1]) {
                    const int64_t nc = hcounts[1];
                    if (select_status(true, qsel2) != SF_OK) return fail();
                    sf_launch_timer t_(ctx, "k2_knn_crowded");
                    hipLaunchKernelGGL(k_knn<2>, dim3(sf_xcd_grid(nc)), dim3(64), 0, ctx->stream, g, c->cell_start, c->xs, c->ys, c->zs, nb->qx, nb->qy,
                                       nb->qz, (const int32_t *)qsel2, nc, k, R2, c->perm, nb->idx, status);
                }
            } else {
                sf_launch_timer tm(ctx, "k2_knn");
#define SF_KNN_LAUNCH(EPL) hipLaunchKernelGGL(k_knn<EPL>, grid, block, 0, ctx->stream, g, c->cell_start, c->xs, c->ys, c->zs, nb->qx, nb->qy, nb->qz, sel, msel, k, R2, c->perm, nb->idx, status)
                if (k <= 64) SF_KNN_LAUNCH(2);         // buffer of 64 * EPL candidates >= k + 64
                else if (k <= 192) SF_KNN_LAUNCH(4);
                else if (k <= 448) SF_KNN_LAUNCH(8);
                else if (k <= 960) SF_KNN_LAUNCH(16);
                else SF_KNN_LAUNCH(32);
#undef SF_KNN_LAUNCH
            }
            if (hipGetLastError() != hipSuccess || count_statuses() != SF_OK) { sf_set_error("sf_knn_search: launch failed"); return fail(); }
            n_pending = hcounts[0];
        }
        if (!n_pending) { resolved = true; break; }
        if (one_cell) break; // cannot happen for k <= n (reported below)
        R = std::max(2.0 * R, std::min(diag, 1e300) * 1e-6); // sparse regions: retry only the unresolved queries on a coarser grid
        if (large) {
            if (sf_cloud_build_grid(ctx, c, R) != SF_OK ||
                hipMemcpyAsync(qsel, pending.data(), pending.size() * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream) != hipSuccess ||
                hipStreamSynchronize(ctx->stream) != hipSuccess)
                return fail();
        } else {
            if (select_status(false, qsel2) != SF_OK) return fail(); // (before the grid is rebuilt: stream order keeps it behind the round's kernels)
            std::swap(qsel, qsel2);
            if (sf_cloud_build_grid(ctx, c, R) != SF_OK) return fail();
        }
        msel = n_pending;
        subset = true;
    }
    if (!resolved) { // never return lists with unwritten rows
        sf_set_error("sf_knn_search: internal error, %lld queries unresolved", (long long)n_pending);
        return fail();
    }
    if (sf_cloud_ensure_inv_perm(ctx, c) != SF_OK) return fail();
    {
        sf_launch_timer t_(ctx, "k2_knn_to_positions");
        hipLaunchKernelGGL(k_knn_to_positions, dim3((unsigned)sf_div_up(nb->total, 256)), dim3(256), 0, ctx->stream, nb->total,
                           c->inv_perm, nb->idx);
    }
    sf_nbrs_stamp(nb, c); // (positions of the FINAL grid of the rounds above)
    return nb;
}

