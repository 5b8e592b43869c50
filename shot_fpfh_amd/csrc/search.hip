// search.hip -- K2: fixed-radius neighbour search on the uniform grid (count pass, scan, fill pass).
//
// Replaces: KDTree.query_radius at fpfh.py:28-30, shot_parallelization.py:167-169/220-231/283-285,
//           pca_based_descriptors.py:48 (sklearn, un-vendored; rule restated in SURVEY 8a-1):
//           j is a neighbour of q  <=>  ((dx*dx + dy*dy) + dz*dz) <= r*r   in float64, no FMA
//           (this TU is compiled with -ffp-contract=off), self-match included.
// Mapping: one 64-lane wave per query.  The 27-cell stencil is 9 contiguous runs of cell-sorted
// points (x is the fastest grid axis); a wave sweeps each run 64 candidates at a time with
// coalesced SoA loads, ballots the hits and compacts them with mbcnt prefix counts.
// Output: CSR in HBM -- int32 sorted positions, int64 offsets -- consumed by K3..K7.
// Roofline: HBM/L2 bound integer+compare work; algorithmic bytes = 24 B per query in + 4 B per pair out.
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_reduce.hpp>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/device/device_select.hpp>
#include <rocprim/iterator/counting_iterator.hpp>
#include <rocprim/iterator/transform_iterator.hpp>

#include <algorithm>
#include <cmath>
#include <numeric>

#include "search_util.h"

sf_grid_desc sf_make_grid_desc(const sf_cloud *c)
{
    sf_grid_desc g;
    for (int a = 0; a < 3; ++a) {
        g.lo[a] = c->lo[a];
        g.dim[a] = c->dim[a];
    }
    g.inv_cell = c->inv_cell;
    g.inv_cell_x = c->inv_cell * c->xsub; // (xsub is a power of two: exact)
    g.cell = c->cell;
    g.xsub = c->xsub;
    return g;
}

namespace {

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

// SEL: the launch serves the `m` queries named by qsel (processing slots) instead of queries 0 .. m - 1 -- the sample a
// search sizes its slots from (MODE 0: counts go to the COMPACT array count[0 .. m)), and the queries whose lists overflowed
// their slot (MODE 1: fill at offset[q], which by then points into the overflow area).
template <int MODE, bool SEL>
__global__ __launch_bounds__(64 * SF_K2_WPB) void k_radius(sf_grid_desc g, const int32_t *__restrict__ cell_start,
                                                const double *__restrict__ xs, const double *__restrict__ ys,
                                                const double *__restrict__ zs, const double *__restrict__ qx,
                                                const double *__restrict__ qy, const double *__restrict__ qz,
                                                int64_t m, double r2, int cap, int32_t *__restrict__ count,
                                                int64_t *__restrict__ offset, int32_t *__restrict__ idx,
                                                const int32_t *__restrict__ qsel)
{
    // A wave serves FOUR consecutive queries.  The run tables of all four are built at once, one query per 16-lane
    // DPP row (lanes 0..8 of a row describe its query's nine runs): the ~150 instructions of that set-up -- band gaps,
    // square root, cell look-ups, scan -- are executed once per four queries instead of once per query, where they were
    // 40 % of the kernel's (issue-bound) vector instructions.  The candidate sweeps then run query after query with the
    // whole wave, as before.
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
    // The stencil is up to 9 runs of consecutive positions, one per (cz, cy) row of cells.  The runs are laid end to
    // end in units of candidate PAIRS (a pair = one even-aligned 16-byte load) and every step of a sweep takes the next
    // 64 pairs, whichever runs they fall in.  Everything a lane needs to find its pair is vector work -- a DPP scan
    // gives the runs' first pair slots, the table goes to LDS -- because the scalar unit is shared by the whole CU.
    __shared__ int4 runs[SF_K2_WPB][4][12];
#if SF_K2_STAGE
    __shared__ int stages[MODE != 0 ? SF_K2_WPB : 1][256];
#endif
    int4(*const tabs)[12] = runs[threadIdx.x >> 6];
    int first_slot = 0; // lane r < 9 of a row: first pair slot of run r; lane 9: the total
    {
        const int r = sl < 9 ? sl : 8;
        const int cz = z0 + r / 3, cy = y0 + r % 3;
        bool ok = sl < 9 && rw < nq && cz <= z1 && cy <= y1;
        const int64_t row = ((int64_t)(ok ? cz : z0) * g.dim[1] + (ok ? cy : y0)) * g.dim[0];
        // The row (cy, cz) of cells is the band [lo + c edge, lo + (c + 1) edge) in y and in z.  A point of it within r
        // of the query is at least (dy, dz) away in those two axes -- the gaps between the query and the bands -- so
        // its x lies within w = sqrt(r^2 - dy^2 - dz^2) of the query's: only the FINE x cells (xsub per edge) that
        // [px - w, px + w] touches are swept, 12-14 edge-lengths of cells per query instead of 27.  The cell of a
        // coordinate is a monotone function of it, so every point with px - w <= x <= px + w lies in a cell between
        // the cells of the two ends; the gaps shrink and w grows by a slack that is far above any rounding involved (next
        // paragraph), so the sweep can only be wider than necessary, never narrower.
        // All of it in GRID-RELATIVE coordinates (p - lo, what the cell assignment itself uses): the rounding of a relative
        // coordinate, of a band edge c * cell and of fl(1 / cell) is a few ulps of the grid's EXTENT, whereas band edges
        // and px +- w formed in absolute coordinates would round by ulps of |lo| and |p| -- for a cloud in UTM-like
        // coordinates (1e6 .. 1e7 with a cell of 1e-2) far more than any fixed fraction of the cell.  `slack` covers both: 1e-9
        // of the cell plus 1e-15 of the relative coordinates involved.
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
        const int base = s & ~1; // pairs start at an EVEN position: every 16-byte load is naturally aligned
        const int npairs = (e - base + 1) >> 1;
        // inclusive scan of npairs inside the 16-lane row (row_shr DPP steps), then exclusive = inclusive - own
        int inc = npairs;
        inc += __builtin_amdgcn_update_dpp(0, inc, 0x111, 0xf, 0xf, false); // row_shr:1
        inc += __builtin_amdgcn_update_dpp(0, inc, 0x112, 0xf, 0xf, false); // row_shr:2
        inc += __builtin_amdgcn_update_dpp(0, inc, 0x114, 0xf, 0xf, false); // row_shr:4
        inc += __builtin_amdgcn_update_dpp(0, inc, 0x118, 0xf, 0xf, false); // row_shr:8
        first_slot = inc - npairs;
        if (sl < 12) tabs[rw][sl] = make_int4(base - 2 * first_slot, s, e, first_slot); // j = .x + 2 * slot; .w: first slot
    }
    __builtin_amdgcn_wave_barrier(); // the tables are written and read by this wave only
    for (int qi = 0; qi < nq; ++qi) {
        const int64_t q = SEL ? (int64_t)sf_uniform(__shfl((int)qm, 16 * qi)) : q0 + qi;
        const int4 *const tab = tabs[qi];
        const double px = __shfl(pxv, 16 * qi), py = __shfl(pyv, 16 * qi), pz = __shfl(pzv, 16 * qi);
        // two of the boundaries between runs and the number of pair slots, as scalars
        const int b4 = __shfl(first_slot, 16 * qi + 4), b8 = __shfl(first_slot, 16 * qi + 8);
#ifdef SF_K2_ABLATE // TIMING ONLY (tools/ab_libs.sh): sweep this percentage of the candidate slots -- what a finer z split could save at best
        const int nslots = sf_uniform(__shfl(first_slot, 16 * qi + 9)) * SF_K2_ABLATE / 100;
#else
        const int nslots = sf_uniform(__shfl(first_slot, 16 * qi + 9)); // lane 9 has npairs = 0: its exclusive sum is the total
#endif
        int total = 0;
        const int64_t out = MODE == 1 ? offset[q] : q * (int64_t)cap;
        const int room = MODE == 2 ? cap : 0x7fffffff;
#if SF_K2_STAGE
        // Hits go to a 256-entry ring in LDS first and leave for HBM 64 list positions at a time: a sweep step finds ~14 hits
        // of its 128 candidates, and 14 x 4 B stored past the L2 (`nt`) is a partial line each time -- the counters showed
        // 2.2 x the lists' bytes written.  Whole, aligned 256-byte runs instead (slots start on 128-byte boundaries).
        int *const stage = stages[threadIdx.x >> 6];
        int flushed = 0; // list positions [0, flushed) are in HBM; a multiple of 64
#endif
        for (int f0 = 0; f0 < nslots; f0 += 64) {
            const int f = f0 + lane;
            // run of slot f = last run whose first slot is <= f: a three-level binary search, the first level against a
            // scalar, the other two against the first-slot column of the table in LDS (an LDS read costs the vector
            // pipe one instruction; eight compare-and-add pairs cost it sixteen and more)
            int r = f >= b4 ? 4 : 0;
            r += f >= tab[r + 2].w ? 2 : 0;
            r += f >= tab[r + 1].w ? 1 : 0;
            r = f >= b8 ? 8 : r;
            const int4 t = tab[r];
            const bool live = f < nslots;
            const int j = live ? t.x + 2 * f : 0; // idle lanes of the last step load pair 0 (always there)
            const bool in0 = live & (j >= t.y), in1 = live & (j + 1 < t.z);
            const double2 X = *reinterpret_cast<const double2 *>(xs + j);
            const double2 Y = *reinterpret_cast<const double2 *>(ys + j);
            const double2 Z = *reinterpret_cast<const double2 *>(zs + j);
            const double dxa = X.x - px, dya = Y.x - py, dza = Z.x - pz;
            const double dxb = X.y - px, dyb = Y.y - py, dzb = Z.y - pz;
            // both distances are evaluated unconditionally (bitwise &): a short-circuit would let the compiler
            // sink half of each 16-byte load into a branch and split it into two 8-byte loads
            const double d2a = (dxa * dxa + dya * dya) + dza * dza, d2b = (dxb * dxb + dyb * dyb) + dzb * dzb;
            const bool hit0 = in0 & (d2a <= r2);
            const bool hit1 = in1 & (d2b <= r2);
            const unsigned long long m0 = __ballot(hit0), m1 = __ballot(hit1);
            if (MODE != 0) {
                const int pos = total + sf_prefix_count(m0) + sf_prefix_count(m1);
                const int pos1 = pos + (hit0 ? 1 : 0);
#if SF_K2_STAGE
                if (hit0 && pos < room) stage[pos & 255] = j;
                if (hit1 && pos1 < room) stage[pos1 & 255] = j + 1;
#else
                if (hit0 && pos < room) SF_LIST_STORE(idx + out + pos, j);
                if (hit1 && pos1 < room) SF_LIST_STORE(idx + out + pos1, j + 1);
#endif
            }
            total += __popcll(m0) + __popcll(m1);
#if SF_K2_STAGE
            if (MODE != 0) { // (at most 63 + 128 positions are pending here: the ring of 256 never wraps onto them)
                const int have = total < room ? total : room;
                while (have - flushed >= 64) {
                    SF_LIST_STORE(idx + out + flushed + lane, stage[(flushed + lane) & 255]);
                    flushed += 64;
                }
            }
#endif
        }
#if SF_K2_STAGE
        if (MODE != 0) {
            const int have = total < room ? total : room;
            if (flushed + lane < have) SF_LIST_STORE(idx + out + flushed + lane, stage[(flushed + lane) & 255]);
        }
#endif
        if (lane == 0) {
            if (MODE != 1) {
                count[SEL ? q0 + qi : q] = total;
                if (!SEL && q == m - 1) count[m] = 0; // (the one element past the last query: no memset of the arrays needed)
            }
            if (MODE == 2) {
                offset[q] = out;
                if (q == m - 1) offset[m] = out + cap;
            }
        }
    }
}

// --------------------------------------------------------------------------------------------------
// K2 + K3 in one sweep: compute_normals(radius=...) needs the covariance of every neighbourhood, not the neighbourhood --
// the candidate sweep of k_radius keeps the hits' offsets from the query in LDS (in the very order k_radius would have
// written their indices), and the barycentre / centred second moments are formed from there exactly as k_pca_cov forms them
// from the materialised list (same terms, same lane assignment t -> lane t % 64, same reductions): the six numbers per query
// are bit-identical to sf_radius_search + sf_normals, and neither the 4 B per pair of the lists nor the 24 B per pair of
// their gather ever touch HBM.  Lists of at most SF_K2C_LIST points sit in LDS whole (4.6 KB per wave); a longer one (wave-uniform,
// per query) is swept twice more through the same LDS as a ring -- whenever 64 consecutive list positions are complete the
// lanes consume them -- once for the barycentre, once for the moments: the streaming form of k_pca_cov, again bit for bit.
// Set-up and sweep are k_radius's (four queries per wave, run tables per 16-lane row, candidate pairs).
// --------------------------------------------------------------------------------------------------
#ifndef SF_K2C_WPB
#define SF_K2C_WPB 4
#endif
#ifndef SF_K2C_LIST
#define SF_K2C_LIST 192 // entries of the LDS list / ring: a multiple of 64, at least 192 (a sweep step adds up to 128 hits to < 64 unconsumed ones)
#endif
struct sf_cov_list { double x[SF_K2C_LIST], y[SF_K2C_LIST], z[SF_K2C_LIST]; };

__global__ __launch_bounds__(64 * SF_K2C_WPB) __attribute__((amdgpu_waves_per_eu(7))) void k_radius_cov(sf_grid_desc g, const int32_t *__restrict__ cell_start,
                                                const double *__restrict__ xs, const double *__restrict__ ys,
                                                const double *__restrict__ zs, const double *__restrict__ qx,
                                                const double *__restrict__ qy, const double *__restrict__ qz,
                                                int64_t m, double r2, double *__restrict__ cov, double *__restrict__ bary,
                                                int32_t *__restrict__ count)
{
    const int lane = threadIdx.x & 63, sl = lane & 15, rw = lane >> 4;
    const int64_t q0 = sf_uniform64((sf_xcd_block() * SF_K2C_WPB + (threadIdx.x >> 6)) * 4);
    if (q0 >= m) return;
    const int nq = (int)(m - q0 < 4 ? m - q0 : 4);
    const int64_t qm = q0 + (rw < nq ? rw : 0);
    const double pxv = qx[qm], pyv = qy[qm], pzv = qz[qm]; // this row's query
    int y0, y1, z0, z1;
    stencil_bounds(pyv, g.lo[1], g.inv_cell, g.dim[1], y0, y1);
    stencil_bounds(pzv, g.lo[2], g.inv_cell, g.dim[2], z0, z1);
    __shared__ int4 runs[SF_K2C_WPB][4][12];
    __shared__ sf_cov_list lists[SF_K2C_WPB];
    int4(*const tabs)[12] = runs[threadIdx.x >> 6];
    sf_cov_list &L = lists[threadIdx.x >> 6];
    int first_slot = 0;
    { // (the run tables: see k_radius)
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
#define SF_K2C_LDS_SYNC()                                                                                            \
    do {                                                                                                            \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");                                                      \
        __builtin_amdgcn_wave_barrier();                                                                            \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");                                                      \
    } while (0)
    for (int qi = 0; qi < nq; ++qi) {
        const int64_t q = q0 + qi;
        const int4 *const tab = tabs[qi];
        const double px = __shfl(pxv, 16 * qi), py = __shfl(pyv, 16 * qi), pz = __shfl(pzv, 16 * qi);
        const int b4 = __shfl(first_slot, 16 * qi + 4), b8 = __shfl(first_slot, 16 * qi + 8);
        const int nslots = sf_uniform(__shfl(first_slot, 16 * qi + 9));
        // one sweep over the candidates; hits land in L at (list position, modulo the list's size in ring mode) when that position is below `keep`;
        // after_step(total so far) runs once per 128 candidates
        auto sweep = [&](int keep, bool wrap, auto &&after_step) -> int { // wrap: L is a ring (positions modulo its size)
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
                const bool hit0 = in0 & (d2a <= r2);
                const bool hit1 = in1 & (d2b <= r2);
                const unsigned long long m0 = __ballot(hit0), m1 = __ballot(hit1);
                const int pos = total + sf_prefix_count(m0) + sf_prefix_count(m1);
                const int pos1 = pos + (hit0 ? 1 : 0);
                const int s0 = wrap ? pos % SF_K2C_LIST : pos, s1 = wrap ? pos1 % SF_K2C_LIST : pos1;
                if (hit0 && pos < keep) { L.x[s0] = dxa; L.y[s0] = dya; L.z[s0] = dza; }
                if (hit1 && pos1 < keep) { L.x[s1] = dxb; L.y[s1] = dyb; L.z[s1] = dzb; }
                total += __popcll(m0) + __popcll(m1);
                after_step(total);
            }
            return total;
        };
        SF_K2C_LDS_SYNC(); // (the previous query's reads of L are done)
        const int k = sweep(SF_K2C_LIST, false, [](int) {});
        SF_K2C_LDS_SYNC();
        const double kk = (double)k, ik = 1.0 / kk; // (one division per query, as k_pca_cov)
        double sx = 0.0, sy = 0.0, sz = 0.0;
        double part[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        double mx, my, mz;
        if (k <= SF_K2C_LIST) {
            for (int t = lane; t < k; t += 64) { sx += L.x[t]; sy += L.y[t]; sz += L.z[t]; }
            {
                const double bs[4] = {sx, sy, sz, 0.0};
                const double bt = sf_wave_sum4(bs); // (as k_pca_cov)
                mx = sf_read_lane(bt, 0) * ik; my = sf_read_lane(bt, 16) * ik; mz = sf_read_lane(bt, 32) * ik;
            }
            for (int t = lane; t < k; t += 64) {
                const double ax = L.x[t] - mx, ay = L.y[t] - my, az = L.z[t] - mz;
                part[0] += ax * ax;
                part[1] += ay * ax;
                part[2] += az * ax;
                part[3] += ay * ay;
                part[4] += az * ay;
                part[5] += az * az;
            }
        } else { // (a long list: two more sweeps, the hits consumed 64 list positions at a time)
            int done = 0;
            auto bary_step = [&](int total) {
                if (done + 64 <= total) {
                    SF_K2C_LDS_SYNC();
                    while (done + 64 <= total) {
                        const int t = (done + lane) % SF_K2C_LIST;
                        sx += L.x[t]; sy += L.y[t]; sz += L.z[t];
                        done += 64;
                    }
                    SF_K2C_LDS_SYNC();
                }
            };
            sweep(0x7fffffff, true, bary_step);
            SF_K2C_LDS_SYNC();
            if (done + lane < k) { const int t = (done + lane) % SF_K2C_LIST; sx += L.x[t]; sy += L.y[t]; sz += L.z[t]; }
            {
                const double bs[4] = {sx, sy, sz, 0.0};
                const double bt = sf_wave_sum4(bs); // (as k_pca_cov)
                mx = sf_read_lane(bt, 0) * ik; my = sf_read_lane(bt, 16) * ik; mz = sf_read_lane(bt, 32) * ik;
            }
            auto add_moments = [&](int t) {
                const double ax = L.x[t] - mx, ay = L.y[t] - my, az = L.z[t] - mz;
                part[0] += ax * ax;
                part[1] += ay * ax;
                part[2] += az * ax;
                part[3] += ay * ay;
                part[4] += az * ay;
                part[5] += az * az;
            };
            done = 0;
            auto mom_step = [&](int total) {
                if (done + 64 <= total) {
                    SF_K2C_LDS_SYNC();
                    while (done + 64 <= total) {
                        add_moments((done + lane) % SF_K2C_LIST);
                        done += 64;
                    }
                    SF_K2C_LDS_SYNC();
                }
            };
            SF_K2C_LDS_SYNC();
            sweep(0x7fffffff, true, mom_step);
            SF_K2C_LDS_SYNC();
            if (done + lane < k) add_moments((done + lane) % SF_K2C_LIST);
        }
        const double tot = sf_wave_sum8(part); // lanes 8 i .. 8 i + 7 hold the sum of part[i]
        const int e = lane >> 3;
        if ((lane & 7) == 0 && e < 6) cov[6 * q + e] = tot * ik; // c11 c21 c31 c22 c32 c33
        if (bary && lane == 0) { bary[3 * q] = mx; bary[3 * q + 1] = my; bary[3 * q + 2] = mz; }
        if (count && lane == 0) count[q] = k;
    }
#undef SF_K2C_LDS_SYNC
}

__global__ void k_query_cells(const double *__restrict__ q, int64_t m, sf_grid_desc g, int32_t *__restrict__ cid,
                              int32_t *__restrict__ val)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    int cx = sf_cell_coord(q[3 * i + 0], g.lo[0], g.inv_cell_x, g.dim[0]);
    int cy = sf_cell_coord(q[3 * i + 1], g.lo[1], g.inv_cell, g.dim[1]);
    int cz = sf_cell_coord(q[3 * i + 2], g.lo[2], g.inv_cell, g.dim[2]);
    cid[i] = (cz * g.dim[1] + cy) * g.dim[0] + cx;
    val[i] = (int32_t)i;
}

__global__ void k_gather_queries(const double *__restrict__ q, const int32_t *__restrict__ qrow, int64_t m,
                                 double *__restrict__ qx, double *__restrict__ qy, double *__restrict__ qz)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    int64_t o = qrow[i];
    qx[i] = q[3 * o + 0];
    qy[i] = q[3 * o + 1];
    qz[i] = q[3 * o + 2];
}

// Export helper: copy every list from its (slot or CSR) position to the exact CSR position eoff[q], mapping
// cell-sorted positions to the caller's point numbering, optionally with sqrt(d2) of each pair
// (KDTree.query_radius(..., return_distance=True)).
__global__ __launch_bounds__(256) void k_export_lists(const double *__restrict__ xs, const double *__restrict__ ys,
                                                      const double *__restrict__ zs, const double *__restrict__ qx,
                                                      const double *__restrict__ qy, const double *__restrict__ qz,
                                                      const int64_t *__restrict__ offset, const int32_t *__restrict__ cnt,
                                                      const int32_t *__restrict__ idx, const int64_t *__restrict__ eoff,
                                                      const int32_t *__restrict__ perm, int64_t m,
                                                      int32_t *__restrict__ out_idx, double *__restrict__ out_dist)
{
    const int lane = threadIdx.x & 63;
    const int64_t q = sf_uniform64((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6));
    if (q >= m) return;
    const int64_t s = offset[q], d = eoff[q];
    const int k = cnt[q];
    const double px = qx[q], py = qy[q], pz = qz[q];
    for (int t = lane; t < k; t += 64) {
        const int j = idx[s + t];
        out_idx[d + t] = perm[j];
        if (out_dist) {
            const double dx = xs[j] - px, dy = ys[j] - py, dz = zs[j] - pz;
            out_dist[d + t] = sqrt((dx * dx + dy * dy) + dz * dz);
        }
    }
}


} // namespace


// ---- list statistics -------------------------------------------------------------------------------------------------
// Sum, maximum and a five-class histogram of the per-query counts (<= 64, <= 128, <= 192, <= 255, longer), plus the number
// and the total length of the lists longer than `cap` (the slot capacity of the single sweep; 0x7fffffff: none can be):
// per-block partials in ONE pass, folded on the host after ONE small read-back into page-locked memory (library
// reductions + pageable copies cost several idle gaps per step).
#define SF_STATS_BLOCKS 128
struct sf_stats_parts { // device / pinned image, one entry per block
    long long sum[SF_STATS_BLOCKS], ovf_sum[SF_STATS_BLOCKS];
    int mx[SF_STATS_BLOCKS], gmx[SF_STATS_BLOCKS], n_ovf[SF_STATS_BLOCKS], hist[5][SF_STATS_BLOCKS];
};
static_assert(sizeof(sf_stats_parts) <= SF_PINNED_BYTES, "statistics exceed the pinned words");

__global__ __launch_bounds__(256) void k_count_stats(const int32_t *__restrict__ count, int64_t m, int cap, sf_stats_parts *__restrict__ out)
{
    long long sacc = 0, oacc = 0;
    int macc = 0, nov = 0, h0 = 0, h1 = 0, h2 = 0, h3 = 0, h4 = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < m; i += (int64_t)gridDim.x * blockDim.x) {
        const int v = count[i];
        sacc += v;
        macc = max(macc, v);
        if (v > cap) { ++nov; oacc += v; }
        h0 += v <= 64;
        h1 += v > 64 && v <= 128;
        h2 += v > 128 && v <= 192;
        h3 += v > 192 && v <= 255;
        h4 += v > 255;
    }
    for (int off = 32; off > 0; off >>= 1) {
        sacc += __shfl_xor(sacc, off);
        oacc += __shfl_xor(oacc, off);
        macc = max(macc, __shfl_xor(macc, off));
        nov += __shfl_xor(nov, off);
        h0 += __shfl_xor(h0, off); h1 += __shfl_xor(h1, off); h2 += __shfl_xor(h2, off);
        h3 += __shfl_xor(h3, off); h4 += __shfl_xor(h4, off);
    }
    __shared__ long long ss[4], so[4];
    __shared__ int sm[4], sn[4], sh[5][4];
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
        ss[w] = sacc; so[w] = oacc; sm[w] = macc; sn[w] = nov;
        sh[0][w] = h0; sh[1][w] = h1; sh[2][w] = h2; sh[3][w] = h3; sh[4][w] = h4;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const int b = blockIdx.x;
        out->sum[b] = ss[0] + ss[1] + ss[2] + ss[3];
        out->ovf_sum[b] = so[0] + so[1] + so[2] + so[3];
        out->mx[b] = max(max(sm[0], sm[1]), max(sm[2], sm[3]));
        out->n_ovf[b] = sn[0] + sn[1] + sn[2] + sn[3];
        for (int c = 0; c < 5; ++c) out->hist[c][b] = sh[c][0] + sh[c][1] + sh[c][2] + sh[c][3];
    }
}

// fills nb->total / max_count / max_count_all / hist / n_overflow; *ovf_total = summed length of the lists longer than cap
static int count_stats(sf_ctx *ctx, sf_nbrs *nb, int cap, int64_t *ovf_total)
{
    const int64_t m = nb->m;
    sf_pool_guard tmp(ctx);
    sf_stats_parts *dev = nullptr;
    void *pin = nullptr;
    SF_CHECK(sf_ctx_pinned(ctx, &pin));
    // sf_comm_collective_stats: the longest list over every rank (the ranks size their SPFH tables by it)
    const bool fold = ctx->collective_stats && ctx->comm;
    if (!fold) {
        // the partial sums go straight into the page-locked block the host reads them from (it is mapped into the device's
        // address space; 28 KB over the host link from 256 workgroups): no copy operation behind the kernel, one dependent
        // GPU operation less in front of the step's one read-back
        SF_LAUNCH(ctx, "k2_reduce", k_count_stats, dim3(SF_STATS_BLOCKS), dim3(256), (const int32_t *)nb->count, m, cap, (sf_stats_parts *)pin);
    } else {
        SF_CHECK(tmp.alloc(&dev, 1));
        SF_LAUNCH(ctx, "k2_reduce", k_count_stats, dim3(SF_STATS_BLOCKS), dim3(256), (const int32_t *)nb->count, m, cap, dev);
        SF_CHECK(sf_comm_allreduce_max_i32(ctx, dev->mx, dev->gmx, SF_STATS_BLOCKS));
        SF_HIP(hipMemcpyAsync(pin, dev, sizeof(sf_stats_parts), hipMemcpyDeviceToHost, ctx->stream));
    }
    SF_HIP(hipStreamSynchronize(ctx->stream));
    const sf_stats_parts *h = (const sf_stats_parts *)pin;
    int64_t t = 0, ot = 0, nov = 0;
    int32_t mm = 0, ga = 0;
    for (int c = 0; c < 5; ++c) nb->hist[c] = 0;
    for (int b = 0; b < SF_STATS_BLOCKS; ++b) {
        t += h->sum[b];
        ot += h->ovf_sum[b];
        nov += h->n_ovf[b];
        mm = std::max<int32_t>(mm, h->mx[b]);
        if (fold) ga = std::max<int32_t>(ga, h->gmx[b]);
        for (int c = 0; c < 5; ++c) nb->hist[c] += h->hist[c][b];
    }
    nb->total = t;
    nb->max_count = mm;
    nb->max_count_all = fold ? std::max(ga, mm) : mm;
    nb->n_overflow = nov;
    if (ovf_total) *ovf_total = ot;
    return SF_OK;
}

// ---- selections of queries by list length ------------------------------------------------------------------------------
struct count_longer {
    const int32_t *count;
    int limit;
    __host__ __device__ bool operator()(int32_t q) const { return count[q] > limit; }
};
struct count_between { // lo < count <= hi
    const int32_t *count;
    int lo, hi;
    __host__ __device__ bool operator()(int32_t q) const { const int c = count[q]; return c > lo && c <= hi; }
};
struct count_of {
    const int32_t *count;
    __host__ __device__ int64_t operator()(int32_t q) const { return (int64_t)count[q]; }
};

// the processing slots whose list length satisfies `pred`, ascending (`expect` of them: known from the statistics)
template <typename Pred>
static int select_slots(sf_ctx *ctx, int64_t m, Pred pred, int64_t expect, int32_t **out)
{
    sf_pool_guard tmp(ctx);
    size_t *dnum = nullptr;
    SF_CHECK(tmp.alloc(&dnum, 1));
    SF_CHECK(sf_palloc(ctx, out, (size_t)expect + 1));
    rocprim::counting_iterator<int32_t> first(0);
    size_t tb = 0;
    SF_HIP(rocprim::select(nullptr, tb, first, *out, dnum, (size_t)m, pred, ctx->stream));
    char *scratch = nullptr;
    SF_CHECK(tmp.alloc(&scratch, tb ? tb : 8));
    {
        sf_launch_timer t_(ctx, "k2_select");
        SF_HIP(rocprim::select(scratch, tb, first, *out, dnum, (size_t)m, pred, ctx->stream));
    }
    return SF_OK;
}
static int select_longer(sf_ctx *ctx, const int32_t *count, int64_t m, int limit, int64_t expect, int32_t **out)
{
    return select_slots(ctx, m, count_longer{count, limit}, expect, out);
}

__global__ void k_iota_stride(int32_t *__restrict__ out, int64_t n, int64_t stride)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (int32_t)(i * stride);
}

__global__ void k_patch_offsets(const int32_t *__restrict__ sel, const int64_t *__restrict__ off, int64_t n, int64_t delta,
                                int64_t *__restrict__ offset)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) offset[sel[i]] = delta + off[i];
}

// Which instantiation the list-driven kernels take for these lists, and which queries a second launch serves (see sf_nbrs).
static int plan_dispatch(sf_ctx *ctx, sf_cloud *c, sf_nbrs *nb)
{
    const int64_t m = nb->m;
    if (m >= SF_K2_SAMPLE) {
        if (c->list_stats.size() > 64) c->list_stats.clear();
        c->list_stats[{nb->radius, nb->self}] = {(double)nb->total / (double)m, nb->max_count};
    }
    // WHICH form serves a keypoint depends on its own list alone -- longer than 255 points: the second launch -- so that a
    // keypoint gets the same bits whatever else is in the launch (another block of a sharded job, a slice, a subset); the
    // chunk count of the main launch's instantiation only has to cover the longest list it serves (the instantiations of
    // one form agree bit for bit: chunks past a list's end contribute nothing).
    const int64_t longest_main = std::min<int64_t>(nb->max_count > 0 ? nb->max_count : 1, 255);
    const int need = (int)sf_div_up(longest_main, 64); // chunks that cover every list of at most 255 points
    // ... but when only a small share of those lists needs the last chunks, the bulk runs the leaner instantiation and that
    // share gets a launch of its own in the 4-chunk instantiation of the same form (bit-identical rows either way): the
    // smallest chunk count that leaves at most 2 % of the (<= 255-point) lists to it.  (At C3 8 % of the lists need the third
    // chunk: a split there costs more than it gains -- the second launch's scattered keypoints run 1.4x slower each, DESIGN
    // fact 19.)
    int chunks = need;
    {
        const int64_t fit = nb->hist[0] + nb->hist[1] + nb->hist[2] + nb->hist[3];
        int64_t within = 0;
        for (int c = 0; c < need; ++c) {
            within += nb->hist[c];
            // (the FOUR-chunk instantiations run at five waves per SIMD instead of seven: when up to a fifth of the lists
            // need the fourth chunk it still pays to keep them out of the bulk -- clustered cloud, 13 % of them: 5.08 -> 4.93 ms;
            // uniform cloud at 159 / 174 / 181 neighbours, 3 / 17 / 30 %: 5.04 -> 4.92, 5.17 -> 5.10, 5.52 -> 5.56)
            const int64_t share = (c + 1 == 3 && need == 4) ? 5 : 50;
            if ((fit - within) * share <= fit) { chunks = c + 1; break; }
        }
        if (getenv("SF_NO_MID_LAUNCH")) chunks = need;
    }
    nb->planned = true;
    nb->main_chunks = chunks;
    nb->n_mid = 0;
    for (int c = chunks; c < 4; ++c) nb->n_mid += nb->hist[c];
    if (chunks == need) nb->n_mid = 0;
    nb->tail_limit = 255;
    nb->n_tail = nb->hist[4];
    if (nb->max_count <= nb->tail_limit) { nb->tail_limit = 0x7fffffff; nb->n_tail = 0; }
    if (nb->n_mid) SF_CHECK(select_slots(ctx, m, count_between{nb->count, 64 * chunks, 255}, nb->n_mid, &nb->mid_sel));
    if (nb->n_tail) SF_CHECK(select_longer(ctx, nb->count, m, nb->tail_limit, nb->n_tail, &nb->tail_sel));
    return SF_OK;
}

// Slot capacity of the single sweep from a SAMPLE of the search itself: the lists of every (m / 2048)-th query are counted
// first (one small launch, 8 KB read back) and the slots sized by their mean and maximum.  (Until round 3 the capacity came
// from the mean density of the bounding box: right for a cloud that fills its box, an order of magnitude low for a surface
// scan -- every list overflowed and the "single" sweep was followed by a scan and a second sweep.)
// 2.25 x the mean + 32 (what a Poisson-like cloud never exceeds: C3's lists reach 1.5 x their mean), at least 1.15 x the
// longest list seen; a multiple of 32.  Lists beyond it are re-done on their own (run_search).
static int64_t capacity_from(const sf_cloud *c, double mean, int64_t mx)
{
    int64_t v = (int64_t)(std::max(mean * 2.25, 1.15 * (double)mx)) + 32;
    return std::min<int64_t>(((v + 31) / 32) * 32, std::max<int64_t>(((c->n + 31) / 32) * 32, 32));
}

static int sample_capacity(sf_ctx *ctx, sf_cloud *c, sf_nbrs *nb, const sf_grid_desc &g, double r2, int64_t *cap)
{
    const int64_t m = nb->m, stride = m / SF_K2_SAMPLE;
    sf_pool_guard tmp(ctx);
    int32_t *sel = nullptr, *cnt = nullptr;
    void *pin = nullptr;
    SF_CHECK(tmp.alloc(&sel, (size_t)SF_K2_SAMPLE));
    SF_CHECK(tmp.alloc(&cnt, (size_t)SF_K2_SAMPLE));
    SF_CHECK(sf_ctx_pinned(ctx, &pin));
    static_assert(SF_K2_SAMPLE * sizeof(int32_t) <= SF_PINNED_BYTES, "sample exceeds the pinned words");
    SF_LAUNCH(ctx, "k2_sample", k_iota_stride, dim3(SF_K2_SAMPLE / 256), dim3(256), sel, (int64_t)SF_K2_SAMPLE, stride);
    SF_LAUNCH(ctx, "k2_sample", (k_radius<0, true>), dim3(sf_xcd_grid(sf_div_up(SF_K2_SAMPLE, 4 * SF_K2_WPB))), dim3(64 * SF_K2_WPB), g,
              c->cell_start, c->xs, c->ys, c->zs, nb->qx, nb->qy, nb->qz, (int64_t)SF_K2_SAMPLE, r2, 0, cnt, (int64_t *)nullptr,
              (int32_t *)nullptr, (const int32_t *)sel);
    SF_HIP(hipMemcpyAsync(pin, cnt, SF_K2_SAMPLE * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    SF_HIP(hipStreamSynchronize(ctx->stream));
    const int32_t *h = (const int32_t *)pin;
    double sum = 0.0;
    int32_t mx = 0;
    for (int i = 0; i < SF_K2_SAMPLE; ++i) { sum += h[i]; mx = std::max(mx, h[i]); }
    *cap = capacity_from(c, sum / SF_K2_SAMPLE, mx);
    return SF_OK;
}

// (for knn.hip) the lists of every (m / SF_K2_SAMPLE)-th query counted at squared radius r2: SF_K2_SAMPLE counts in cnt_dev
int sf_k2_count_sample(sf_ctx *ctx, sf_cloud *c, sf_nbrs *nb, double r2, int32_t *sel_dev, int32_t *cnt_dev)
{
    const sf_grid_desc g = sf_make_grid_desc(c);
    sf_launch_timer t_(ctx, "k2_sample");
    hipLaunchKernelGGL(k_iota_stride, dim3(SF_K2_SAMPLE / 256), dim3(256), 0, ctx->stream, sel_dev, (int64_t)SF_K2_SAMPLE, nb->m / SF_K2_SAMPLE);
    hipLaunchKernelGGL((k_radius<0, true>), dim3(sf_xcd_grid(sf_div_up(SF_K2_SAMPLE, 4 * SF_K2_WPB))), dim3(64 * SF_K2_WPB), 0, ctx->stream, g,
                       c->cell_start, c->xs, c->ys, c->zs, nb->qx, nb->qy, nb->qz, (int64_t)SF_K2_SAMPLE, r2, 0, cnt_dev, (int64_t *)nullptr,
                       (int32_t *)nullptr, (const int32_t *)sel_dev);
    SF_HIP(hipGetLastError());
    return SF_OK;
}

static int exact_scan_fill(sf_ctx *ctx, sf_cloud *c, sf_nbrs *nb, const sf_grid_desc &g, const dim3 &grid, const dim3 &block, double r2)
{
    const int64_t m = nb->m;
    auto in = rocprim::make_transform_iterator(nb->count, to_i64());
    size_t tb1 = 0;
    SF_HIP(rocprim::exclusive_scan(nullptr, tb1, in, nb->offset, (int64_t)0, (size_t)(m + 1), rocprim::plus<int64_t>(),
                                   ctx->stream));
    void *tmp = nullptr;
    SF_CHECK(sf_pool_alloc(ctx, tb1 ? tb1 : 8, &tmp));
    {
        sf_launch_timer t_(ctx, "k2_scan");
        SF_HIP(rocprim::exclusive_scan(tmp, tb1, in, nb->offset, (int64_t)0, (size_t)(m + 1), rocprim::plus<int64_t>(),
                                       ctx->stream));
    }
    sf_pool_release(ctx, tmp);
    SF_CHECK(sf_palloc(ctx, &nb->idx, (size_t)nb->total + 4));
    nb->cap = 0;
    if (nb->total) {
        SF_LAUNCH(ctx, "k2_radius_fill", (k_radius<1, false>), grid, block, g, c->cell_start, c->xs, c->ys, c->zs, nb->qx, nb->qy,
                  nb->qz, m, r2, 0, nb->count, nb->offset, nb->idx, (const int32_t *)nullptr);
    }
    return SF_OK;
}

static int run_search(sf_ctx *ctx, sf_cloud *c, sf_nbrs *nb)
{
    const int64_t m = nb->m;
    const double r2 = nb->radius * nb->radius;
    SF_CHECK(sf_palloc(ctx, &nb->count, (size_t)(m + 1)));
    SF_CHECK(sf_palloc(ctx, &nb->offset, (size_t)(m + 1)));
    // (every element of count / offset is written by the kernels below -- count[m] by the last query's wave, offset by the
    // slots kernel or the scan -- so neither array is cleared first; only the empty query set needs its single element)
    if (!m) {
        SF_HIP(hipMemsetAsync(nb->count, 0, sizeof(int32_t), ctx->stream));
        SF_HIP(hipMemsetAsync(nb->offset, 0, sizeof(int64_t), ctx->stream));
    }
    sf_grid_desc g = sf_make_grid_desc(c);
    const dim3 grid(sf_xcd_grid(sf_div_up(m ? m : 1, 4 * SF_K2_WPB))), block(64 * SF_K2_WPB); // waves x 4 queries
    if (!m) {
        // (an empty block of a sharded job still takes part in the all-reduce of the list statistics its peers are in -- and
        // stays out of it exactly when they do: on the repeated search of its (empty) range, which its peers serve from their records)
        if (ctx->collective_stats && ctx->comm) {
            const sf_cloud::search_key ekey{nb->radius, c->cell, c->xsub, nb->self_begin, 0};
            const bool seen = nb->self && !getenv("SF_K2_NO_HINT") && c->search_records.count(ekey);
            if (!seen || getenv("SF_K2_CHECK_RECORD")) {
                SF_CHECK(count_stats(ctx, nb, 0x7fffffff, nullptr));
                if (nb->self) {
                    sf_cloud::search_record r;
                    r.max_count_all = (int32_t)nb->max_count_all;
                    r.folded = true;
                    c->search_records[ekey] = r;
                }
            } else {
                nb->max_count_all = c->search_records[ekey].max_count_all;
            }
        }
        SF_CHECK(sf_palloc(ctx, &nb->idx, (size_t)8));
        return SF_OK;
    }
    // ---- single sweep into fixed-capacity slots, sized from a sample of the lists ------------------------------------------
    // (small query sets take the exact count -> scan -> fill scheme: two sweeps of next to nothing.  So do searches whose
    // slots would take more than 24 GiB.)
    int64_t cap = 0;
    bool optimistic = m >= 8 * SF_K2_SAMPLE && !getenv("SF_K2_EXACT");
    if (optimistic) {
        // the lists of the previous search with this radius on this cloud say how long this one's will be (same cloud, same
        // radius: the same lists, or a sub-range of them); the first search counts a sample instead.  SF_K2_NO_HINT=1: always.
        auto hint = c->list_stats.find({nb->radius, nb->self});
        if (hint != c->list_stats.end() && !getenv("SF_K2_NO_HINT")) cap = capacity_from(c, hint->second.first, hint->second.second);
        else SF_CHECK(sample_capacity(ctx, c, nb, g, r2, &cap));
        // slots of at most 24 GiB: a cloud whose longest list is far above its mean keeps the slots of the mean and re-does
        // the long lists
        if (const char *e = getenv("SF_K2_CAP")) { const long v = atol(e); if (v >= 32) cap = (v / 32) * 32; } // (tests: force the slot size)
        const int64_t most = std::max<int64_t>(((((int64_t)24 << 30) / (4 * m)) / 32) * 32, 32);
        if (cap > most) cap = std::max<int64_t>(most, 64);
        optimistic = cap * m * 4 <= ((int64_t)24 << 30) && cap < 0x7fffffff;
    }
    if (!optimistic) {
        SF_LAUNCH(ctx, "k2_radius_count", (k_radius<0, false>), grid, block, g, c->cell_start, c->xs, c->ys, c->zs, nb->qx, nb->qy,
                  nb->qz, m, r2, 0, nb->count, nb->offset, (int32_t *)nullptr, (const int32_t *)nullptr);
        // (small query sets: count -> scan -> fill.  A repeated self search of the range takes the total -- the size of the index
        // array -- and the rest of the statistics from its record, like the single sweep below: no read-back)
        const sf_cloud::search_key xkey{nb->radius, c->cell, c->xsub, nb->self_begin, -m}; // (-m: the exact scheme's records)
        auto xit = nb->self && !getenv("SF_K2_NO_HINT") && !getenv("SF_K2_CHECK_RECORD") ? c->search_records.find(xkey) : c->search_records.end();
        if (xit != c->search_records.end() && (xit->second.folded || !(ctx->collective_stats && ctx->comm))) {
            const sf_cloud::search_record &r = xit->second;
            nb->total = r.total;
            nb->max_count = r.max_count;
            nb->max_count_all = r.max_count_all;
            nb->n_overflow = 0;
            for (int h = 0; h < 5; ++h) nb->hist[h] = r.hist[h];
        } else {
            SF_CHECK(count_stats(ctx, nb, 0x7fffffff, nullptr));
            if (nb->self) {
                if (c->search_records.size() > 256) c->search_records.clear();
                sf_cloud::search_record r;
                r.total = nb->total; r.max_count = nb->max_count; r.max_count_all = nb->max_count_all;
                for (int h = 0; h < 5; ++h) r.hist[h] = nb->hist[h];
                r.folded = ctx->collective_stats && ctx->comm;
                c->search_records[xkey] = r;
            }
        }
        SF_CHECK(exact_scan_fill(ctx, c, nb, g, grid, block, r2));
        return plan_dispatch(ctx, c, nb);
    }
    // a self search of a range that has been searched before with this radius: its record (sf_cloud::search_records)
    const sf_cloud::search_key rkey{nb->radius, c->cell, c->xsub, nb->self_begin, m};
    const sf_cloud::search_record *rec = nullptr;
    if (nb->self && !getenv("SF_K2_NO_HINT")) {
        auto it = c->search_records.find(rkey);
        if (it != c->search_records.end() && (it->second.folded || !(ctx->collective_stats && ctx->comm))) rec = &it->second;
    }
    if (rec) cap = rec->cap;
    SF_CHECK(sf_palloc(ctx, &nb->idx, (size_t)(cap * m) + 4)); // +4: consumers read indices 16 B at a time
    nb->cap = cap;
    SF_LAUNCH(ctx, "k2_radius_slots", (k_radius<2, false>), grid, block, g, c->cell_start, c->xs, c->ys, c->zs, nb->qx, nb->qy,
              nb->qz, m, r2, (int)cap, nb->count, nb->offset, nb->idx, (const int32_t *)nullptr);
    int64_t ovf_total = 0;
    if (rec && !getenv("SF_K2_CHECK_RECORD")) { // no statistics pass, no read-back: the step's launches follow without a wait
        nb->total = rec->total;
        nb->max_count = rec->max_count;
        nb->max_count_all = rec->max_count_all;
        nb->n_overflow = rec->n_overflow;
        for (int h = 0; h < 5; ++h) nb->hist[h] = rec->hist[h];
        ovf_total = rec->ovf_total;
    } else {
        SF_CHECK(count_stats(ctx, nb, (int)cap, &ovf_total));
        if (rec && (nb->total != rec->total || nb->max_count != rec->max_count || nb->n_overflow != rec->n_overflow ||
                    ovf_total != rec->ovf_total || nb->hist[4] != rec->hist[4] || nb->hist[0] != rec->hist[0])) {
            sf_set_error("sf_radius_search_self: the lists of positions %lld .. + %lld at radius %g differ from the record of the previous "
                         "search of that range (total %lld / %lld)", (long long)nb->self_begin, (long long)m, nb->radius,
                         (long long)nb->total, (long long)rec->total);
            return SF_ERR_STATE;
        }
        // (a record is stored after EVERY first search of a range -- also the one whose sample misjudged the cloud and is re-done
        // exactly below: with the statistics folded over the ranks, count_stats is a collective, and a rank that kept no record
        // would issue its all-reduce in the next step while its peers, holding theirs, skip it -- advisor, round 5.  Such a record
        // either names slots that hold the longest list, or replays the exact re-do from its numbers.)
        if (nb->self) {
            if (c->search_records.size() > 256) c->search_records.clear();
            sf_cloud::search_record r;
            r.total = nb->total; r.n_overflow = nb->n_overflow; r.ovf_total = ovf_total; r.cap = cap;
            for (int h = 0; h < 5; ++h) r.hist[h] = nb->hist[h];
            r.max_count = nb->max_count; r.max_count_all = nb->max_count_all;
            r.folded = ctx->collective_stats && ctx->comm;
            if (r.n_overflow) { // lists outgrew their slots this time: the next search of the range takes slots that hold the longest
                const int64_t most = std::max<int64_t>(((((int64_t)24 << 30) / (4 * m)) / 32) * 32, 32);
                const int64_t cap2 = capacity_from(c, (double)nb->total / (double)m, nb->max_count);
                if (cap2 >= nb->max_count && cap2 <= most) { r.cap = cap2; r.n_overflow = 0; r.ovf_total = 0; }
            }
            c->search_records[rkey] = r;
        }
    }
    if (nb->n_overflow * 2 > m) { // the sample misjudged the cloud as a whole: redo exactly
        sf_pool_release(ctx, nb->idx);
        nb->idx = nullptr;
        nb->n_overflow = 0;
        SF_CHECK(exact_scan_fill(ctx, c, nb, g, grid, block, r2));
        return plan_dispatch(ctx, c, nb);
    }
    if (nb->n_overflow) {
        // ---- the lists that did not fit their slot, and only those: exact offsets in an area of their own ---------------------
        const int64_t no = nb->n_overflow;
        sf_pool_guard tmp(ctx);
        int32_t *sel = nullptr;
        int64_t *off = nullptr;
        SF_CHECK(select_longer(ctx, nb->count, m, (int)cap, no, &sel));
        tmp.held.push_back(sel);
        SF_CHECK(tmp.alloc(&off, (size_t)no));
        auto in = rocprim::make_transform_iterator((const int32_t *)sel, count_of{nb->count});
        size_t tb = 0;
        SF_HIP(rocprim::exclusive_scan(nullptr, tb, in, off, (int64_t)0, (size_t)no, rocprim::plus<int64_t>(), ctx->stream));
        char *scratch = nullptr;
        SF_CHECK(tmp.alloc(&scratch, tb ? tb : 8));
        {
            sf_launch_timer t_(ctx, "k2_scan");
            SF_HIP(rocprim::exclusive_scan(scratch, tb, in, off, (int64_t)0, (size_t)no, rocprim::plus<int64_t>(), ctx->stream));
        }
        SF_CHECK(sf_palloc(ctx, &nb->idx_ovf, (size_t)ovf_total + 4));
        // offset[q] is an element offset from nb->idx: the overflow area is addressed through the same base pointer
        const int64_t delta = (int64_t)(((intptr_t)nb->idx_ovf - (intptr_t)nb->idx) / (intptr_t)sizeof(int32_t));
        SF_LAUNCH(ctx, "k2_select", k_patch_offsets, dim3((unsigned)sf_div_up(no, 256)), dim3(256), (const int32_t *)sel,
                  (const int64_t *)off, no, delta, nb->offset);
        SF_LAUNCH(ctx, "k2_radius_refill", (k_radius<1, true>), dim3(sf_xcd_grid(sf_div_up(no, 4 * SF_K2_WPB))), block, g, c->cell_start,
                  c->xs, c->ys, c->zs, nb->qx, nb->qy, nb->qz, no, r2, 0, nb->count, nb->offset, nb->idx, (const int32_t *)sel);
    }
    return plan_dispatch(ctx, c, nb);
}

static int ensure_grid(sf_ctx *ctx, sf_cloud *c, double radius, int64_t need_begin = 0, int64_t need_end = -1)
{
    if (!(radius > 0.0) || !std::isfinite(radius)) {
        sf_set_error("radius must be positive and finite (got %g)", radius);
        return SF_ERR_ARG;
    }
    // a grid built for a larger radius stays valid; a far too coarse one is rebuilt for speed.  A grid that only
    // populates a block's slab (sf_cloud_build_grid_block) serves self-searches inside [need_begin, need_end).
    const bool covers = need_end < 0 ? (c->pop_begin == 0 && c->pop_end == c->n)
                                     : (need_begin >= c->pop_begin && need_end <= c->pop_end);
    if (c->cell_start && covers && c->cell >= radius && c->cell <= 2.0 * radius * (1.0 + 1e-6)) return SF_OK;
    return sf_cloud_build_grid(ctx, c, radius);
}

extern "C" sf_nbrs *sf_radius_search_self(sf_ctx *ctx, sf_cloud *c, double radius, int64_t begin, int64_t end)
{
    if (!ctx || !c) { sf_set_error("sf_radius_search_self: null argument"); return nullptr; }
    if (begin < 0 || end > c->n || begin > end) {
        sf_set_error("sf_radius_search_self: bad range [%lld, %lld) for n=%lld", (long long)begin, (long long)end,
                     (long long)c->n);
        return nullptr;
    }
    if (hipSetDevice(ctx->device) != hipSuccess) { sf_set_error("hipSetDevice failed"); return nullptr; }
    if (ensure_grid(ctx, c, radius, begin, end) != SF_OK) return nullptr;
    sf_nbrs *nb = new sf_nbrs();
    nb->m = end - begin;
    nb->radius = radius;
    nb->self = true;
    nb->self_begin = begin;
    nb->qx = c->xs + begin;
    nb->qy = c->ys + begin;
    nb->qz = c->zs + begin;
    sf_nbrs_stamp(nb, c);
    if (run_search(ctx, c, nb) != SF_OK) {
        sf_nbrs_free(ctx, nb);
        return nullptr;
    }
    return nb;
}

int sf_k2_prepare_queries(sf_ctx *ctx, sf_cloud *c, sf_nbrs *nb, const double *queries, int flags)
{
    const int64_t m = nb->m;
    size_t mm = (size_t)(m ? m : 1);
    double *dq = nullptr;
    bool own_dq = false;
    if (flags & SF_IN_DEVICE) {
        dq = const_cast<double *>(queries);
    } else {
        SF_CHECK(sf_palloc(ctx, &dq, mm * 3));
        own_dq = true;
        if (m) SF_HIP(hipMemcpyAsync(dq, queries, (size_t)m * 24, hipMemcpyHostToDevice, ctx->stream));
    }
    SF_CHECK(sf_palloc(ctx, &nb->qx, mm));
    SF_CHECK(sf_palloc(ctx, &nb->qy, mm));
    SF_CHECK(sf_palloc(ctx, &nb->qz, mm));
    SF_CHECK(sf_palloc(ctx, &nb->qrow, mm));
    if (m) {
        int32_t *cid = nullptr, *cid_s = nullptr, *val = nullptr;
        SF_CHECK(sf_palloc(ctx, &cid, mm));
        SF_CHECK(sf_palloc(ctx, &cid_s, mm));
        SF_CHECK(sf_palloc(ctx, &val, mm));
        sf_grid_desc g = sf_make_grid_desc(c);
        SF_LAUNCH(ctx, "k2_query_cells", k_query_cells, dim3((unsigned)sf_div_up(m, 256)), dim3(256), dq, m, g, cid,
                  val);
        int bits = 1;
        while (((int64_t)1 << bits) < c->ncell) ++bits;
        size_t tb = 0;
        SF_HIP(rocprim::radix_sort_pairs<sf_sort_config>(nullptr, tb, cid, cid_s, val, nb->qrow, (size_t)m, 0, bits, ctx->stream));
        void *tmp = nullptr;
        SF_CHECK(sf_pool_alloc(ctx, tb ? tb : 8, &tmp));
        {
            sf_launch_timer t_(ctx, "k2_query_sort");
            SF_HIP(rocprim::radix_sort_pairs<sf_sort_config>(tmp, tb, cid, cid_s, val, nb->qrow, (size_t)m, 0, bits, ctx->stream));
        }
        SF_LAUNCH(ctx, "k2_gather_queries", k_gather_queries, dim3((unsigned)sf_div_up(m, 256)), dim3(256), dq,
                  nb->qrow, m, nb->qx, nb->qy, nb->qz);
        sf_pool_release(ctx, tmp);
        sf_pool_release(ctx, cid);
        sf_pool_release(ctx, cid_s);
        sf_pool_release(ctx, val);
    }
    if (own_dq) {
        SF_HIP(hipStreamSynchronize(ctx->stream)); // the host source buffer of the async copy is the caller's
        sf_pool_release(ctx, dq);
    }
    return SF_OK;
}

extern "C" sf_nbrs *sf_radius_search(sf_ctx *ctx, sf_cloud *c, const double *queries, int64_t m, double radius,
                                     int flags)
{
    if (!ctx || !c || (!queries && m > 0) || m < 0 || m > 2147483000LL) {
        sf_set_error("sf_radius_search: bad arguments (m=%lld)", (long long)m);
        return nullptr;
    }
    if (hipSetDevice(ctx->device) != hipSuccess) { sf_set_error("hipSetDevice failed"); return nullptr; }
    if (ensure_grid(ctx, c, radius) != SF_OK) return nullptr;
    sf_nbrs *nb = new sf_nbrs();
    nb->m = m;
    nb->radius = radius;
    nb->self = false;
    sf_nbrs_stamp(nb, c);
    if (sf_k2_prepare_queries(ctx, c, nb, queries, flags) != SF_OK || run_search(ctx, c, nb) != SF_OK) {
        sf_nbrs_free(ctx, nb);
        return nullptr;
    }
    return nb;
}


int sf_launch_pca_solve_normals(sf_ctx *ctx, const double *cov, const int32_t *qrow, int64_t m, const double *pre, double *out); // descriptors.hip

// compute_normals(query_points, cloud_points, radius=...) (pca_based_descriptors.py:29-59) without materialising the
// neighbour lists: K1 (if needed), the fused sweep k_radius_cov, one eigen-solve per query.  queries == NULL: the cloud's own
// points at cell-sorted positions [begin, end) (row i of `out` = position begin + i); else m coordinate queries (rows in the
// caller's order).  Bit-identical to sf_radius_search(_self) + sf_normals.
extern "C" int sf_normals_radius(sf_ctx *ctx, sf_cloud *c, const double *queries, int64_t m, int64_t begin, int64_t end, double radius,
                                 const double *pre, double *out, int flags)
{
    if (!ctx || !c || !out) { sf_set_error("sf_normals_radius: null argument"); return SF_ERR_ARG; }
    if (!queries) {
        if (begin < 0 || end > c->n || begin > end) { sf_set_error("sf_normals_radius: bad range [%lld, %lld)", (long long)begin, (long long)end); return SF_ERR_ARG; }
        m = end - begin;
    } else if (m < 0 || m > 2147483000LL) {
        sf_set_error("sf_normals_radius: bad query count %lld", (long long)m);
        return SF_ERR_ARG;
    }
    SF_HIP(hipSetDevice(ctx->device));
    SF_CHECK(queries ? ensure_grid(ctx, c, radius) : ensure_grid(ctx, c, radius, begin, end));
    sf_nbrs q; // (query coordinates only: no lists are made)
    q.m = m;
    q.radius = radius;
    struct release_queries {
        sf_ctx *ctx; sf_nbrs *q; bool own;
        ~release_queries() { if (own) { sf_pool_release(ctx, q->qx); sf_pool_release(ctx, q->qy); sf_pool_release(ctx, q->qz); } sf_pool_release(ctx, q->qrow); }
    } rel{ctx, &q, queries != nullptr};
    if (queries) SF_CHECK(sf_k2_prepare_queries(ctx, c, &q, queries, flags));
    else { q.self = true; q.self_begin = begin; q.qx = c->xs + begin; q.qy = c->ys + begin; q.qz = c->zs + begin; }
    sf_pool_guard tmp(ctx);
    double *cov = nullptr, *dout = out;
    const double *dpre = pre;
    SF_CHECK(tmp.alloc(&cov, (size_t)(m ? m : 1) * 6));
    if (!(flags & SF_OUT_DEVICE)) SF_CHECK(tmp.alloc(&dout, (size_t)(m ? m : 1) * 3));
    if (pre && !(flags & SF_IN_DEVICE)) {
        double *p = nullptr;
        SF_CHECK(tmp.alloc(&p, (size_t)(m ? m : 1) * 3));
        if (m) SF_HIP(hipMemcpyAsync(p, pre, (size_t)m * 24, hipMemcpyHostToDevice, ctx->stream));
        dpre = p;
    }
    if (m) {
        sf_grid_desc g = sf_make_grid_desc(c);
        SF_LAUNCH(ctx, "k23_radius_cov", k_radius_cov, dim3(sf_xcd_grid(sf_div_up(m, 4 * SF_K2C_WPB))), dim3(64 * SF_K2C_WPB), g,
                  c->cell_start, c->xs, c->ys, c->zs, q.qx, q.qy, q.qz, m, radius * radius, cov, (double *)nullptr, (int32_t *)nullptr);
        SF_CHECK(sf_launch_pca_solve_normals(ctx, cov, q.qrow, m, dpre, dout));
        if (dout != out) SF_HIP(hipMemcpyAsync(out, dout, (size_t)m * 24, hipMemcpyDeviceToHost, ctx->stream));
    }
    if ((flags & (SF_IN_DEVICE | SF_OUT_DEVICE)) != (SF_IN_DEVICE | SF_OUT_DEVICE)) SF_HIP(hipStreamSynchronize(ctx->stream));
    return SF_OK;
}
// ---- caller-supplied neighbourhoods ------------------------------------------------------------------------------------------
// ShotMultiprocessor.compute_local_rf / compute_descriptor take the lists the caller hands them -- support[neighborhoods[i]],
// shot_parallelization.py:46-84, 86-133 -- whatever produced them: KDTree.query_radius of another radius, KDTree.query (k-NN),
// a hand-made selection.  sf_nbrs_import turns such lists (CSR: offsets[m + 1], idx[offsets[m]] in the caller's point numbering)
// into a list set the list-driven kernels (K3, K4, K5) consume like a search result: indices -> cell-sorted positions of the
// cloud's current grid (inv_perm), queries in the caller's order (no qrow), `radius` the value K4 / K5 put in their formulas
// (shot.py:28, 95-117 -- NOT a filter: a neighbour beyond it keeps its, then negative, weight exactly as in the reference).
// Order inside a list matters to no consumer beyond the rounding of a sum (K5's statements are elections by distance, K3 / K4
// are sums over the list); an index listed twice counts twice, as support[neighborhoods[i]] repeats the point.
__global__ void k_import_lists(const int64_t *__restrict__ src, int64_t total, int64_t n, const int32_t *__restrict__ inv_perm,
                               int32_t *__restrict__ idx, int *__restrict__ bad)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int64_t j = src[i];
    if (j < 0 || j >= n) { atomicOr(bad, 1); idx[i] = 0; return; }
    idx[i] = inv_perm[j];
}

__global__ void k_import_csr(const int64_t *__restrict__ offsets, int64_t m, int32_t *__restrict__ count)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i > m) return;
    count[i] = i < m ? (int32_t)(offsets[i + 1] - offsets[i]) : 0;
}

__global__ void k_split_queries(const double *__restrict__ q, int64_t m, double *__restrict__ qx, double *__restrict__ qy,
                                double *__restrict__ qz)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    qx[i] = q[3 * i];
    qy[i] = q[3 * i + 1];
    qz[i] = q[3 * i + 2];
}

extern "C" sf_nbrs *sf_nbrs_import(sf_ctx *ctx, sf_cloud *c, const double *queries, int64_t m, const int64_t *offsets,
                                   const int64_t *idx, double radius, int flags)
{
    if (!ctx || !c || m < 0 || m > 2147483000LL || !offsets || (!queries && m > 0)) {
        sf_set_error("sf_nbrs_import: bad arguments (m=%lld)", (long long)m);
        return nullptr;
    }
    if (!(radius > 0.0) || !std::isfinite(radius)) { sf_set_error("sf_nbrs_import: radius must be positive and finite (got %g)", radius); return nullptr; }
    if (hipSetDevice(ctx->device) != hipSuccess) { sf_set_error("hipSetDevice failed"); return nullptr; }
    const bool dev_in = (flags & SF_IN_DEVICE) != 0;
    // the offsets are looked at on the host (m + 1 words): monotone from 0, no list of 2^31 entries
    std::vector<int64_t> hoff;
    const int64_t *ho = offsets;
    if (dev_in) {
        hoff.resize((size_t)m + 1);
        if (hipMemcpyAsync(hoff.data(), offsets, ((size_t)m + 1) * sizeof(int64_t), hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
            hipStreamSynchronize(ctx->stream) != hipSuccess) { sf_set_error("sf_nbrs_import: reading the offsets failed"); return nullptr; }
        ho = hoff.data();
    }
    if (ho[0] != 0) { sf_set_error("sf_nbrs_import: offsets[0] must be 0 (got %lld)", (long long)ho[0]); return nullptr; }
    int64_t longest = 0;
    for (int64_t i = 0; i < m; ++i) {
        const int64_t k = ho[i + 1] - ho[i];
        if (k < 0 || k > 2147483000LL) { sf_set_error("sf_nbrs_import: offsets must ascend (list %lld has %lld entries)", (long long)i, (long long)k); return nullptr; }
        longest = std::max(longest, k);
    }
    const int64_t total = ho[m];
    if (total > 0 && !idx) { sf_set_error("sf_nbrs_import: null index array"); return nullptr; }
    // any grid of the cloud will do (the lists are not searched for): the current one, else one of `radius`
    if (!c->cell_start || !(c->pop_begin == 0 && c->pop_end == c->n)) {
        if (sf_cloud_build_grid(ctx, c, radius) != SF_OK) return nullptr;
    }
    if (sf_cloud_ensure_inv_perm(ctx, c) != SF_OK) return nullptr;
    sf_nbrs *nb = new sf_nbrs();
    nb->m = m;
    nb->radius = radius;
    nb->self = false;
    nb->total = total;
    nb->max_count = longest;
    nb->max_count_all = longest;
    sf_nbrs_stamp(nb, c);
    auto fail = [&]() { sf_nbrs_free(ctx, nb); return (sf_nbrs *)nullptr; };
    const size_t mm = (size_t)(m ? m : 1);
    if (sf_palloc(ctx, &nb->qx, mm) != SF_OK || sf_palloc(ctx, &nb->qy, mm) != SF_OK || sf_palloc(ctx, &nb->qz, mm) != SF_OK ||
        sf_palloc(ctx, &nb->count, (size_t)m + 1) != SF_OK || sf_palloc(ctx, &nb->offset, (size_t)m + 1) != SF_OK ||
        sf_palloc(ctx, &nb->idx, (size_t)total + 4) != SF_OK)
        return fail();
    sf_pool_guard tmp(ctx);
    const double *dq = queries;
    const int64_t *didx = idx;
    int *dbad = nullptr;
    if (tmp.alloc(&dbad, 1) != SF_OK) return fail();
    if (!dev_in) {
        double *q = nullptr;
        int64_t *ix = nullptr;
        if (tmp.alloc(&q, mm * 3) != SF_OK || tmp.alloc(&ix, (size_t)total + 1) != SF_OK) return fail();
        if ((m && hipMemcpyAsync(q, queries, (size_t)m * 24, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) ||
            (total && hipMemcpyAsync(ix, idx, (size_t)total * sizeof(int64_t), hipMemcpyHostToDevice, ctx->stream) != hipSuccess)) {
            sf_set_error("sf_nbrs_import: upload failed");
            return fail();
        }
        dq = q;
        didx = ix;
    }
    if (hipMemcpyAsync(nb->offset, ho, ((size_t)m + 1) * sizeof(int64_t), hipMemcpyHostToDevice, ctx->stream) != hipSuccess ||
        hipMemsetAsync(dbad, 0, sizeof(int), ctx->stream) != hipSuccess) { sf_set_error("sf_nbrs_import: upload failed"); return fail(); }
    {
        sf_launch_timer t_(ctx, "k2_import");
        hipLaunchKernelGGL(k_import_csr, dim3((unsigned)sf_div_up(m + 1, 256)), dim3(256), 0, ctx->stream, (const int64_t *)nb->offset, m, nb->count);
        if (m) hipLaunchKernelGGL(k_split_queries, dim3((unsigned)sf_div_up(m, 256)), dim3(256), 0, ctx->stream, dq, m, nb->qx, nb->qy, nb->qz);
        if (total) hipLaunchKernelGGL(k_import_lists, dim3((unsigned)sf_div_up(total, 256)), dim3(256), 0, ctx->stream, didx, total, c->n,
                                      (const int32_t *)c->inv_perm, nb->idx, dbad);
    }
    int bad = 0;
    if (hipGetLastError() != hipSuccess || hipMemcpyAsync(&bad, dbad, sizeof(int), hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
        hipStreamSynchronize(ctx->stream) != hipSuccess) { sf_set_error("sf_nbrs_import: launch failed"); return fail(); }
    if (bad) { sf_set_error("sf_nbrs_import: a neighbour index lies outside 0 .. %lld", (long long)c->n - 1); return fail(); }
    return nb;
}

extern "C" sf_nbrs *sf_nbrs_slice(sf_ctx *ctx, sf_nbrs *nb, int64_t first, int64_t count)
{
    if (!ctx || !nb || first < 0 || count < 0 || first + count > nb->m) {
        sf_set_error("sf_nbrs_slice: bad range");
        return nullptr;
    }
    sf_nbrs *v = new sf_nbrs();
    v->view = true;
    v->m = count;
    v->radius = nb->radius;
    v->self = nb->self;
    v->self_begin = nb->self_begin + first;
    v->qx = nb->qx + first;
    v->qy = nb->qy + first;
    v->qz = nb->qz + first;
    v->qrow = nb->qrow ? nb->qrow + first : nullptr;
    v->count = nb->count + first;
    v->offset = nb->offset + first; // offsets stay absolute into idx
    v->idx = nb->idx;
    v->max_count = nb->max_count;
    v->max_count_all = nb->max_count_all;
    // (a view runs the owner's dispatch: same main form, the owner's tail selection filtered to the view's range)
    for (int c = 0; c < 5; ++c) v->hist[c] = nb->hist[c];
    v->planned = nb->planned;
    v->main_chunks = nb->main_chunks;
    v->tail_limit = nb->tail_limit;
    v->tail_sel = nb->tail_sel;
    v->n_tail = nb->n_tail;
    v->mid_sel = nb->mid_sel;
    v->n_mid = nb->n_mid;
    v->view_first = nb->view_first + first;
    v->cap = nb->cap;
    v->grid_gen = nb->grid_gen;
    v->total = -1; // unknown without a device read; views are for compute, not export
    return v;
}

extern "C" int64_t sf_nbrs_num_queries(const sf_nbrs *nb) { return nb ? nb->m : -1; }
extern "C" int64_t sf_nbrs_total(const sf_nbrs *nb) { return nb ? nb->total : -1; }
extern "C" int64_t sf_nbrs_max_count(const sf_nbrs *nb) { return nb ? nb->max_count : -1; }
extern "C" int64_t sf_nbrs_max_count_all(const sf_nbrs *nb) { return nb ? std::max(nb->max_count_all, nb->max_count) : -1; }

extern "C" int sf_nbrs_export(sf_ctx *ctx, sf_cloud *c, sf_nbrs *nb, int64_t *offsets, int32_t *idx, double *dist)
{
    // The device compacts the lists into an exact CSR in the caller's point numbering (k_export_lists); the host
    // only orders each list by ascending index and the rows by the caller's query order.  For a self search
    // row i is the point at cell-sorted position self_begin + i.
    if (!ctx || !c || !nb || !offsets) { sf_set_error("sf_nbrs_export: null argument"); return SF_ERR_ARG; }
    if (nb->view) { sf_set_error("sf_nbrs_export: not available on a slice view"); return SF_ERR_ARG; }
    SF_CHECK(sf_nbrs_on_grid(nb, c, "sf_nbrs_export"));
    SF_HIP(hipSetDevice(ctx->device));
    const int64_t m = nb->m, total = nb->total;
    std::vector<int32_t> cnt((size_t)m + 1), qrow;
    SF_HIP(hipMemcpyAsync(cnt.data(), nb->count, (size_t)(m + 1) * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    if (nb->qrow && m) {
        qrow.resize((size_t)m);
        SF_HIP(hipMemcpyAsync(qrow.data(), nb->qrow, (size_t)m * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    }
    SF_HIP(hipStreamSynchronize(ctx->stream));
    std::vector<int64_t> eoff((size_t)m + 1), slot_of_row((size_t)m);
    eoff[0] = 0;
    for (int64_t s = 0; s < m; ++s) {
        eoff[(size_t)s + 1] = eoff[(size_t)s] + cnt[(size_t)s];
        slot_of_row[(size_t)(nb->qrow ? qrow[(size_t)s] : s)] = s;
    }
    offsets[0] = 0;
    for (int64_t r = 0; r < m; ++r) offsets[r + 1] = offsets[r] + cnt[(size_t)slot_of_row[(size_t)r]];
    if (!idx) return SF_OK;
    if (eoff[(size_t)m] != total) { sf_set_error("sf_nbrs_export: inconsistent counts"); return SF_ERR_STATE; }
    int64_t *d_eoff = nullptr;
    int32_t *d_idx = nullptr;
    double *d_dist = nullptr;
    SF_CHECK(sf_palloc(ctx, &d_eoff, (size_t)m + 1));
    SF_CHECK(sf_palloc(ctx, &d_idx, (size_t)total));
    if (dist) SF_CHECK(sf_palloc(ctx, &d_dist, (size_t)total));
    SF_HIP(hipMemcpyAsync(d_eoff, eoff.data(), (size_t)(m + 1) * sizeof(int64_t), hipMemcpyHostToDevice, ctx->stream));
    std::vector<int32_t> raw((size_t)(total ? total : 1));
    std::vector<double> rawd;
    if (m && total) {
        SF_LAUNCH(ctx, "k2_export_lists", k_export_lists, dim3((unsigned)sf_div_up(m, 4)), dim3(256), c->xs, c->ys, c->zs,
                  nb->qx, nb->qy, nb->qz, nb->offset, nb->count, nb->idx, d_eoff, c->perm, m, d_idx, d_dist);
        SF_HIP(hipMemcpyAsync(raw.data(), d_idx, (size_t)total * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
        if (dist) {
            rawd.resize((size_t)total);
            SF_HIP(hipMemcpyAsync(rawd.data(), d_dist, (size_t)total * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        }
    }
    SF_HIP(hipStreamSynchronize(ctx->stream));
    sf_pool_release(ctx, d_eoff);
    sf_pool_release(ctx, d_idx);
    if (d_dist) sf_pool_release(ctx, d_dist);
    std::vector<std::pair<int32_t, double>> tmp;
    for (int64_t r = 0; r < m; ++r) {
        const int64_t s = slot_of_row[(size_t)r], k = cnt[(size_t)s], base = eoff[(size_t)s];
        int32_t *dst = idx + offsets[r];
        if (!dist) {
            for (int64_t t = 0; t < k; ++t) dst[t] = raw[(size_t)(base + t)];
            std::sort(dst, dst + k);
        } else {
            tmp.resize((size_t)k);
            for (int64_t t = 0; t < k; ++t) tmp[(size_t)t] = {raw[(size_t)(base + t)], rawd[(size_t)(base + t)]};
            std::sort(tmp.begin(), tmp.end());
            for (int64_t t = 0; t < k; ++t) {
                dst[t] = tmp[(size_t)t].first;
                dist[offsets[r] + t] = tmp[(size_t)t].second;
            }
        }
    }
    return SF_OK;
}

extern "C" void sf_nbrs_free(sf_ctx *ctx, sf_nbrs *nb)
{
    if (!nb) return;
    if (nb->view) { delete nb; return; }
    if (!ctx) { delete nb; return; } // leaked on purpose: no context to return the blocks to
    if (!nb->self) {
        sf_pool_release(ctx, nb->qx);
        sf_pool_release(ctx, nb->qy);
        sf_pool_release(ctx, nb->qz);
    }
    sf_pool_release(ctx, nb->qrow);
    sf_pool_release(ctx, nb->count);
    sf_pool_release(ctx, nb->offset);
    sf_pool_release(ctx, nb->idx);
    sf_pool_release(ctx, nb->idx_ovf);
    sf_pool_release(ctx, nb->tail_sel);
    sf_pool_release(ctx, nb->mid_sel);
    delete nb;
}

