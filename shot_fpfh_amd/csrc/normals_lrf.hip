// normals_lrf.hip -- K3 (PCA normals / local PCA) and K4 (SHOT local reference frames).
//
// Replaces: compute_normals / pca        pca_based_descriptors.py:15-59   (K3)
//           get_local_rf                 shot.py:16-48                     (K4)
// Mapping: one 16-lane DPP row per query, four queries per wave.  Lists are the CSR of sorted positions written by search.hip;
// neighbours are gathered from the cell-sorted AoS records (L2-resident: consecutive queries share cells).  All arithmetic is
// float64 with FMA contraction off.  HBM roofline, algorithmic bytes: K3 24 B in + 24 B out per query; K4 24 + 72.
// (Until round 6 this was the head of descriptors.hip, K5 its body: shot.hip.)
#include "common.h"
#include "device_util.h"
#include "eigh3.h"
#include "host_stage.h"

namespace {

// --------------------------------------------------------------------------------------------------
// K3 / K4 share one structure: a wave owns 64 consecutive queries and works on FOUR of them at a time, one
// per 16-lane DPP row.  In round r, row w sweeps the list of query 16 w + r, 16 neighbours per step; the
// moment sums are reduced inside the row with four register-to-register DPP steps and lane 16 w + r --
// which sits in that same row -- keeps them (no LDS, no scratch).  Then every lane runs the
// LAPACK-compatible 3x3 eigensolver on ITS query: one solve per lane instead of one redundant solve per wave.
// Lists of ~110 points fill 16-lane steps as well as they fill 64-lane ones (7/8 against 113/128).
// --------------------------------------------------------------------------------------------------
__device__ inline double lane_bcast(double v, int src) { return __shfl(v, src); }

// longest of the four rows' lists (wave-uniform loop bound)
__device__ inline int sf_rows_max(int k)
{
    const int a = __builtin_amdgcn_readlane(k, 0), b = __builtin_amdgcn_readlane(k, 16);
    const int c = __builtin_amdgcn_readlane(k, 32), d = __builtin_amdgcn_readlane(k, 48);
    return max(max(a, b), max(c, d));
}

// One row-sweep over a list: f(x, y, z, on) for every neighbour, 4 x 16 neighbours per trip with the four index
// loads, then the four coordinate gathers, issued together.  `on` is false on padding lanes (point 0 is loaded).
template <typename F>
__device__ inline void sf_row_sweep(const double *__restrict__ rec, const int32_t *__restrict__ idx, int64_t s, int k,
                                    int kmax, int sl, F f)
{
    for (int base = 0; base < kmax; base += 64) {
        int j[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int u = base + 16 * c + sl;
            j[c] = u < k ? idx[s + u] : -1;
        }
        double x[4], y[4], z[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) sf_load_xyz(rec, j[c] < 0 ? 0 : j[c], x[c], y[c], z[c]);
#pragma unroll
        for (int c = 0; c < 4; ++c)
            if (base + 16 * c < kmax) f(x[c], y[c], z[c], j[c] >= 0); // wave-uniform test
    }
}

// K3: local PCA of every query's neighbourhood.  cov = centered^T centered / k about the barycentre
// (pca_based_descriptors.py:15-26), numpy.linalg.eigh.
//   MODE 0  normals: eigenvector of the smallest eigenvalue (:51), optional re-orientation (:53-57)
//   MODE 1  eigenvalues (ascending) + the eigenvector matrix as eigh returns it (column k = eigenvector k)
//   MODE 2  MODE 1 + the eight moments of compute_local_pca_with_moments (:121-144):
//           |mean(c V^T)| (3), mean((c V^T)^2) (3), mean(c_z), mean(c_z^2) with c the centred neighbours
template <int MODE>
__global__ __launch_bounds__(256) void k_pca(const double *__restrict__ rec, const double *__restrict__ qx,
                                             const double *__restrict__ qy, const double *__restrict__ qz,
                                             const int64_t *__restrict__ offset, const int32_t *__restrict__ cnt,
                                             const int32_t *__restrict__ idx, const int32_t *__restrict__ qrow,
                                             int64_t m, const double *__restrict__ pre, double *__restrict__ out_n,
                                             double *__restrict__ out_w, double *__restrict__ out_v,
                                             double *__restrict__ out_m, const double *__restrict__ cov_in = nullptr,
                                             const double *__restrict__ bary_in = nullptr)
{
    // cov_in / bary_in: the covariance (6 per query) and the barycentre relative to the query (3) are already there
    // (k_pca_cov, the fast form below): the two sweeps that compute them are skipped and the decomposition -- hence the
    // moments on top of it -- is bit for bit the one sf_pca returns without moments
    const int lane = threadIdx.x & 63, sl = lane & 15, rw = lane >> 4;
    const int64_t q0 = sf_uniform64((sf_xcd_block() * 4 + (threadIdx.x >> 6)) * 64);
    if (q0 >= m) return;
    const int nq = (int)(m - q0 < 64 ? m - q0 : 64);
    // every lane fetches the header of ITS query once; rows read it from lane 16 w + r with shuffles
    const bool mine = lane < nq;
    const int64_t qm = q0 + (mine ? lane : 0);
    const int64_t smine = offset[qm];
    const int kmine = mine ? cnt[qm] : 0;
    const double pxm = qx[qm], pym = qy[qm], pzm = qz[qm];
    double c11 = 0, c21 = 0, c31 = 0, c22 = 0, c32 = 0, c33 = 0;
    double bx = 0, by = 0, bz = 0; // barycentre relative to the query (kept for the moments)
    if (cov_in) {
        if (mine) {
            const double *cc = cov_in + 6 * qm, *bb = bary_in + 3 * qm;
            c11 = cc[0]; c21 = cc[1]; c31 = cc[2]; c22 = cc[3]; c32 = cc[4]; c33 = cc[5];
            bx = bb[0]; by = bb[1]; bz = bb[2];
        }
    } else
    for (int r = 0; r < 16; ++r) {
        const int src = 16 * rw + r;
        const int64_t s = __shfl(smine, src);
        const int k = __shfl(kmine, src);
        const double px = __shfl(pxm, src), py = __shfl(pym, src), pz = __shfl(pzm, src);
        const int kmax = sf_rows_max(k);
        // pass 1: barycentre, accumulated relative to the query to keep the sums small
        double sx = 0.0, sy = 0.0, sz = 0.0;
        sf_row_sweep(rec, idx, s, k, kmax, sl, [&](double x, double y, double z, bool on) {
            sx += on ? x - px : 0.0;
            sy += on ? y - py : 0.0;
            sz += on ? z - pz : 0.0;
        });
        const double kk = (double)k;
        const double mx = sf_row16_sum(sx) / kk, my = sf_row16_sum(sy) / kk, mz = sf_row16_sum(sz) / kk;
        // pass 2: lower triangle of the centred second moments
        double a11 = 0, a21 = 0, a31 = 0, a22 = 0, a32 = 0, a33 = 0;
        sf_row_sweep(rec, idx, s, k, kmax, sl, [&](double x, double y, double z, bool on) {
            const double ax = on ? (x - px) - mx : 0.0, ay = on ? (y - py) - my : 0.0, az = on ? (z - pz) - mz : 0.0;
            a11 += ax * ax;
            a21 += ay * ax;
            a31 += az * ax;
            a22 += ay * ay;
            a32 += az * ay;
            a33 += az * az;
        });
        a11 = sf_row16_sum(a11) / kk;
        a21 = sf_row16_sum(a21) / kk;
        a31 = sf_row16_sum(a31) / kk;
        a22 = sf_row16_sum(a22) / kk;
        a32 = sf_row16_sum(a32) / kk;
        a33 = sf_row16_sum(a33) / kk;
        if (sl == r) {
            c11 = a11; c21 = a21; c31 = a31; c22 = a22; c32 = a32; c33 = a33;
            bx = mx; by = my; bz = mz;
        }
    }
    sf_eig::eig3 e;
    e.w1 = e.w2 = e.w3 = 0.0;
    e.v11 = e.v21 = e.v31 = e.v12 = e.v22 = e.v32 = e.v13 = e.v23 = e.v33 = 0.0;
    if (mine) e = sf_eig::eigh3_lower(c11, c21, c31, c22, c32, c33);
    const int64_t row = mine ? (qrow ? (int64_t)qrow[qm] : qm) : 0;
    if (MODE == 0) {
        if (mine) {
            double nx = e.v11, ny = e.v21, nz = e.v31;
            if (pre) {
                const double dot = (nx * pre[3 * row] + ny * pre[3 * row + 1]) + nz * pre[3 * row + 2];
                if (dot < 0.0) { nx = -nx; ny = -ny; nz = -nz; }
            }
            out_n[3 * row + 0] = nx;
            out_n[3 * row + 1] = ny;
            out_n[3 * row + 2] = nz;
        }
        return;
    }
    if (mine) {
        out_w[3 * row + 0] = e.w1; out_w[3 * row + 1] = e.w2; out_w[3 * row + 2] = e.w3;
        double *v = out_v + 9 * row; // row-major: v[3 i + k] = component i of eigenvector k
        v[0] = e.v11; v[1] = e.v12; v[2] = e.v13;
        v[3] = e.v21; v[4] = e.v22; v[5] = e.v23;
        v[6] = e.v31; v[7] = e.v32; v[8] = e.v33;
    }
    if (MODE == 2) {
        double m0 = 0, m1 = 0, m2 = 0, m3 = 0, m4 = 0, m5 = 0, m6 = 0, m7 = 0;
        for (int r = 0; r < 16; ++r) {
            const int src = 16 * rw + r;
            const int64_t s = __shfl(smine, src);
            const int k = __shfl(kmine, src);
            const double px = __shfl(pxm, src), py = __shfl(pym, src), pz = __shfl(pzm, src);
            const double mx = lane_bcast(bx, src), my = lane_bcast(by, src), mz = lane_bcast(bz, src);
            // moment = centred @ eigenvectors.T : component i uses ROW i of the eigenvector matrix (:124)
            const double r11 = lane_bcast(e.v11, src), r12 = lane_bcast(e.v12, src), r13 = lane_bcast(e.v13, src);
            const double r21 = lane_bcast(e.v21, src), r22 = lane_bcast(e.v22, src), r23 = lane_bcast(e.v23, src);
            const double r31 = lane_bcast(e.v31, src), r32 = lane_bcast(e.v32, src), r33 = lane_bcast(e.v33, src);
            const int kmax = sf_rows_max(k);
            double t0 = 0, t1 = 0, t2 = 0, t3 = 0, t4 = 0, t5 = 0, t6 = 0, t7 = 0;
            sf_row_sweep(rec, idx, s, k, kmax, sl, [&](double x, double y, double z, bool on) {
                const double ax = on ? (x - px) - mx : 0.0, ay = on ? (y - py) - my : 0.0, az = on ? (z - pz) - mz : 0.0;
                const double u0 = (ax * r11 + ay * r12) + az * r13;
                const double u1 = (ax * r21 + ay * r22) + az * r23;
                const double u2 = (ax * r31 + ay * r32) + az * r33;
                t0 += u0; t1 += u1; t2 += u2;
                t3 += u0 * u0; t4 += u1 * u1; t5 += u2 * u2;
                t6 += az; t7 += az * az;
            });
            const double kk = (double)k;
            t0 = fabs(sf_row16_sum(t0) / kk); t1 = fabs(sf_row16_sum(t1) / kk); t2 = fabs(sf_row16_sum(t2) / kk);
            t3 = sf_row16_sum(t3) / kk; t4 = sf_row16_sum(t4) / kk; t5 = sf_row16_sum(t5) / kk;
            t6 = sf_row16_sum(t6) / kk; t7 = sf_row16_sum(t7) / kk;
            if (sl == r) { m0 = t0; m1 = t1; m2 = t2; m3 = t3; m4 = t4; m5 = t5; m6 = t6; m7 = t7; }
        }
        if (mine) {
            double *o = out_m + 8 * row;
            o[0] = m0; o[1] = m1; o[2] = m2; o[3] = m3; o[4] = m4; o[5] = m5; o[6] = m6; o[7] = m7;
        }
    }
}

// K3 from materialised lists (normals and the plain decomposition): TWO kernels.
//   k_pca_cov<NCH>   one WAVE per query: every neighbour is gathered once into registers (one index round trip, one
//                    gather round trip -- k_pca's 16-lane rows take two dependent round trips per 16 neighbours, twice over),
//                    the barycentre and the centred second moments are wave reductions (the second "sweep" of
//                    pca_based_descriptors.py:21-23 runs on registers), 6 doubles per query go to memory;
//   k_pca_solve<M>   one eigen-solve per lane on those, outputs as k_pca writes them.
// Same arithmetic per term as k_pca (coordinates relative to the query, mean subtracted before the products, / k); only
// the association of the sums differs, as it already did between k_pca's lane-strided partial sums and NumPy's.
// NCH = 0: streaming form for any list length (two passes over the list, as the reference's `centred = x - mean` makes).
// limit / SEL: dispatch by list length, per query (sf_nbrs_dispatch) -- the main launch leaves out the queries whose own
// list exceeds its form, a second launch (SEL, NCH = 0) serves exactly those.
template <int NCH, bool SEL>
__global__ __launch_bounds__(128) void k_pca_cov(const double *__restrict__ rec, const double *__restrict__ qx,
                                                 const double *__restrict__ qy, const double *__restrict__ qz,
                                                 const int64_t *__restrict__ offset, const int32_t *__restrict__ cnt,
                                                 const int32_t *__restrict__ idx, int64_t m, double *__restrict__ cov,
                                                 double *__restrict__ bary, int limit, const int32_t *__restrict__ sel,
                                                 int64_t nsel, int64_t view_first)
{
    const int lane = threadIdx.x & 63;
    int64_t q = sf_uniform64(sf_xcd_block() * 2 + (threadIdx.x >> 6));
    if (SEL) {
        if (q >= nsel) return;
        q = (int64_t)sf_uniform(sel[q]) - view_first;
        if (q < 0) return;
    }
    if (q >= m) return;
    const int64_t s = offset[q];
    const int k = sf_uniform(cnt[q]);
    if (!SEL && k > limit) return;
    const double px = qx[q], py = qy[q], pz = qz[q];
    const double kk = (double)k;
    constexpr int NC = NCH > 0 ? NCH : 1;
    int jj[NC];
    double x[NC], y[NC], z[NC];
    double sx = 0.0, sy = 0.0, sz = 0.0;
    if (NCH > 0) {
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int t = c * 64 + lane;
            jj[c] = -1;
            if (c == 0 || c * 64 < k) jj[c] = t < k ? idx[s + t] : -1;
        }
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            x[c] = y[c] = z[c] = 0.0;
            if (c == 0 || c * 64 < k) {
                double gx, gy, gz;
                sf_load_xyz(rec, jj[c] < 0 ? 0 : jj[c], gx, gy, gz);
                const bool on = jj[c] >= 0;
                x[c] = on ? gx - px : 0.0;
                y[c] = on ? gy - py : 0.0;
                z[c] = on ? gz - pz : 0.0;
            }
        }
#pragma unroll
        for (int c = 0; c < NC; ++c) { sx += x[c]; sy += y[c]; sz += z[c]; }
    } else {
        for (int t = lane; t < k; t += 64) {
            double gx, gy, gz;
            sf_load_xyz(rec, idx[s + t], gx, gy, gz);
            sx += gx - px; sy += gy - py; sz += gz - pz;
        }
    }
    // (ONE division per query: 1 / k, then products -- a float64 division is ~35 instructions executed by the whole wave, and
    // the four of `mean = sum / k`, `cov = moments / k` were a fifth of this kernel; the quotients differ from true divisions
    // in the last bit, as the sums already differ from NumPy's in their association.  k_radius_cov forms the same products.)
    const double ik = 1.0 / kk;
    const double bs[4] = {sx, sy, sz, 0.0};
    const double bt = sf_wave_sum4(bs); // (row i of 16 lanes: the sum of bs[i]; k_radius_cov reduces the same way)
    const double mx = sf_read_lane(bt, 0) * ik, my = sf_read_lane(bt, 16) * ik, mz = sf_read_lane(bt, 32) * ik;
    double part[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (NCH > 0) {
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const bool on = jj[c] >= 0;
            const double ax = on ? x[c] - mx : 0.0, ay = on ? y[c] - my : 0.0, az = on ? z[c] - mz : 0.0;
            part[0] += ax * ax;
            part[1] += ay * ax;
            part[2] += az * ax;
            part[3] += ay * ay;
            part[4] += az * ay;
            part[5] += az * az;
        }
    } else {
        for (int t = lane; t < k; t += 64) {
            double gx, gy, gz;
            sf_load_xyz(rec, idx[s + t], gx, gy, gz);
            const double ax = (gx - px) - mx, ay = (gy - py) - my, az = (gz - pz) - mz;
            part[0] += ax * ax;
            part[1] += ay * ax;
            part[2] += az * ax;
            part[3] += ay * ay;
            part[4] += az * ay;
            part[5] += az * az;
        }
    }
    const double tot = sf_wave_sum8(part); // lanes 8 i .. 8 i + 7 hold the sum of part[i]
    const int e = lane >> 3;
    if ((lane & 7) == 0 && e < 6) cov[6 * q + e] = tot * ik; // c11 c21 c31 c22 c32 c33
    if (bary && lane == 0) { bary[3 * q] = mx; bary[3 * q + 1] = my; bary[3 * q + 2] = mz; }
}

template <int MODE>
__global__ __launch_bounds__(64) void k_pca_solve(const double *__restrict__ cov, const int32_t *__restrict__ qrow, int64_t m,
                                                  const double *__restrict__ pre, double *__restrict__ out_n,
                                                  double *__restrict__ out_w, double *__restrict__ out_v)
{
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= m) return;
    const double *c = cov + 6 * q;
    const sf_eig::eig3 e = sf_eig::eigh3_lower(c[0], c[1], c[2], c[3], c[4], c[5]);
    const int64_t row = qrow ? (int64_t)qrow[q] : q;
    if (MODE == 0) {
        double nx = e.v11, ny = e.v21, nz = e.v31;
        if (pre) {
            const double dot = (nx * pre[3 * row] + ny * pre[3 * row + 1]) + nz * pre[3 * row + 2];
            if (dot < 0.0) { nx = -nx; ny = -ny; nz = -nz; }
        }
        out_n[3 * row + 0] = nx;
        out_n[3 * row + 1] = ny;
        out_n[3 * row + 2] = nz;
    } else {
        out_w[3 * row + 0] = e.w1; out_w[3 * row + 1] = e.w2; out_w[3 * row + 2] = e.w3;
        double *v = out_v + 9 * row; // row-major: v[3 i + k] = component i of eigenvector k
        v[0] = e.v11; v[1] = e.v12; v[2] = e.v13;
        v[3] = e.v21; v[4] = e.v22; v[5] = e.v23;
        v[6] = e.v31; v[7] = e.v32; v[8] = e.v33;
    }
}

// K4: SHOT local reference frame (shot.py:16-48), query included in its own support.
// A wave owns 64 consecutive queries and works on FOUR of them at a time, one per 16-lane DPP row: in round
// r, row w sweeps the list of query 16 w + r, 16 neighbours per step.  The seven moment sums are then reduced
// inside the row with four register-to-register DPP steps (a full-wave reduction costs six steps plus a
// readlane, per query instead of per four queries), and lane 16 w + r -- which sits in that same row -- keeps
// them, so that phase B runs one eigen-solve per lane.  Lists of ~110 points fill 16-lane steps as well as
// they fill 64-lane ones (7/8 against 113/128).
// sqrt here only feeds the continuous weight r - |c| (no bin or sign decision hangs on its last bit), so it
// is the 8-instruction v_rsq_f64 + Newton form instead of the correctly rounded 22-instruction one.
__global__ __launch_bounds__(256) void k_shot_lrf(const double *__restrict__ rec, const double *__restrict__ qx,
                                                  const double *__restrict__ qy, const double *__restrict__ qz,
                                                  const int64_t *__restrict__ offset, const int32_t *__restrict__ cnt,
                                                  const int32_t *__restrict__ idx, const int32_t *__restrict__ qrow,
                                                  int64_t m, double radius, int raw, int skip_zero,
                                                  double *__restrict__ lrf)
{
    // skip_zero != 0: the support is the list minus its points at distance zero -- the serial compute_shot_descriptor
    // drops them BEFORE get_local_rf (shot.py:361-363), so neither their weight r nor their ">= 0" vote counts.
    // raw != 0: stop after the eigen-decomposition and store the largest / smallest eigenvectors as returned, with their
    // cross product, in the frame's layout (shot_finish_frame); the fused SHOT kernel does the sign votes from the neighbours it has
    // in registers anyway and completes the frame in place.
    const int lane = threadIdx.x & 63, sl = lane & 15, rw = lane >> 4;
    const int64_t q0 = sf_uniform64((sf_xcd_block() * 4 + (threadIdx.x >> 6)) * 64);
    if (q0 >= m) return;
    const int nq = (int)(m - q0 < 64 ? m - q0 : 64);
    // Every lane fetches the header of ITS query once (coalesced); in round r the row reads the header of
    // query 16 w + r from lane 16 w + r with shuffles, so a round starts without a memory round trip.
    const bool mine = lane < nq;
    const int64_t qm = q0 + (mine ? lane : 0);
    const int64_t smine = offset[qm];
    const int kmine = mine ? cnt[qm] : 0;
    const double pxm = qx[qm], pym = qy[qm], pzm = qz[qm];
    // phase A: weighted covariance, w = r - ||c|| (shot.py:27-35)
    double c11 = 0, c21 = 0, c31 = 0, c22 = 0, c32 = 0, c33 = 0;
    for (int r = 0; r < 16; ++r) {
        const int src = 16 * rw + r;
        const int64_t s = __shfl(smine, src);
        const int k = __shfl(kmine, src);
        const double px = __shfl(pxm, src), py = __shfl(pym, src), pz = __shfl(pzm, src);
        const int kmax = sf_rows_max(k);
        double ws = 0, a11 = 0, a21 = 0, a31 = 0, a22 = 0, a32 = 0, a33 = 0;
        for (int base = 0; base < kmax; base += 128) { // eight 16-neighbour steps per trip, loads issued together
            int j[8];
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const int u = base + 16 * c + sl;
                j[c] = u < k ? idx[s + u] : -1;
            }
            double x[8], y[8], z[8];
#pragma unroll
            for (int c = 0; c < 8; ++c)
                if (base + 16 * c < kmax) sf_load_xyz(rec, j[c] < 0 ? 0 : j[c], x[c], y[c], z[c]); // wave-uniform test
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                if (base + 16 * c < kmax) {
                    const double cx = x[c] - px, cy = y[c] - py, cz = z[c] - pz;
                    const double d2 = (cx * cx + cy * cy) + cz * cz;
                    const double wv = radius - sf_sqrt_fast(d2);
                    const double w = (j[c] < 0) | ((skip_zero != 0) & (d2 == 0.0)) ? 0.0 : wv;
                    ws += w;
                    const double wx = cx * w, wy = cy * w, wz = cz * w;
                    a11 += cx * wx; a21 += cy * wx; a31 += cz * wx;
                    a22 += cy * wy; a32 += cz * wy; a33 += cz * wz;
                }
            }
        }
        const double iw = sf_rcp_fast(sf_row16_sum(ws)); // empty list: 0 * inf = NaN, overridden by the k == 0 rule
        a11 = sf_row16_sum(a11) * iw;
        a21 = sf_row16_sum(a21) * iw;
        a31 = sf_row16_sum(a31) * iw;
        a22 = sf_row16_sum(a22) * iw;
        a32 = sf_row16_sum(a32) * iw;
        a33 = sf_row16_sum(a33) * iw;
        if (sl == r) { c11 = a11; c21 = a21; c31 = a31; c22 = a22; c32 = a32; c33 = a33; }
    }
    // phase B: one eigen-decomposition per lane (shot.py:36)
    double x0 = 0, x1 = 0, x2 = 0, z0 = 0, z1 = 0, z2 = 0;
    if (lane < nq) {
        const sf_eig::eig3 e = sf_eig::eigh3_lower(c11, c21, c31, c22, c32, c33);
        x0 = e.v13; x1 = e.v23; x2 = e.v33; // eigenvectors[:, 2]
        z0 = e.v11; z1 = e.v21; z2 = e.v31; // eigenvectors[:, 0]
    }
    if (raw) {
        if (lane < nq) {
            const int64_t q = q0 + lane;
            double *o = lrf + 9 * (qrow ? qrow[q] : q);
            // (the frame's final layout, unflipped, y = cross(z, x) of the unflipped axes: shot_finish_frame)
            o[0] = x0; o[1] = z1 * x2 - z2 * x1; o[2] = z0;
            o[3] = x1; o[4] = z2 * x0 - z0 * x2; o[5] = z1;
            o[6] = x2; o[7] = z0 * x1 - z1 * x0; o[8] = z2;
        }
        return;
    }
    // phase C: sign votes (shot.py:40-45): flip when strictly more neighbours project negative than >= 0
    bool flipx = false, flipz = false;
    int nvote = 0;
    for (int r = 0; r < 16; ++r) {
        const int src = 16 * rw + r; // lane holding this row's header and axes
        const int64_t s = __shfl(smine, src);
        const int k = __shfl(kmine, src);
        const double px = __shfl(pxm, src), py = __shfl(pym, src), pz = __shfl(pzm, src);
        const int kmax = sf_rows_max(k);
        const double bx0 = lane_bcast(x0, src), bx1 = lane_bcast(x1, src), bx2 = lane_bcast(x2, src);
        const double bz0 = lane_bcast(z0, src), bz1 = lane_bcast(z1, src), bz2 = lane_bcast(z2, src);
        int xneg = 0, zneg = 0, nzero = 0;
        for (int base = 0; base < kmax; base += 64) {
            int j[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int u = base + 16 * c + sl;
                j[c] = u < k ? idx[s + u] : -1;
            }
            double x[4], y[4], z[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) sf_load_xyz(rec, j[c] < 0 ? 0 : j[c], x[c], y[c], z[c]);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const double cx = x[c] - px, cy = y[c] - py, cz = z[c] - pz;
                const double xo = (cx * bx0 + cy * bx1) + cz * bx2;
                const double zo = (cx * bz0 + cy * bz1) + cz * bz2;
                xneg += ((j[c] >= 0) & (xo < 0.0)) ? 1 : 0;
                zneg += ((j[c] >= 0) & (zo < 0.0)) ? 1 : 0;
                nzero += ((j[c] >= 0) & (((cx * cx + cy * cy) + cz * cz) == 0.0)) ? 1 : 0;
            }
        }
        xneg = sf_row16_sum(xneg);
        zneg = sf_row16_sum(zneg);
        const int kv = k - (skip_zero ? sf_row16_sum(nzero) : 0); // voters
        // coordinates are finite (checked at upload), so the ">= 0" voters are the remaining kv - neg
        if (sl == r) { flipx = xneg > kv - xneg; flipz = zneg > kv - zneg; nvote = kv; }
    }
    if (lane < nq) {
        const int64_t q = q0 + lane;
        const int64_t row = qrow ? qrow[q] : q;
        double *o = lrf + 9 * row;
        if (nvote == 0) { // empty support: shot.py:24-25
            o[0] = 1.0; o[1] = 0.0; o[2] = 0.0;
            o[3] = 0.0; o[4] = 1.0; o[5] = 0.0;
            o[6] = 0.0; o[7] = 0.0; o[8] = 1.0;
        } else {
            if (flipx) { x0 = -x0; x1 = -x1; x2 = -x2; }
            if (flipz) { z0 = -z0; z1 = -z1; z2 = -z2; }
            const double y0 = z1 * x2 - z2 * x1, y1 = z2 * x0 - z0 * x2, y2 = z0 * x1 - z1 * x0; // cross(z, x) :46
            o[0] = x0; o[1] = y0; o[2] = z0; // columns [x y z] (:48)
            o[3] = x1; o[4] = y1; o[5] = z1;
            o[6] = x2; o[7] = y2; o[8] = z2;
        }
    }
}
} // namespace

static int launch_pca_cov(sf_ctx *ctx, sf_cloud *c, sf_nbrs *nb, double *cov, double *bary = nullptr)
{
    const int64_t m = nb->m;
    const dim3 grid(sf_xcd_grid(sf_div_up(m, 2))), block(128);
    const sf_dispatch d = sf_nbrs_dispatch(nb);
#define SF_K3_COV(NAME, NCH, SEL, GRID)                                                                                 \
    SF_LAUNCH(ctx, NAME, (k_pca_cov<NCH, SEL>), GRID, block, c->rec, nb->qx, nb->qy, nb->qz, nb->offset, nb->count, nb->idx, m, cov, \
              bary, d.limit, d.tail_sel, d.n_tail, d.view_first)
    if (d.chunks == 1) { SF_K3_COV("k3_normals", 1, false, grid); }
    else if (d.chunks == 2) { SF_K3_COV("k3_normals", 2, false, grid); }
    else if (d.chunks == 3) { SF_K3_COV("k3_normals", 3, false, grid); }
    else if (d.chunks == 4) { SF_K3_COV("k3_normals", 4, false, grid); }
    else { SF_K3_COV("k3_normals", 0, false, grid); }
    if (d.n_mid) {
        SF_LAUNCH(ctx, "k3_normals_mid", (k_pca_cov<4, true>), dim3(sf_xcd_grid(sf_div_up(d.n_mid, 2))), block, c->rec, nb->qx, nb->qy, nb->qz,
                  nb->offset, nb->count, nb->idx, m, cov, bary, 255, d.mid_sel, d.n_mid, d.view_first);
    }
    if (d.n_tail) { SF_K3_COV("k3_normals_tail", 0, true, dim3(sf_xcd_grid(sf_div_up(d.n_tail, 2)))); }
#undef SF_K3_COV
    return SF_OK;
}

// (for search.hip::sf_normals_radius: the eigen-solves of the fused K2 + K3 sweep)
int sf_launch_pca_solve_normals(sf_ctx *ctx, const double *cov, const int32_t *qrow, int64_t m, const double *pre, double *out)
{
    if (!m) return SF_OK;
    SF_LAUNCH(ctx, "k3_normals", k_pca_solve<0>, dim3((unsigned)sf_div_up(m, 64)), dim3(64), cov, qrow, m, pre, out, (double *)nullptr,
              (double *)nullptr);
    return SF_OK;
}

extern "C" int sf_normals(sf_ctx *ctx, sf_cloud *c, sf_nbrs *nb, const double *pre, double *out, int flags)
{
    SF_CHECK(check_nbrs(ctx, c, nb, "sf_normals"));
    if (!out) { sf_set_error("sf_normals: null output"); return SF_ERR_ARG; }
    const int64_t m = nb->m;
    sf_pool_guard tmp(ctx);
    const double *dpre;
    double *dout;
    SF_CHECK(stage_in(tmp, pre, (size_t)m * 3, flags, &dpre));
    SF_CHECK(stage_out(tmp, out, (size_t)m * 3, flags, &dout));
    if (m) { // covariance per wave (k_pca_cov), then the eigen-solves a lane each (k_pca_solve)
        double *cov = nullptr;
        SF_CHECK(tmp.alloc(&cov, (size_t)m * 6));
        SF_CHECK(launch_pca_cov(ctx, c, nb, cov));
        SF_LAUNCH(ctx, "k3_normals", k_pca_solve<0>, dim3((unsigned)sf_div_up(m, 64)), dim3(64), (const double *)cov,
                  (const int32_t *)nb->qrow, m, dpre, dout, (double *)nullptr, (double *)nullptr);
    }
    SF_CHECK(finish_out(ctx, out, (size_t)m * 3, flags, dout));
    return stage_sync(ctx, flags);
}

extern "C" int sf_pca(sf_ctx *ctx, sf_cloud *c, sf_nbrs *nb, double *eigenvalues, double *eigenvectors, double *moments,
                      int flags)
{
    SF_CHECK(check_nbrs(ctx, c, nb, "sf_pca"));
    if (!eigenvalues || !eigenvectors) { sf_set_error("sf_pca: null output"); return SF_ERR_ARG; }
    const int64_t m = nb->m;
    sf_pool_guard tmp(ctx);
    double *dw, *dv, *dm = nullptr;
    SF_CHECK(stage_out(tmp, eigenvalues, (size_t)m * 3, flags, &dw));
    SF_CHECK(stage_out(tmp, eigenvectors, (size_t)m * 9, flags, &dv));
    if (moments) SF_CHECK(stage_out(tmp, moments, (size_t)m * 8, flags, &dm));
    if (m) {
        const dim3 grid(sf_xcd_grid(sf_div_up(m, 256))), block(256);
        if (moments) {
            double *cov = nullptr, *bary = nullptr;
            SF_CHECK(tmp.alloc(&cov, (size_t)m * 6));
            SF_CHECK(tmp.alloc(&bary, (size_t)m * 3));
            SF_CHECK(launch_pca_cov(ctx, c, nb, cov, bary));
            SF_LAUNCH(ctx, "k3_pca_moments", k_pca<2>, grid, block, c->rec, nb->qx, nb->qy, nb->qz, nb->offset, nb->count,
                      nb->idx, nb->qrow, m, (const double *)nullptr, (double *)nullptr, dw, dv, dm, (const double *)cov,
                      (const double *)bary);
        } else {
            double *cov = nullptr;
            SF_CHECK(tmp.alloc(&cov, (size_t)m * 6));
            SF_CHECK(launch_pca_cov(ctx, c, nb, cov));
            SF_LAUNCH(ctx, "k3_pca", k_pca_solve<1>, dim3((unsigned)sf_div_up(m, 64)), dim3(64), (const double *)cov,
                      (const int32_t *)nb->qrow, m, (const double *)nullptr, (double *)nullptr, dw, dv);
        }
    }
    SF_CHECK(finish_out(ctx, eigenvalues, (size_t)m * 3, flags, dw));
    SF_CHECK(finish_out(ctx, eigenvectors, (size_t)m * 9, flags, dv));
    if (moments) SF_CHECK(finish_out(ctx, moments, (size_t)m * 8, flags, dm));
    return stage_sync(ctx, flags);
}

extern "C" int sf_shot_lrf(sf_ctx *ctx, sf_cloud *c, sf_nbrs *nb, double *lrf, int flags)
{
    SF_CHECK(check_nbrs(ctx, c, nb, "sf_shot_lrf"));
    if (!lrf) { sf_set_error("sf_shot_lrf: null output"); return SF_ERR_ARG; }
    const int64_t m = nb->m;
    sf_pool_guard tmp(ctx);
    double *dout;
    SF_CHECK(stage_out(tmp, lrf, (size_t)m * 9, flags, &dout));
    if (m) {
        SF_LAUNCH(ctx, "k4_shot_lrf", k_shot_lrf, dim3(sf_xcd_grid(sf_div_up(m, 256))), dim3(256), c->rec,
                  nb->qx, nb->qy, nb->qz, nb->offset, nb->count, nb->idx, nb->qrow, m, nb->radius, 0, 0, dout);
    }
    SF_CHECK(finish_out(ctx, lrf, (size_t)m * 9, flags, dout));
    return stage_sync(ctx, flags);
}

namespace {
// K4 when the moments come from K6 (sf_spfh_compute_moments): one LAPACK-compatible 3 x 3 eigen-solve per lane, the
// largest / smallest eigenvectors stored as k_shot_lrf does in its raw mode (the fused K5 completes the frame).
// It is meant to run on the side stream UNDER K7, which holds eight 64-register waves on every SIMD: a wave of this kernel
// only ever finds room if it fits the hole ONE retiring K7 wave leaves (<= 64 VGPRs, single-wave workgroups) -- at 84
// registers in 4-wave workgroups it was starved until K7's tail and ended after it (1.43 ms against K7's 1.33 ms).
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(7, 8))) void k_lrf_from_cov(const double *__restrict__ cov, int64_t m, double *__restrict__ lrf)
{
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= m) return;
    const double *c = cov + 6 * q;
    const sf_eig::eig3 e = sf_eig::eigh3_lower(c[0], c[1], c[2], c[3], c[4], c[5]);
    double *o = lrf + 9 * q;
    const double x0 = e.v13, x1 = e.v23, x2 = e.v33; // eigenvectors[:, 2]
    const double z0 = e.v11, z1 = e.v21, z2 = e.v31; // eigenvectors[:, 0]
    // (the frame's final layout, unflipped, y = cross(z, x) of the unflipped axes: shot_finish_frame)
    o[0] = x0; o[1] = z1 * x2 - z2 * x1; o[2] = z0;
    o[3] = x1; o[4] = z2 * x0 - z0 * x2; o[5] = z1;
    o[6] = x2; o[7] = z0 * x1 - z1 * x0; o[8] = z2;
}

} // namespace

// The two halves of sf_shot_from_moments as separate calls, for a caller that runs the eigen-solves on the side
// stream next to K7 (neither needs the other; K5 needs both K6 and the frames): device pointers only.
extern "C" int sf_lrf_raw_from_moments(sf_ctx *ctx, sf_cloud *c, sf_nbrs *nb, const double *cov_dev, double *lrf_dev)
{
    SF_CHECK(check_nbrs(ctx, c, nb, "sf_lrf_raw_from_moments"));
    if (!cov_dev || !lrf_dev) { sf_set_error("sf_lrf_raw_from_moments: null argument"); return SF_ERR_ARG; }
    if (nb->qrow) { sf_set_error("sf_lrf_raw_from_moments: needs a self search"); return SF_ERR_UNSUPPORTED; }
    const int64_t m = nb->m;
    if (m) {
        SF_LAUNCH(ctx, "k4_lrf_from_cov", k_lrf_from_cov, dim3((unsigned)sf_div_up(m, 64)), dim3(64), cov_dev, m, lrf_dev);
    }
    return SF_OK;
}

// K4's launches for the SHOT entry points of shot.hip: the frames of `nb`'s queries (raw != 0: the eigenvectors only, the sign votes
// are taken by the fused K5; skip_zero != 0: the neighbours at distance zero left out -- the serial variant, shot.py:361-363), and
// the frames from moments K6 accumulated.
int sf_launch_shot_lrf(sf_ctx *ctx, sf_cloud *c, sf_nbrs *nb, int raw, int skip_zero, double *dlrf)
{
    const int64_t m = nb->m;
    if (m) {
        SF_LAUNCH(ctx, "k4_shot_lrf", k_shot_lrf, dim3(sf_xcd_grid(sf_div_up(m, 256))), dim3(256), c->rec, nb->qx, nb->qy, nb->qz,
                  nb->offset, nb->count, nb->idx, nb->qrow, m, nb->radius, raw, skip_zero, dlrf);
    }
    return SF_OK;
}

int sf_launch_lrf_from_cov(sf_ctx *ctx, const double *cov_dev, int64_t m, double *dlrf)
{
    if (m) SF_LAUNCH(ctx, "k4_lrf_from_cov", k_lrf_from_cov, dim3((unsigned)sf_div_up(m, 64)), dim3(64), cov_dev, m, dlrf);
    return SF_OK;
}

