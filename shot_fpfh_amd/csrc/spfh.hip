// spfh.hip -- K6: SPFH integer histograms for every cloud point, the table they live in and its exchange between ranks.
// (Until round 6: the first half of fpfh.hip; K7, the weighted reduction, stays there.)
//
//
// Replaces: compute_fpfh_descriptor, fpfh.py:16-117 (decorrelated=False):
//   K6  fpfh.py:38-90   per point i, per neighbour j with d > 0:  u = n_i, v = (p_j-p_i) x u (NOT
//       normalised), w = u x v, alpha = v.n_j, phi = (p_j-p_i).u / d, theta = atan2(n_j.w, n_j.u);
//       np.histogramdd over (-1,1) x (-1,1) x (-pi/2,pi/2) with np.linspace edges -- samples outside
//       any range are DROPPED while the normaliser stays k = len(neighbourhood), self included.
//   K7  fpfh.py:101-116 fpfh[kp] = spfh[kp] + (sum_{j in nbrs(kp), d_j > 0} spfh[j] / d_j) / k_kp.
// Data layout in HBM: the SPFH table is kept as INTEGER bin counts plus the per-point k, by cell-sorted
// position -- uint8 (stored as count ^ 128) when no neighbourhood exceeds 255 points and there are at most
// 128 bins, else uint16, or uint32 when a neighbourhood exceeds 65535.  spfh[j][b] is reconstructed as
// (double)count/k exactly as the reference computed it, but a row costs 128 / 256 B instead of 1000 B in
// the K7 gather (k x row per keypoint).
// Mapping: one wave per point.  K6 bins with per-wave LDS atomics.  K7 on the uint8 table is an exact
// int8 matrix-core contraction (k_fpfh_mc); on the wider tables it streams the neighbour rows through the
// vector ALU, eight bins per lane (k_fpfh).
// HBM roofline, algorithmic bytes (float64 API widths): 48 in + 1000 SPFH write + 1000 SPFH
// read + 1000 FPFH write = 3048 B per descriptor when every point is a keypoint.
#include <algorithm>
#include <cmath>
#include <cstring>

#include "common.h"
#include "device_util.h"

namespace {

// Bin counts above SF_FAST_FPFH_BINS that the fast K6 still serves: the ODD ones whose central alpha bin's n^2 slots fit the
// 128-column byte row -- 9 (81 bins) and 11 (121) -- when the radius pins alpha to that bin (a windowed table, see
// sf_spfh_create_for_radius).  Without a window these counts take the generic kernels.
#define SF_WIN_FPFH_BINS 11
struct fpfh_edges {
    double a[SF_WIN_FPFH_BINS + 1], p[SF_WIN_FPFH_BINS + 1], t[SF_WIN_FPFH_BINS + 1];
    double tan_t[SF_WIN_FPFH_BINS + 1]; // tan of the interior theta edges (index 1..nb-1)
    double p_inv_width;                 // n_bins / (p[n_bins] - p[0]): formed on the host (in the kernel the IEEE division of two
                                        // wave-uniform numbers was a 15-instruction vector sequence per point)
};

// np.histogramdd bin of x: searchsorted(edges, x, 'right') - 1, x == last edge -> last bin, out of
// range / NaN -> -1 (dropped).
__device__ inline int hist_bin(const double *e, int nb, double x)
{
    if (!(x >= e[0]) || x > e[nb]) return -1;
    int b = 0;
#pragma unroll
    for (int i = 1; i < SF_WIN_FPFH_BINS; ++i)
        if (i < nb && x >= e[i]) b = i;
    return b;
}

// Bin of theta = atan2(a, b) over the theta edges WITHOUT evaluating atan2: inside (-pi/2, pi/2) (b > 0)
// theta >= e_i  <=>  a >= tan(e_i) * b.  Whenever a comparison is within a 1e-13 relative band of
// equality, or b is within that band of 0 (theta near +-pi/2, the outer edges), the reference's own
// expression -- atan2 then the histogramdd rule -- decides, so the result is the reference's in all cases.
__device__ inline int theta_bin(const fpfh_edges &ed, int nb, double a, double b)
{
    const double band = 1e-13;
    const double aa = fabs(a);
    if (b > band * aa) {
        int bin = 0;
        double gap = 1.0e300; // smallest |a - tan(e_i) b| over the interior edges
#pragma unroll
        for (int i = 1; i < SF_WIN_FPFH_BINS; ++i)
            if (i < nb) {
                // a - tan(e_i) b with ONE rounding: its sign is that of the exact difference, and whenever that differs
                // from the rounded product's verdict the gap is within an ulp, far inside the band the fallback owns
                const double di = __builtin_fma(-ed.tan_t[i], b, a);
                bin += di >= 0.0 ? 1 : 0;
                gap = fmin(gap, fabs(di));
            }
        // one (conservative) test for all edges: |tan(e_i) b| <= |tan(e_1)| b, the outermost interior edge
        if (gap > band * (aa + fabs(ed.tan_t[1]) * b)) return bin;
    } else if (b < -band * aa) {
        return -1; // |theta| > pi/2: outside the histogram range, dropped (fpfh.py:86)
    }
    return hist_bin(ed.t, nb, atan2(a, b));
}

// The same decision from a CHEAPER form of a.  a = n_j . (u x (c x u)) = (n_j . c) |u|^2 - (n_j . u)(c . u) exactly (the
// triple-product expansion), which needs one dot product and three operations where the two cross products need 23; evaluated
// in floating point it differs from the reference's cross-product evaluation by at most E (the caller's bound on both
// rounding errors together).  Every test of theta_bin is made with that margin added on the safe side, so an answer given
// here is the answer theta_bin gives on the reference's a; -2 = undecided at this precision, the caller evaluates the
// reference's expression and asks theta_bin.
__device__ inline int theta_bin_fast(const fpfh_edges &ed, int nb, double a, double b, double E)
{
    const double band = 1e-13;
    const double aa = fabs(a) + E; // >= |a_ref|
    if (b > band * aa) {
        int bin = 0;
        double gap = 1.0e300;
#pragma unroll
        for (int i = 1; i < SF_WIN_FPFH_BINS; ++i)
            if (i < nb) {
                const double di = __builtin_fma(-ed.tan_t[i], b, a);
                bin += di >= 0.0 ? 1 : 0;
                gap = fmin(gap, fabs(di));
            }
        // gap - E <= the reference's gap; beyond its band with E to spare every sign above is the reference's sign
        return gap > band * (aa + fabs(ed.tan_t[1]) * b) + E ? bin : -2;
    }
    return b < -band * aa ? -1 : -2;
}

// NCH > 0: neighbourhoods of at most 64*NCH points -- every chunk's indices, then every chunk's
// coordinates / normals, are requested before any is used, so a wave pays ONE index round trip and ONE
// gather round trip instead of one per chunk.  NCH == 0: streaming loop for any size.
// NB: the bin count as a compile-time constant (1..8, one instantiation each): the edge comparisons unroll to exactly
// NB - 1 per feature and only the edges in use occupy SGPRs.  With a run-time count every slot of the 4 x 9 edge
// table stays live and the compiler spills SGPRs into VGPR lanes (a v_readlane per comparison: +40 % time).
#ifndef SF_SPFH_WPB
#define SF_SPFH_WPB 2 // waves (= points) per workgroup (0.655 / 0.643 / 0.645 ms at C3 for 4 / 2 / 1)
#endif

// limit / SEL: dispatch by list length, per point (sf_nbrs_dispatch) -- the main launch leaves out the points whose own list
// exceeds its form, a second launch (SEL, the streaming form) serves exactly those.
template <typename CT, int NCH, int NB, bool SEL>
// (waves per SIMD: six for the forms of up to three chunks -- 80 registers; the four-chunk and the streaming form, asked for six,
// spilled 52-140 bytes in their sweep: on the clustered cloud K6 1.09 + 0.34 ms, with five waves and no spill 0.81 + 0.27)
__global__ __launch_bounds__(64 * SF_SPFH_WPB) __attribute__((amdgpu_waves_per_eu((NCH == 0 || NCH >= 4) ? 5 : 6))) void k_spfh(const double *__restrict__ rec,
                                              const int64_t *__restrict__ offset, const int32_t *__restrict__ cnt,
    const int32_t *__restrict__ idx,
                                              int64_t m, int64_t self_begin, fpfh_edges ed, int nb_rt, int nb3, int stride,
                                              CT *__restrict__ counts, int32_t *__restrict__ kout, unsigned bias,
                                              double *__restrict__ p4, double mom_radius, double *__restrict__ cov,
                                              unsigned *__restrict__ live, int alpha_bin, double nrm_max,
                                              uint8_t *__restrict__ packed, int pack_b0, int pack_b1,
                                              uint8_t *__restrict__ hi, int limit, const int32_t *__restrict__ sel,
                                              int64_t nsel, int64_t view_first, int alpha_pair, int win_lo, int win_len)
{
    const int nb = NB > 0 ? NB : nb_rt;
    // (the wave's histogram holds the table's WINDOW of the bins: all n_bins^3 of at most 512 for the unwindowed tables)
    __shared__ unsigned int hist[SF_SPFH_WPB][SF_FAST_FPFH_BINS * SF_FAST_FPFH_BINS * SF_FAST_FPFH_BINS];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int64_t q = sf_uniform64(sf_xcd_block() * SF_SPFH_WPB + wave);
    if (SEL) {
        if (q >= nsel) return;
        q = (int64_t)sf_uniform(sel[q]) - view_first;
        if (q < 0) return;
    }
    if (q >= m) return; // whole wave exits together; no block-wide barrier below
    unsigned int *h = hist[wave];
    for (int b = lane; b < win_len; b += 64) h[b] = 0;
    const int64_t i = self_begin + q; // cell-sorted position of this point
    const int64_t s = offset[q];
    const int k = cnt[q];
#ifndef SF_AB_NOLIMIT
    if (!SEL && sf_uniform(k) > limit) return; // (a point of the second launch)
#endif
    const double px = rec[6 * i + 0], py = rec[6 * i + 1], pz = rec[6 * i + 2];
    const double ux = rec[6 * i + 3], uy = rec[6 * i + 4], uz = rec[6 * i + 5];
    __builtin_amdgcn_wave_barrier();
    const double p_inv_width = ed.p_inv_width; // np.linspace edges: equal widths up to rounding
    // theta_bin_fast's margin: |a_fast - a_reference| <= ~24 eps |n_j| |u|^2 |c| (a dozen roundings on either side, each
    // relative to a product of those norms); 64 eps max|n| |u|^2 per unit of |c| is the bound used
    const double uu = (ux * ux + uy * uy) + uz * uz;
    const double e_per_dist = 1.5e-14 * nrm_max * uu;
    auto pair = [&](double cx, double cy, double cz, double njx, double njy, double njz) {
        const double d2 = (cx * cx + cy * cy) + cz * cz;
        if (d2 > 0.0) { // dist > 0 (fpfh.py:50-57)
            // phi = (c . u) / sqrt(d2) only picks a bin.  One Newton step on v_rsq_f64 gives it to ~1e-15; the
            // reference's own expression (sqrt, then the division: 34 instructions) is evaluated only when that
            // value lies within 1e-9 bin widths of an edge, so the bin is the reference's in every case.
            const double num = (cx * ux + cy * uy) + cz * uz;
            const double y0 = __builtin_amdgcn_rsq(d2);
            const double y1 = __builtin_fma(0.5 * y0, __builtin_fma(-(d2 * y0), y0, 1.0), y0);
            double phi = num * y1;
            const double pos = (phi - ed.p[0]) * p_inv_width;
            if (fabs(pos - rint(pos)) <= 1e-9) phi = num / sqrt(d2);
            const double b = (njx * ux + njy * uy) + njz * uz;
            // alpha = v . n_j with v = c x u NOT normalised (fpfh.py:60): |alpha| <= |c| |u| |n_j| <= radius when no normal
            // is longer than 1, so with the radius below the smallest |edge| of the alpha histogram every sample is in the
            // bin around 0 -- the host passes that bin (alpha_bin >= 0, wave-uniform) and alpha is never formed.  Then v and
            // w = u x v are not needed either unless theta's cheap form cannot decide (theta_bin_fast).
            // (an EVEN bin count has an edge at 0: alpha then falls into one of the TWO central bins, alpha_pair and alpha_pair + 1,
            // and only its side of that one edge has to be found -- the reference's own alpha, compared as searchsorted does)
            // (np.linspace(-1, 1, n + 1) has an edge at 0 exactly when n is even: the pair form exists for even bin counts only, and
            // the odd ones -- 5 bins, the headline -- compile to the code they had before it)
            constexpr bool EVEN = NB > 0 && NB % 2 == 0;
            int ba = alpha_bin, bt = -2;
            if (alpha_bin >= 0 || (EVEN && alpha_pair >= 0)) {
                const double nc = (njx * cx + njy * cy) + njz * cz;
                bt = theta_bin_fast(ed, nb, nc * uu - b * num, b, e_per_dist * (d2 * y1) * 1.01);
            }
            if (!EVEN) {
                if (bt == -2) { // the reference's own expressions (fpfh.py:58-66)
                    const double vx = cy * uz - cz * uy, vy = cz * ux - cx * uz, vz = cx * uy - cy * ux; // cross(c, u)
                    const double wx = uy * vz - uz * vy, wy = uz * vx - ux * vz, wz = ux * vy - uy * vx; // cross(u, v)
                    bt = theta_bin(ed, nb, (njx * wx + njy * wy) + njz * wz, b);
                    if (alpha_bin < 0) ba = hist_bin(ed.a, nb, (vx * njx + vy * njy) + vz * njz);
                }
            } else if (bt == -2 || alpha_bin < 0) {
                const double vx = cy * uz - cz * uy, vy = cz * ux - cx * uz, vz = cx * uy - cy * ux; // cross(c, u)
                if (bt == -2) {
                    const double wx = uy * vz - uz * vy, wy = uz * vx - ux * vz, wz = ux * vy - uy * vx; // cross(u, v)
                    bt = theta_bin(ed, nb, (njx * wx + njy * wy) + njz * wz, b);
                }
                if (alpha_bin < 0) {
                    const double alpha = (vx * njx + vy * njy) + vz * njz;
                    ba = alpha_pair >= 0 ? alpha_pair + (alpha >= ed.a[alpha_pair + 1] ? 1 : 0) : hist_bin(ed.a, nb, alpha);
                }
            }
            const int bp = hist_bin(ed.p, nb, phi);
            if ((ba | bp | bt) >= 0) atomicAdd(&h[(ba * nb + bp) * nb + bt - win_lo], 1u);
        }
    };
    // Optional by-product (cov != NULL): the weighted covariance of the SHOT frame (shot.py:27-35, w = r - ||c||, the
    // point itself included), from the neighbours this wave gathers anyway -- K4 then only has its eigen-solves left.
    double ws = 0, a11 = 0, a21 = 0, a31 = 0, a22 = 0, a32 = 0, a33 = 0;
    auto moments = [&](double cx, double cy, double cz) {
        const double w = mom_radius - sf_sqrt_fast((cx * cx + cy * cy) + cz * cz);
        ws += w;
        const double wx = cx * w, wy = cy * w, wz = cz * w;
        // (multiply-adds: these sums feed an eigen-decomposition, not a bin decision, and the reference forms them in an FMA
        // BLAS -- `(w * centred.T) @ centred`, shot.py:33)
        a11 = __builtin_fma(cx, wx, a11); a21 = __builtin_fma(cy, wx, a21); a31 = __builtin_fma(cz, wx, a31);
        a22 = __builtin_fma(cy, wy, a22); a32 = __builtin_fma(cz, wy, a32); a33 = __builtin_fma(cz, wz, a33);
    };
    if (NCH > 0) {
        constexpr int NC = NCH > 0 ? NCH : 1;
        // (instantiated for the longest list of the launch; a chunk past THIS point's list -- the last one for nine points
        // in ten at C3 -- is skipped wave-uniformly: no index load, no gather)
        const int ku = sf_uniform(k);
        int jj[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int t = c * 64 + lane;
            jj[c] = -1;
            if (c == 0 || c * 64 < ku) jj[c] = t < k ? SF_LIST_LOAD(idx + s + t) : -1;
        }
        double cx[NC], cy[NC], cz[NC], ax[NC], ay[NC], az[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            cx[c] = cy[c] = cz[c] = ax[c] = ay[c] = az[c] = 0.0;
            if (c == 0 || c * 64 < ku) {
                const int j = jj[c] < 0 ? 0 : jj[c];
                sf_load_pn(rec, j, cx[c], cy[c], cz[c], ax[c], ay[c], az[c]);
            }
        }
#pragma unroll
        for (int c = 0; c < NC; ++c)
            if ((c == 0 || c * 64 < ku) && jj[c] >= 0) {
                if (cov) moments(cx[c] - px, cy[c] - py, cz[c] - pz);
                pair(cx[c] - px, cy[c] - py, cz[c] - pz, ax[c], ay[c], az[c]);
            }
    } else {
        for (int t = lane; t < k; t += 64) {
            double x, y, z, a, b, c;
            sf_load_pn(rec, idx[s + t], x, y, z, a, b, c);
            if (cov) moments(x - px, y - py, z - pz);
            pair(x - px, y - py, z - pz, a, b, c);
        }
    }
    if (cov) {
        const double part[8] = {ws, a11, a21, a31, a22, a32, a33, 0.0};
        const double tot = sf_wave_sum8(part); // lanes 8 i .. 8 i + 7 hold the sum of part[i]
        const double wsum = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(tot), 0),
                                             __builtin_amdgcn_readlane(__double2loint(tot), 0));
        const double iw = sf_rcp_fast(wsum);
        const int e = lane >> 3;
        if ((lane & 7) == 0 && e >= 1 && e <= 6) cov[6 * q + e - 1] = tot * iw; // c11 c21 c31 c22 c32 c33
    }
    __builtin_amdgcn_wave_barrier();
    __threadfence_block();
    CT *row = counts + i * (int64_t)stride;
    bool hi_empty = false;
    if (live) { // uint8 table: 128 bins, two per lane; which 16-bin blocks of this row hold a count goes into the table-wide mask
        // (column c of the row = bin win_lo + c; columns past the window are padding: count 0)
        const unsigned v0 = lane < win_len ? h[lane] : 0u, v1 = lane + 64 < win_len ? h[lane + 64] : 0u;
        // (the uint8 table keeps count & 255; a point with more than 255 neighbours also has count >> 8 in the table of high
        // bytes.  Streamed past the L2 -- K7 gathers the PACKED rows; of this table it reads each keypoint's own row, once)
        __builtin_nontemporal_store((CT)(v0 ^ bias), row + lane);
        __builtin_nontemporal_store((CT)(v1 ^ bias), row + lane + 64);
        if (hi && sf_uniform(k) > 255) {
            hi[i * 128 + lane] = (uint8_t)(v0 >> 8);
            hi[i * 128 + lane + 64] = (uint8_t)(v1 >> 8);
            // (a point with more than 255 neighbours whose counts all stay below 256 -- the rule unless its neighbourhood is a
            // smooth surface with consistent normals -- has a row of zero high bytes: marked in its K7 record below, so that no
            // keypoint that has it as a neighbour goes and reads that row)
            hi_empty = __ballot(((v0 | v1) >> 8) != 0u) == 0ull;
        }
        if (packed && lane < 8) { // the host knows which two 16-bin blocks can be live (spfh_compute): the packed copy K7 gathers is
                                  // written here, straight from the LDS histogram, instead of by a kernel of its own re-reading the
                                  // table: lane l takes bins 4 (l & 3) .. + 3 of block (l < 4 ? b0 : b1), eight dwords = the 32-byte row
            const int first = 16 * (lane < 4 ? pack_b0 : pack_b1) + 4 * (lane & 3);
            unsigned w = 0u;
#pragma unroll
            for (int t = 0; t < 4; ++t) w |= (((first + t < win_len ? h[first + t] : 0u) ^ bias) & 0xffu) << (8 * t);
            reinterpret_cast<unsigned *>(packed + i * 32)[lane] = w;
        }
        const unsigned long long n0 = __ballot(v0 != 0u), n1 = __ballot(v1 != 0u);
        unsigned mask = 0u;
#pragma unroll
        for (int kq = 0; kq < 4; ++kq) {
            mask |= ((n0 >> (16 * kq)) & 0xffffull) ? 1u << kq : 0u;
            mask |= ((n1 >> (16 * kq)) & 0xffffull) ? 16u << kq : 0u;
        }
        // (one plain, cacheable read per wave -- a stale value only costs a redundant atomic; the atomic itself only while
        // the table-wide mask is still growing)
        if (lane == 0 && (mask & ~*live)) atomicOr(live, mask);
    } else {
        for (int b = lane; b < stride; b += 64) row[b] = (CT)((b < nb3 ? h[b] : 0u) ^ bias); // (padding bins: count 0; win_lo = 0 here)
    }
    if (lane == 0) {
        kout[i] = k;
        if (p4) { // the per-neighbour record of the matrix-core K7
            double2 *o = reinterpret_cast<double2 *>(p4 + 4 * i);
            o[0] = make_double2(rec[6 * i + 0], rec[6 * i + 1]);
            // k as a double; NEGATIVE for a point with more than 255 neighbours and no high byte set: K7 squares it for the weight
            // and asks "k > 255" to know whether the row of high bytes has anything to add
            o[1] = make_double2(rec[6 * i + 2], hi_empty ? -(double)k : (double)k);
        }
    }
}

// After K6 on the uint8 table: when at most two of the eight 16-bin blocks are live, their two 16-byte chunks of every
// row are copied side by side into `packed` (32 bytes per row).  Rows [begin, end) were just (re)computed: k_spfh_pack.
// The others are re-packed too if the mask they were packed under (live[1]) is not the current one -- the mask only ever
// grows, and a row packed under an older mask may hold a different pair of blocks: k_spfh_repack, a small grid that
// returns at once in the usual case.  k_spfh_pack_done then records the mask.
__device__ inline void spfh_pack_row(const uint8_t *__restrict__ counts, uint8_t *__restrict__ packed, int64_t row, unsigned mask)
{
    const int b0 = mask ? __ffs(mask) - 1 : 0;
    const unsigned rest = mask & (mask - 1u);
    const int b1 = rest ? __ffs(rest) - 1 : (b0 + 1) & 7;
    const uint4 *src = reinterpret_cast<const uint4 *>(counts + row * 128);
    uint4 *dst = reinterpret_cast<uint4 *>(packed + row * 32);
    dst[0] = src[b0];
    dst[1] = src[b1];
}

__global__ __launch_bounds__(256) void k_spfh_pack(const uint8_t *__restrict__ counts, int64_t begin, int64_t end,
                                                   const unsigned *__restrict__ live, uint8_t *__restrict__ packed)
{
    const unsigned mask = live[0] & 0xffu;
    if (__popc(mask) > 2) return;
    const int64_t row = begin + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (row < end) spfh_pack_row(counts, packed, row, mask);
}

__global__ __launch_bounds__(256) void k_spfh_repack(const uint8_t *__restrict__ counts, int64_t n, int64_t begin, int64_t end,
                                                     const unsigned *__restrict__ live, uint8_t *__restrict__ packed)
{
    const unsigned mask = live[0] & 0xffu, sig = live[1];
    if (__popc(mask) > 2 || sig == mask) return;
    for (int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; row < n; row += (int64_t)gridDim.x * blockDim.x)
        if (row < begin || row >= end) spfh_pack_row(counts, packed, row, mask);
}

// det != 0: the blocks K6 can possibly touch are known on the host (alpha's bin is pinned, see spfh_compute); they are
// marked live BEFORE K6 runs, so the mask after K6 is a function of the call's parameters alone -- the same on every
// rank of a sharded job, and known to the host without a read-back.
__global__ void k_spfh_live_or(unsigned *__restrict__ live, unsigned det) { live[0] |= det; }

__global__ void k_spfh_pack_done(unsigned *__restrict__ live)
{
    const unsigned mask = live[0] & 0xffu;
    live[1] = __popc(mask) <= 2 ? mask : ~0u;
}

template <typename CT>
__global__ void k_spfh_export(const CT *__restrict__ counts, const int32_t *__restrict__ kk,
                              const int32_t *__restrict__ perm, int64_t n, int nb3, int stride, unsigned bias,
                              double *__restrict__ out, const uint8_t *__restrict__ hi, int win_lo, int win_len)
{
    int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n * nb3) return;
    int64_t i = g / nb3;
    int b = (int)(g - i * nb3);
    const int col = b - win_lo; // (a bin outside the table's window: structurally empty)
    unsigned cnt = 0u;
    if (col >= 0 && col < win_len) {
        cnt = (unsigned)counts[i * stride + col] ^ bias;
        if (hi && kk[i] > 255) cnt += 256u * (unsigned)hi[i * stride + col]; // (byte table: a long point's high bytes)
    }
    out[(int64_t)perm[i] * nb3 + b] = (double)cnt / (double)kk[i];
}

// ---- any bin count (n_bins > SF_FAST_FPFH_BINS): the reference takes whatever `n_bins` it is given (fpfh.py:16) ---------
// K6g: one wave per point, edges from memory with a binary search (np.histogramdd's searchsorted rule), theta from
// atan2 as the reference computes it; the row of the (uint32) table is private to the wave, so the counts go straight
// into it with global atomics (the row is zeroed first).  K7g: one workgroup per keypoint, neighbours staged through
// LDS in tiles (index, 1 / d_j, k_j), every thread owns bins tid, tid + 256, ... and sums the neighbours in list
// order.  Plain and bandwidth-hungry on purpose: n_bins^3 bins per point leave no room for the LDS / matrix-core
// schemes above, and the configurations the pipeline uses (n_bins <= 8) never come here.
__device__ inline int hist_bin_search(const double *__restrict__ e, int nb, double x)
{
    if (!(x >= e[0]) || x > e[nb]) return -1;
    if (x == e[nb]) return nb - 1;
    int lo = 0, hi = nb + 1; // first index with e[i] > x
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (e[mid] <= x) lo = mid + 1; else hi = mid;
    }
    return lo - 1;
}

__global__ __launch_bounds__(256) void k_spfh_generic(const double *__restrict__ rec, const int64_t *__restrict__ offset,
                                                      const int32_t *__restrict__ cnt, const int32_t *__restrict__ idx, int64_t m,
                                                      int64_t self_begin, const double *__restrict__ edges, int nb, int nb3,
                                                      int stride, unsigned *__restrict__ counts, int32_t *__restrict__ kk)
{
    const int lane = threadIdx.x & 63;
    const int64_t q = sf_uniform64(sf_xcd_block() * 4 + (threadIdx.x >> 6));
    if (q >= m) return;
    const int64_t i = self_begin + q, s = offset[q];
    const int k = cnt[q];
    unsigned *row = counts + i * (int64_t)stride;
    for (int b = lane; b < stride; b += 64) row[b] = 0u;
    if (lane == 0) kk[i] = k;
    __threadfence(); // the zeroed row is in memory before any lane's atomic reaches it
    double px, py, pz, ux, uy, uz;
    sf_load_pn(rec, (int)i, px, py, pz, ux, uy, uz);
    const double *ea = edges, *ep = edges + (nb + 1), *et = edges + 2 * (nb + 1);
    for (int t = lane; t < k; t += 64) {
        double x, y, z, nx, ny, nz;
        sf_load_pn(rec, idx[s + t], x, y, z, nx, ny, nz);
        const double cx = x - px, cy = y - py, cz = z - pz;
        const double dist = sqrt((cx * cx + cy * cy) + cz * cz); // fpfh.py:48
        if (!(dist > 0.0)) continue;
        const double vx = cy * uz - cz * uy, vy = cz * ux - cx * uz, vz = cx * uy - cy * ux; // cross(c, u)  :50
        const double wx = uy * vz - uz * vy, wy = uz * vx - ux * vz, wz = ux * vy - uy * vx; // cross(u, v)  :51
        const double alpha = (vx * nx + vy * ny) + vz * nz;                                   // :52
        const double phi = ((cx * ux + cy * uy) + cz * uz) / dist;                            // :53
        const double theta = atan2((nx * wx + ny * wy) + nz * wz, (nx * ux + ny * uy) + nz * uz); // :54-57
        const int ba = hist_bin_search(ea, nb, alpha), bp = hist_bin_search(ep, nb, phi), bt = hist_bin_search(et, nb, theta);
        if (ba < 0 || bp < 0 || bt < 0) continue;
        atomicAdd(&row[(ba * nb + bp) * nb + bt], 1u);
    }
}
} // namespace


// The alpha bins a radius can reach: |alpha| <= |c| |u| |n_j| <= radius * max|n|^2 (alpha = (c x u) . n_j with v NOT normalised,
// fpfh.py:60).  first .. last: the bins of -reach and +reach under np.histogramdd's rule; false when the reach leaves the histogram.
static bool alpha_bins_within_reach(const double *edges_a, int nb, double reach, int *first, int *last)
{
    if (!(reach >= 0.0) || !std::isfinite(reach) || !(-reach > edges_a[0]) || !(reach < edges_a[nb])) return false;
    int lo = 0, hi = 0;
    for (int i = 1; i < nb; ++i) { // searchsorted(edges, x, 'right') - 1
        if (edges_a[i] <= -reach) lo = i;
        if (edges_a[i] <= reach) hi = i;
    }
    *first = lo;
    *last = hi;
    return true;
}

static sf_spfh *spfh_create(sf_ctx *ctx, sf_cloud *c, int n_bins, int64_t max_count, double radius);

extern "C" sf_spfh *sf_spfh_create(sf_ctx *ctx, sf_cloud *c, int n_bins, int64_t max_count)
{
    return spfh_create(ctx, c, n_bins, max_count, 0.0);
}

// The table for a KNOWN search radius.  With more than 128 bins (n_bins 6, 7, 8) a row no longer fits the 128-byte row of the
// matrix-core K7 -- unless most of it is structurally empty: when radius * max|n|^2 stays inside the one or two central bins of
// the alpha histogram (an even bin count has an edge at 0), only those bins' n_bins^2 (2 n_bins^2) slots can ever receive a
// count: 72 of 216, 49 of 343, 128 of 512.  The table then keeps exactly that WINDOW of bins, one byte each, and everything
// downstream -- K6's row, the packed rows, the high bytes, K7 on the matrix cores, the exchange's wire image -- is the 5-bin
// path; K7 writes zeros for the bins outside the window.  (Until round 5 these bin counts took a 16-bit table and the
// vector K7: 3.5 ms per 1M keypoints at 6 bins, 7.3 ms at 8, against 0.85 ms at 5.)  sf_spfh_compute checks that the radius it
// is then run with keeps alpha inside the window.
extern "C" sf_spfh *sf_spfh_create_for_radius(sf_ctx *ctx, sf_cloud *c, int n_bins, int64_t max_count, double radius)
{
    return spfh_create(ctx, c, n_bins, max_count, radius);
}

extern "C" int sf_spfh_elem_bytes(const sf_spfh *sp) { return sp ? sp->elem_bytes : 0; }

static sf_spfh *spfh_create(sf_ctx *ctx, sf_cloud *c, int n_bins, int64_t max_count, double radius)
{
    if (!ctx || !c) { sf_set_error("sf_spfh_create: null argument"); return nullptr; }
    if (n_bins < 1 || n_bins > SF_MAX_FPFH_BINS) {
        sf_set_error("sf_spfh_create: n_bins=%d outside 1..%d (n_bins^3 bins per point)", n_bins, SF_MAX_FPFH_BINS);
        return nullptr;
    }
    if (hipSetDevice(ctx->device) != hipSuccess) { sf_set_error("hipSetDevice failed"); return nullptr; }
    sf_spfh *sp = new sf_spfh();
    sp->n = c->n;
    sp->n_bins = n_bins;
    sp->nb3 = n_bins * n_bins * n_bins;
    // neighbourhoods of at most 255 points and at most 128 bins: one BYTE per bin, biased by 128 (a 128-byte row the
    // matrix-core K7 consumes as int8); else uint16, uint32 beyond 65535
    // (lists longer than 255 points: the byte table keeps count & 255 and the long points' rows get a table of high bytes
    // beside it -- one long list does not move every point's row to 16 bits and every keypoint to the vector K7)
    sp->win_lo = 0;
    sp->win_len = sp->nb3;
    bool window = false;
    if (sp->nb3 > 128 && n_bins <= SF_WIN_FPFH_BINS && max_count <= 65535 && radius > 0.0 && c->nrm_orig && !getenv("SF_FPFH_NO_WINDOW")) {
        double n2 = 0.0, ea[SF_WIN_FPFH_BINS + 1];
        if (sf_cloud_normals_max2(ctx, c, &n2) != SF_OK) { delete sp; return nullptr; }
        for (int i = 0; i <= n_bins; ++i) ea[i] = -1.0 + 2.0 * (double)i / (double)n_bins; // (np.linspace(-1, 1, n + 1) up to an ulp:
        int a0 = 0, a1 = 0;                                                                 // sf_spfh_compute checks the real edges)
        if (alpha_bins_within_reach(ea, n_bins, radius * n2 * (1.0 + 1e-6), &a0, &a1) && (a1 - a0 + 1) * n_bins * n_bins <= 128) {
            window = true;
            sp->win_lo = a0 * n_bins * n_bins;
            sp->win_len = (a1 - a0 + 1) * n_bins * n_bins;
        }
    }
    sp->elem_bytes = (max_count <= 65535 && (sp->nb3 <= 128 || window)) ? 1 : (max_count > 65535 ? 4 : 2);
    if (n_bins > SF_FAST_FPFH_BINS && !window) sp->elem_bytes = 4; // the generic kernels keep 32-bit counts
    sp->bias = sp->elem_bytes == 1 ? 128 : 0;
    // rows padded to a multiple of 128 elements: lane l of a wave owns elements 2l, 2l+1 of each 128-element
    // slice, so no lane of the K7 row loads ever falls outside its row (256 B rows for 125 uint16 bins)
    sp->stride = 128; // ... and to a power of two, so that a row is 256 B, 512 B, 1 KiB or 2 KiB (the K7 row shapes)
    while (sp->stride < (window ? sp->win_len : sp->nb3)) sp->stride *= 2;
    if (n_bins > SF_FAST_FPFH_BINS && !window) sp->stride = (sp->nb3 + 3) & ~3; // generic kernels: no shape constraint
    // room for ceil(n / nranks) rows per rank so the table can be all-gathered in place
    const int64_t nr = ctx->nranks > 0 ? ctx->nranks : 1;
    sp->rows_alloc = std::max<int64_t>(sf_div_up(c->n, nr) * nr, 1);
    size_t nn = (size_t)sp->rows_alloc;
    if ((double)nn * sp->stride * sp->elem_bytes > 2.0e11) {
        sf_set_error("sf_spfh_create: an SPFH table of %lld x %d bins does not fit the device", (long long)nn, sp->nb3);
        delete sp;
        return nullptr;
    }
    // (every block from the context's stream-ordered pool: a drop-in call creates a table and frees it again)
    const bool bytes_tab = sp->elem_bytes == 1;
    bool ok = sf_pool_alloc(ctx, nn * sp->stride * sp->elem_bytes, &sp->counts) == SF_OK && sf_pool_alloc(ctx, nn * sizeof(int32_t), (void **)&sp->k) == SF_OK;
    if (ok && bytes_tab) ok = sf_pool_alloc(ctx, nn * 4 * sizeof(double), (void **)&sp->p4) == SF_OK;
    if (ok && bytes_tab && max_count > 255) ok = sf_pool_alloc(ctx, nn * 128, (void **)&sp->hi) == SF_OK;
    if (ok && bytes_tab)
        ok = sf_pool_alloc(ctx, 4 * sizeof(unsigned), (void **)&sp->live) == SF_OK && sf_pool_alloc(ctx, nn * 32, (void **)&sp->packed) == SF_OK &&
             hipMemsetAsync(sp->live, 0, 4 * sizeof(unsigned), ctx->stream) == hipSuccess &&
             // SF_FPFH_DENSE=1: every block counts as live from the start (K7 always takes its full form)
             hipMemsetAsync(sp->live, getenv("SF_FPFH_DENSE") ? 0xff : 0, sizeof(unsigned), ctx->stream) == hipSuccess &&
             hipMemsetAsync(sp->live + 1, 0xff, sizeof(unsigned), ctx->stream) == hipSuccess;
    if (!ok) {
        sf_set_error("sf_spfh_create: out of device memory");
        sf_spfh_free(ctx, sp);
        return nullptr;
    }
    // the host's mirror of live[0..1]: exact as long as every mask that went in was known here (mask_known)
    sp->host_live[0] = getenv("SF_FPFH_DENSE") ? ~0u : 0u;
    sp->host_live[1] = ~0u;
    sp->host_live_valid = sp->mask_known = sp->elem_bytes == 1;
    return sp;
}

extern "C" void sf_spfh_free(sf_ctx *ctx, sf_spfh *sp)
{
    if (!sp) return;
    if (ctx) (void)hipStreamSynchronize(ctx->stream);
    for (void *p : {(void *)sp->counts, (void *)sp->k, (void *)sp->p4, (void *)sp->live, (void *)sp->packed, (void *)sp->hi})
        if (p) { if (ctx) sf_pool_release(ctx, p); else (void)hipFree(p); }
    delete sp;
}

static int spfh_compute(sf_ctx *ctx, sf_cloud *c, sf_nbrs *nb, sf_spfh *sp, const double *edges, double *cov)
{
    if (!ctx || !c || !nb || !sp || !edges) { sf_set_error("sf_spfh_compute: null argument"); return SF_ERR_ARG; }
    if (!nb->self) { sf_set_error("sf_spfh_compute: needs a sf_radius_search_self result"); return SF_ERR_ARG; }
    SF_CHECK(sf_nbrs_on_grid(nb, c, "sf_spfh_compute"));
    if (sp->n != c->n || nb->self_begin + nb->m > sp->n) { sf_set_error("sf_spfh_compute: table/cloud size mismatch"); return SF_ERR_ARG; }
    if (sp->elem_bytes == 1 && nb->max_count > 255 && !sp->hi) {
        sf_set_error("sf_spfh_compute: neighbourhood of %lld points needs a wider table (pass max_count to sf_spfh_create)",
                     (long long)nb->max_count);
        return SF_ERR_ARG;
    }
    if (sp->elem_bytes <= 2 && nb->max_count > 65535) {
        sf_set_error("sf_spfh_compute: neighbourhood of %lld points needs a 32-bit table (pass max_count to sf_spfh_create)",
                     (long long)nb->max_count);
        return SF_ERR_ARG;
    }
    SF_HIP(hipSetDevice(ctx->device));
    SF_CHECK(sf_cloud_ensure_sorted_normals(ctx, c));
    if (sp->n_bins > SF_FAST_FPFH_BINS && sp->elem_bytes != 1) {
        if (cov) { sf_set_error("sf_spfh_compute_moments: n_bins=%d has no shared-sweep form (use sf_spfh_compute)", sp->n_bins); return SF_ERR_UNSUPPORTED; }
        const int64_t mg = nb->m;
        if (!mg) return SF_OK;
        sf_pool_guard tmp(ctx);
        double *dedges = nullptr;
        const size_t ne = (size_t)3 * (sp->n_bins + 1);
        SF_CHECK(tmp.alloc(&dedges, ne));
        SF_HIP(hipMemcpyAsync(dedges, edges, ne * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
        SF_LAUNCH(ctx, "k6_spfh", k_spfh_generic, dim3(sf_xcd_grid(sf_div_up(mg, 4))), dim3(256), c->rec, nb->offset, nb->count,
                  nb->idx, mg, nb->self_begin, (const double *)dedges, sp->n_bins, sp->nb3, sp->stride, (unsigned *)sp->counts, sp->k);
        SF_HIP(hipStreamSynchronize(ctx->stream)); // `edges` is a host buffer
        return SF_OK;
    }
    fpfh_edges ed;
    const int nbn = sp->n_bins;
    for (int i = 0; i <= SF_WIN_FPFH_BINS; ++i) {
        ed.a[i] = edges[i <= nbn ? i : nbn];
        ed.p[i] = edges[(nbn + 1) + (i <= nbn ? i : nbn)];
        ed.t[i] = edges[2 * (nbn + 1) + (i <= nbn ? i : nbn)];
        ed.tan_t[i] = std::tan(ed.t[i]);
    }
    ed.p_inv_width = (double)nbn / (ed.p[nbn] - ed.p[0]);
    const int64_t m = nb->m;
    if (!m) return SF_OK;
    // alpha's bin is known beforehand when every |alpha| <= radius * max|n|^2 stays clear of the histogram's edges (see
    // k_spfh): the bin that holds 0, if 0 is strictly inside one
    int alpha_bin = -1, alpha_pair = -1; // alpha pinned to ONE bin / to one of TWO adjacent bins (an even count: an edge at 0)
    double nrm_max = 1.0;
    {
        double n2 = 0.0;
        SF_CHECK(sf_cloud_normals_max2(ctx, c, &n2));
        nrm_max = std::sqrt(n2) * (1.0 + 1e-12);
        const double reach = nb->radius * n2 * (1.0 + 1e-9); // |u| |n_j| <= max |n|^2
        double nearest = INFINITY;
        for (int i = 0; i <= nbn; ++i) nearest = std::min(nearest, std::fabs(ed.a[i]));
        if (std::isfinite(reach) && reach < nearest)
            for (int i = 0; i < nbn; ++i)
                if (ed.a[i] < 0.0 && 0.0 < ed.a[i + 1]) alpha_bin = i;
        int a0 = 0, a1 = 0;
        const bool within = alpha_bins_within_reach(ed.a, nbn, reach, &a0, &a1);
        if (alpha_bin < 0 && within && a1 == a0 + 1) alpha_pair = a0;
        if (getenv("SF_FPFH_NO_ALPHA_SHORTCUT")) alpha_bin = alpha_pair = -1;
        if (sp->elem_bytes == 1 && sp->win_len != sp->nb3) {
            // a table that keeps a WINDOW of the bins (sf_spfh_create_for_radius): every alpha this radius can produce must
            // fall into it, or counts would be lost without a trace
            if (!within || a0 * nbn * nbn < sp->win_lo || (a1 + 1) * nbn * nbn > sp->win_lo + sp->win_len) {
                sf_set_error("sf_spfh_compute: the table keeps bins %d .. %d of %d (created for a smaller radius); a search radius of %g "
                             "reaches other alpha bins -- create the table for this radius", sp->win_lo, sp->win_lo + sp->win_len - 1,
                             sp->nb3, nb->radius);
                return SF_ERR_STATE;
            }
        }
    }
    const dim3 grid(sf_xcd_grid(sf_div_up(m, SF_SPFH_WPB))), block(64 * SF_SPFH_WPB);
    const sf_dispatch dsp = sf_nbrs_dispatch(nb);
    const dim3 grid_tail(sf_xcd_grid(sf_div_up(dsp.n_tail > 0 ? dsp.n_tail : 1, SF_SPFH_WPB)));
    const dim3 grid_mid(sf_xcd_grid(sf_div_up(dsp.n_mid > 0 ? dsp.n_mid : 1, SF_SPFH_WPB)));
    uint8_t *const hi_rows = sp->elem_bytes == 1 ? sp->hi : nullptr;
#define SF_SPFH_NB(NAME, GRID, CT, NCH, NB, SEL, SELP, NSEL)                                                            \
    SF_LAUNCH(ctx, NAME, (k_spfh<CT, NCH, NB, SEL>), GRID, block, c->rec, nb->offset, nb->count, nb->idx, m,            \
              nb->self_begin, ed, nbn, sp->nb3, sp->stride, (CT *)sp->counts, sp->k, (unsigned)sp->bias, sp->p4, nb->radius, cov, \
              sizeof(CT) == 1 ? sp->live : (unsigned *)nullptr, alpha_bin, nrm_max, fused_packed, fused_b0, fused_b1,   \
              hi_rows, dsp.limit, SELP, NSEL, dsp.view_first, alpha_pair, sizeof(CT) == 1 ? sp->win_lo : 0,                \
              sizeof(CT) == 1 ? sp->win_len : sp->nb3)
    // (9 and 11 bins: byte tables with a window only -- instantiated for that element type alone: WIDE = 1)
#define SF_SPFH_WIDE_1(NAME, GRID, CT, NCH, SEL, SELP, NSEL)                                                            \
    if (nbn == 9) { SF_SPFH_NB(NAME, GRID, CT, NCH, 9, SEL, SELP, NSEL); }                                              \
    else if (nbn == 11) { SF_SPFH_NB(NAME, GRID, CT, NCH, 11, SEL, SELP, NSEL); }                                       \
    else { sf_set_error("sf_spfh_compute: no fast kernel for %d bins", nbn); return SF_ERR_UNSUPPORTED; }
#define SF_SPFH_WIDE_0(NAME, GRID, CT, NCH, SEL, SELP, NSEL)                                                            \
    { sf_set_error("sf_spfh_compute: no fast kernel for %d bins on this table", nbn); return SF_ERR_UNSUPPORTED; }
#define SF_SPFH_LAUNCH(NAME, GRID, CT, NCH, SEL, SELP, NSEL, WIDE)                                                            \
    switch (nbn) { /* every supported bin count gets its own instantiation: no spilled edge table */                   \
    case 1: { SF_SPFH_NB(NAME, GRID, CT, NCH, 1, SEL, SELP, NSEL); } break;                                            \
    case 2: { SF_SPFH_NB(NAME, GRID, CT, NCH, 2, SEL, SELP, NSEL); } break;                                            \
    case 3: { SF_SPFH_NB(NAME, GRID, CT, NCH, 3, SEL, SELP, NSEL); } break;                                            \
    case 4: { SF_SPFH_NB(NAME, GRID, CT, NCH, 4, SEL, SELP, NSEL); } break;                                            \
    case 5: { SF_SPFH_NB(NAME, GRID, CT, NCH, 5, SEL, SELP, NSEL); } break;                                            \
    case 6: { SF_SPFH_NB(NAME, GRID, CT, NCH, 6, SEL, SELP, NSEL); } break;                                            \
    case 7: { SF_SPFH_NB(NAME, GRID, CT, NCH, 7, SEL, SELP, NSEL); } break;                                            \
    case 8: { SF_SPFH_NB(NAME, GRID, CT, NCH, 8, SEL, SELP, NSEL); } break;                                            \
    default: { SF_SPFH_WIDE_##WIDE(NAME, GRID, CT, NCH, SEL, SELP, NSEL); } break;                                       \
    }
    // the main launch in the register-cached form the bulk of the lists calls for; the few lists that need more chunks in the
    // 4-chunk instantiation of the same form; the points with more than 255 neighbours in the streaming form
#define SF_SPFH_DISPATCH(CT, WIDE)                                                                        \
    switch (dsp.chunks) {                                                                           \
    case 1: { SF_SPFH_LAUNCH("k6_spfh", grid, CT, 1, false, (const int32_t *)nullptr, (int64_t)0, WIDE); } break;                \
    case 2: { SF_SPFH_LAUNCH("k6_spfh", grid, CT, 2, false, (const int32_t *)nullptr, (int64_t)0, WIDE); } break;                \
    case 3: { SF_SPFH_LAUNCH("k6_spfh", grid, CT, 3, false, (const int32_t *)nullptr, (int64_t)0, WIDE); } break;                \
    case 4: { SF_SPFH_LAUNCH("k6_spfh", grid, CT, 4, false, (const int32_t *)nullptr, (int64_t)0, WIDE); } break;                \
    default: { SF_SPFH_LAUNCH("k6_spfh", grid, CT, 0, false, (const int32_t *)nullptr, (int64_t)0, WIDE); } break;               \
    }                                                                                               \
    if (dsp.n_mid) { SF_SPFH_LAUNCH("k6_spfh_mid", grid_mid, CT, 4, true, dsp.mid_sel, dsp.n_mid, WIDE); } \
    if (dsp.n_tail) { SF_SPFH_LAUNCH("k6_spfh_tail", grid_tail, CT, 0, true, dsp.tail_sel, dsp.n_tail, WIDE); }
    uint8_t *fused_packed = nullptr;
    int fused_b0 = -1, fused_b1 = -1;
    if (sp->elem_bytes == 1) {
        // alpha pinned to one bin => only the 16-bin blocks that hold that bin's n_bins^2 slots can receive a count
        // (blocks of table COLUMNS: bin b sits in column b - win_lo)
        unsigned det = 0u;
        const int pin0 = alpha_bin >= 0 ? alpha_bin : alpha_pair, pin1 = alpha_bin >= 0 ? alpha_bin : alpha_pair + 1;
        if (pin0 >= 0)
            for (int blk = (pin0 * nbn * nbn - sp->win_lo) / 16; blk <= ((pin1 + 1) * nbn * nbn - 1 - sp->win_lo) / 16; ++blk) det |= 1u << blk;
        // Steady state of a resident table (every pass after the first with the same parameters): the device's mask already
        // holds `det`, it names at most two blocks and the packed copy was written under it -- and the host KNOWS all that
        // (host_live mirrors live[] exactly while mask_known).  Then K6 writes the packed rows itself and none of the four
        // little kernels around it (mask OR, pack, repack check, signature) has anything left to do.
        const unsigned m8 = sp->host_live[0] & 0xffu;
        if (det && sp->mask_known && sp->host_live_valid && (m8 | det) == m8 && __builtin_popcount(m8) <= 2 && sp->host_live[1] == m8) {
            fused_packed = sp->packed;
            fused_b0 = m8 ? __builtin_ffs((int)m8) - 1 : 0;
            const unsigned rest = m8 & (m8 - 1u);
            fused_b1 = rest ? __builtin_ffs((int)rest) - 1 : (fused_b0 + 1) & 7; // (same pairing as spfh_pack_row)
            SF_SPFH_DISPATCH(uint8_t, 1)
            return SF_OK;
        }
        if (det) SF_LAUNCH(ctx, "k6_spfh_pack", k_spfh_live_or, dim3(1), dim3(1), sp->live, det);
        SF_SPFH_DISPATCH(uint8_t, 1)
        // rows [self_begin, self_begin + m) are new: pack their live blocks (a no-op on the device when more than two are)
        SF_LAUNCH(ctx, "k6_spfh_pack", k_spfh_pack, dim3((unsigned)sf_div_up(m, 256)), dim3(256), (const uint8_t *)sp->counts,
                  nb->self_begin, nb->self_begin + m, (const unsigned *)sp->live, sp->packed);
        SF_LAUNCH(ctx, "k6_spfh_pack", k_spfh_repack, dim3(2048), dim3(256), (const uint8_t *)sp->counts, sp->n, nb->self_begin,
                  nb->self_begin + m, (const unsigned *)sp->live, sp->packed);
        SF_LAUNCH(ctx, "k6_spfh_pack", k_spfh_pack_done, dim3(1), dim3(1), sp->live);
        if (det && sp->mask_known) { // the device's mask words, without asking the device
            sp->host_live[0] |= det;
            const unsigned m8 = sp->host_live[0] & 0xffu;
            sp->host_live[1] = __builtin_popcount(m8) <= 2 ? m8 : ~0u;
            sp->host_live_valid = true;
        } else { // data decides which blocks are live: sf_fpfh reads the mask back
            sp->mask_known = false;
            sp->host_live_valid = false;
        }
    } else if (sp->elem_bytes == 2) {
        SF_SPFH_DISPATCH(uint16_t, 0)
    } else {
        SF_SPFH_DISPATCH(uint32_t, 0)
    }
#undef SF_SPFH_DISPATCH
#undef SF_SPFH_WIDE_1
#undef SF_SPFH_WIDE_0
#undef SF_SPFH_LAUNCH
#undef SF_SPFH_NB
    return SF_OK;
}

extern "C" int sf_spfh_compute(sf_ctx *ctx, sf_cloud *c, sf_nbrs *nb, sf_spfh *sp, const double *edges)
{
    return spfh_compute(ctx, c, nb, sp, edges, nullptr);
}

// SPFH of every query of `nb` AND, from the same sweep over the neighbours, the weighted covariance of the SHOT
// local frame (6 doubles per query: c11 c21 c31 c22 c32 c33, device memory) for sf_shot_from_moments.
extern "C" int sf_spfh_compute_moments(sf_ctx *ctx, sf_cloud *c, sf_nbrs *nb, sf_spfh *sp, const double *edges,
                                       double *cov_dev)
{
    if (!cov_dev) { sf_set_error("sf_spfh_compute_moments: null cov"); return SF_ERR_ARG; }
    return spfh_compute(ctx, c, nb, sp, edges, cov_dev);
}

extern "C" int sf_spfh_allgather(sf_ctx *ctx, sf_spfh *sp, int64_t rows_per_rank)
{
    if (!ctx || !sp) { sf_set_error("sf_spfh_allgather: null argument"); return SF_ERR_ARG; }
    if (ctx->nranks == 1 && !ctx->comm) return SF_OK;
    if (rows_per_rank <= 0 || rows_per_rank * ctx->nranks > sp->rows_alloc || rows_per_rank * ctx->nranks < sp->n) {
        sf_set_error("sf_spfh_allgather: %lld rows/rank x %d ranks does not tile a table of %lld (+pad %lld) rows",
                     (long long)rows_per_rank, ctx->nranks, (long long)sp->n, (long long)sp->rows_alloc);
        return SF_ERR_ARG;
    }
    const size_t row_bytes = (size_t)sp->stride * sp->elem_bytes;
    char *base = (char *)sp->counts;
    if (ctx->comm && ctx->nranks > 1) {
        // Every rank must hold the same storage (element width, high-byte rows, packed rows): the number and size of the
        // collectives below follow from it, and ranks that disagree would hang or scramble the table.  One 8-byte
        // all-reduce(max) of (format, -format) in front of the gathers; min != max fails HERE, on every rank alike.
        sf_pool_guard tmp(ctx);
        int *fw = nullptr;
        SF_CHECK(tmp.alloc(&fw, 2));
        const int word = (int)sp->elem_bytes | (sp->hi ? 1 << 8 : 0) | (sp->p4 ? 1 << 9 : 0) | ((int)sp->stride << 12);
        void *pin = nullptr;
        SF_CHECK(sf_ctx_pinned(ctx, &pin));
        int *hw = (int *)((char *)pin + SF_PINNED_BYTES - 16); // (the block's last words: nothing else lives there)
        hw[0] = word;
        hw[1] = -word;
        SF_HIP(hipMemcpyAsync(fw, hw, 2 * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
        SF_CHECK(sf_comm_allreduce_max_i32(ctx, fw, fw, 2));
        SF_HIP(hipMemcpyAsync(hw, fw, 2 * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
        SF_HIP(hipStreamSynchronize(ctx->stream));
        if (hw[0] != word || hw[1] != -word) {
            sf_set_error("sf_spfh_allgather: the ranks hold SPFH tables of different storage (this rank: %d-byte counts%s%s, stride %d; "
                         "format words over the ranks %#x .. %#x) -- size every rank's table by the longest list of ANY rank",
                         sp->elem_bytes, sp->hi ? " + high-byte rows" : "", sp->p4 ? " + packed rows" : "", (int)sp->stride,
                         (unsigned)-hw[1], (unsigned)hw[0]);
            return SF_ERR_STATE;
        }
    }
    SF_CHECK(sf_comm_allgather(ctx, base + (size_t)ctx->rank * rows_per_rank * row_bytes, base,
                               (size_t)rows_per_rank * row_bytes));
    char *kb = (char *)sp->k;
    SF_CHECK(sf_comm_allgather(ctx, kb + (size_t)ctx->rank * rows_per_rank * sizeof(int32_t), kb,
                               (size_t)rows_per_rank * sizeof(int32_t)));
    if (sp->p4) {
        char *pb = (char *)sp->p4;
        SF_CHECK(sf_comm_allgather(ctx, pb + (size_t)ctx->rank * rows_per_rank * 32, pb, (size_t)rows_per_rank * 32));
    }
    if (sp->hi) {
        char *hb = (char *)sp->hi;
        SF_CHECK(sf_comm_allgather(ctx, hb + (size_t)ctx->rank * rows_per_rank * 128, hb, (size_t)rows_per_rank * 128));
    }
    // the gathered rows come from other ranks' K6: every block of the table counts as live from here on
    if (sp->live) {
        SF_HIP(hipMemsetAsync(sp->live, 0xff, 2 * sizeof(unsigned), ctx->stream));
        sp->host_live[0] = sp->host_live[1] = ~0u;
        sp->host_live_valid = sp->mask_known = true; // (every block: nothing left for the data to decide)
    }
    return SF_OK;
}

// The two arrays that make up the wire image of a table row (see sf_spfh_exchange_rows), decided from the table's storage and
// the HOST-known block mask alone.  A byte table whose mask the data decided is switched to "every block live" first.
struct spfh_part { char *base; size_t row; };
#define SF_WIRE_PARTS 3 // (unused parts have row == 0)
static int spfh_wire_parts(sf_ctx *ctx, sf_spfh *sp, spfh_part parts[SF_WIRE_PARTS])
{
    parts[2] = spfh_part{nullptr, 0};
    if (sp->elem_bytes == 1) {
        const unsigned m8 = sp->host_live[0] & 0xffu;
        const bool sparse = sp->mask_known && sp->host_live_valid && __builtin_popcount(m8) <= 2 && sp->host_live[1] == m8;
        if (!sparse) { // rows from other ranks' K6 under masks this rank cannot know: every block counts as live
            // (on the MAIN stream whatever the current one is: the exchange itself may be running on the side stream, and the
            // K7 launches that must see the new mask are queued on the main stream after this call returns)
            SF_HIP(hipMemsetAsync(sp->live, 0xff, 2 * sizeof(unsigned), ctx->streams[0]));
            sp->host_live[0] = sp->host_live[1] = ~0u;
            sp->host_live_valid = sp->mask_known = true;
        }
        parts[0] = sparse ? spfh_part{(char *)sp->packed, 32} : spfh_part{(char *)sp->counts, 128};
        parts[1] = spfh_part{(char *)sp->p4, 32};
        // a table with long lists (every rank sizes its table by the longest list of ANY rank): the high bytes travel too
        if (sp->hi) parts[2] = spfh_part{(char *)sp->hi, 128};
    } else {
        parts[0] = spfh_part{(char *)sp->counts, (size_t)sp->stride * sp->elem_bytes};
        parts[1] = spfh_part{(char *)sp->k, sizeof(int32_t)};
    }
    return SF_OK;
}

// The wire image of rows [begin, end) in host memory -- for transports other than RCCL (and for tests): write_back = 0
// copies the image out of the table, 1 copies it into the table's rows.  *bytes (nullable) = size of the image.
extern "C" int sf_spfh_rows_image(sf_ctx *ctx, sf_spfh *sp, int64_t begin, int64_t end, void *host, size_t cap, int write_back,
                                  size_t *bytes)
{
    if (!ctx || !sp || begin < 0 || begin > end || end > sp->n) { sf_set_error("sf_spfh_rows_image: bad argument"); return SF_ERR_ARG; }
    SF_HIP(hipSetDevice(ctx->device));
    spfh_part parts[SF_WIRE_PARTS];
    SF_CHECK(spfh_wire_parts(ctx, sp, parts));
    const size_t rows = (size_t)(end - begin), need = rows * (parts[0].row + parts[1].row + parts[2].row);
    if (bytes) *bytes = need;
    if (!host) return SF_OK; // size query
    if (cap < need) { sf_set_error("sf_spfh_rows_image: %zu bytes needed, %zu given", need, cap); return SF_ERR_ARG; }
    char *h = (char *)host;
    for (const spfh_part &pt : parts) {
        if (!pt.row) continue;
        char *d = pt.base + (size_t)begin * pt.row;
        if (rows) {
            if (write_back) SF_HIP(hipMemcpyAsync(d, h, rows * pt.row, hipMemcpyHostToDevice, ctx->stream));
            else SF_HIP(hipMemcpyAsync(h, d, rows * pt.row, hipMemcpyDeviceToHost, ctx->stream));
        }
        h += rows * pt.row;
    }
    SF_HIP(hipStreamSynchronize(ctx->stream));
    return SF_OK;
}

// Neighbour-to-neighbour exchange of SPFH rows (collective C1 without the all-gather): for operation i this
// rank sends its rows [send_begin[i], send_end[i]) to rank peer[i] and receives that rank's rows into
// [recv_begin[i], recv_end[i]) -- cell-sorted positions, which number the replicated cloud identically on every rank, so
// a rank's z-slab block borrows exactly the one-layer halo its FPFH reduction reads (sharding.py plans the ranges).
// What travels per row is what K7 reads per neighbour, no more: on the byte table with at most two live 16-bin blocks
// the 32-byte packed row + the 32-byte {x, y, z, k} record (64 B instead of the 1000 B of the float64 row); with more
// live blocks the 128-byte row + the record; on the wider tables the row + k.  The choice is a function of the table's
// storage and of the HOST-known block mask only, so every rank makes the same one (the ranks size their tables by
// sf_nbrs_max_count_all).  One RCCL group: all sends and receives of a rank are in flight together.
extern "C" int sf_spfh_exchange_rows(sf_ctx *ctx, sf_spfh *sp, int n_ops, const int *peer, const int64_t *send_begin,
                                     const int64_t *send_end, const int64_t *recv_begin, const int64_t *recv_end)
{
    if (!ctx || !sp || n_ops < 0 || (n_ops && (!peer || !send_begin || !send_end || !recv_begin || !recv_end))) {
        sf_set_error("sf_spfh_exchange_rows: bad argument");
        return SF_ERR_ARG;
    }
    for (int i = 0; i < n_ops; ++i)
        if (send_begin[i] < 0 || send_begin[i] > send_end[i] || send_end[i] > sp->n || recv_begin[i] < 0 ||
            recv_begin[i] > recv_end[i] || recv_end[i] > sp->n) {
            sf_set_error("sf_spfh_exchange_rows: operation %d names rows outside the table of %lld", i, (long long)sp->n);
            return SF_ERR_ARG;
        }
    SF_HIP(hipSetDevice(ctx->device));
    spfh_part parts[SF_WIRE_PARTS];
    SF_CHECK(spfh_wire_parts(ctx, sp, parts));
    std::vector<int> peers;
    std::vector<const void *> sends;
    std::vector<void *> recvs;
    std::vector<size_t> sbytes, rbytes;
    for (int i = 0; i < n_ops; ++i)
        for (const spfh_part &pt : parts) {
            if (!pt.row) continue;
            peers.push_back(peer[i]);
            sends.push_back(pt.base + (size_t)send_begin[i] * pt.row);
            sbytes.push_back((size_t)(send_end[i] - send_begin[i]) * pt.row);
            recvs.push_back(pt.base + (size_t)recv_begin[i] * pt.row);
            rbytes.push_back((size_t)(recv_end[i] - recv_begin[i]) * pt.row);
        }
    return sf_comm_exchange(ctx, (int)peers.size(), peers.data(), sends.data(), sbytes.data(), recvs.data(), rbytes.data());
}

extern "C" int sf_spfh_export(sf_ctx *ctx, sf_cloud *c, sf_spfh *sp, double *out, int flags)
{
    if (!ctx || !c || !sp || !out) { sf_set_error("sf_spfh_export: null argument"); return SF_ERR_ARG; }
    SF_HIP(hipSetDevice(ctx->device));
    const int64_t n = sp->n, tot = n * sp->nb3;
    sf_pool_guard tmp(ctx);
    double *dout = out, *owned = nullptr;
    if (!(flags & SF_OUT_DEVICE)) {
        SF_CHECK(tmp.alloc(&owned, (size_t)tot));
        dout = owned;
    }
    if (tot) {
        const dim3 grid((unsigned)sf_div_up(tot, 256)), block(256);
        if (sp->elem_bytes == 1) {
            SF_LAUNCH(ctx, "k6_spfh_export", k_spfh_export<uint8_t>, grid, block, (const uint8_t *)sp->counts, sp->k,
                      c->perm, n, sp->nb3, sp->stride, (unsigned)sp->bias, dout, (const uint8_t *)sp->hi, sp->win_lo, sp->win_len);
        } else if (sp->elem_bytes == 2) {
            SF_LAUNCH(ctx, "k6_spfh_export", k_spfh_export<uint16_t>, grid, block, (const uint16_t *)sp->counts, sp->k,
                      c->perm, n, sp->nb3, sp->stride, 0u, dout, (const uint8_t *)nullptr, 0, sp->nb3);
        } else {
            SF_LAUNCH(ctx, "k6_spfh_export", k_spfh_export<uint32_t>, grid, block, (const uint32_t *)sp->counts, sp->k,
                      c->perm, n, sp->nb3, sp->stride, 0u, dout, (const uint8_t *)nullptr, 0, sp->nb3);
        }
    }
    if (owned) {
        if (tot) SF_HIP(hipMemcpyAsync(out, owned, (size_t)tot * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        SF_HIP(hipStreamSynchronize(ctx->stream));
    }
    return SF_OK;
}
