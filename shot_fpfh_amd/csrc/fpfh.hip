// fpfh.hip -- K6 (SPFH integer histograms for every cloud point) and K7 (FPFH weighted reduction).
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
    const double p_inv_width = (double)nb / (ed.p[nb] - ed.p[0]); // np.linspace edges: equal widths up to rounding
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

// K7.  The vector-memory pipe of a CU takes 16 cycles per wave instruction whatever the width per lane, so
// the neighbour rows are fetched 16 B per lane (dwordx4): a row of RB bytes occupies LPR = RB/16 lanes and
// ONE load instruction brings in 64/LPR rows (4 rows of 125 uint16 bins).  Each lane accumulates the
// 16/sizeof(CT) bins of its 16-byte piece (NP pieces when a row is longer than 1 KiB) in float64 and the
// lane groups are summed with shuffles at the end.  NCH as in K6 (0 = any list length).
template <typename CT, int LPR, int NP, int NCH>
__global__ __launch_bounds__(256) void k_fpfh(const double *__restrict__ rec, const int64_t *__restrict__ offset,
                                              const int32_t *__restrict__ cnt, const int32_t *__restrict__ idx,
                                              int64_t nbrs_begin,
                                              const int32_t *__restrict__ kp_pos, int64_t m, int nb3, int stride,
                                              const CT *__restrict__ counts, unsigned table_bytes,
                                              const int32_t *__restrict__ kk, double *__restrict__ out)
{
    constexpr int BPP = 16 / (int)sizeof(CT); // bins per 16-byte piece
    constexpr int RPI = 64 / LPR;             // rows per load instruction
    const int lane = threadIdx.x & 63;
    const int64_t q = sf_uniform64(sf_xcd_block() * 4 + (threadIdx.x >> 6));
    if (q >= m) return;
    // keypoint's cell-sorted position and its slot in the neighbour lists
    const int64_t i = kp_pos ? (int64_t)kp_pos[q] : nbrs_begin + q;
    const int64_t slot = i - nbrs_begin;
    const int64_t s = offset[slot];
    const int k = cnt[slot];
    const double px = rec[6 * i + 0], py = rec[6 * i + 1], pz = rec[6 * i + 2];
    const int grp = lane / LPR, piece = lane % LPR;
    double acc[NP][BPP];
#pragma unroll
    for (int u = 0; u < NP; ++u)
#pragma unroll
        for (int e = 0; e < BPP; ++e) acc[u][e] = 0.0;
    // descriptor of the whole SPFH table (wave-uniform); table_bytes < 4 GiB is checked by the host
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<CT *>(counts), 0, (int)table_bytes, 0x00020000);

    // exact uint32 -> double without v_cvt_f64_u32: the bit pattern {hi = 0x43300000, lo = c} is the double
    // 2^52 + c, and subtracting 2^52 is exact for c < 2^32
    auto u2d = [](unsigned c) -> double { return __hiloint2double(0x43300000, (int)c) - 4503599627370496.0; };
    // weight of neighbour j: spfh[j] / d_j with spfh[j] = count_j / k_j ; d == 0 is masked out (fpfh.py:110-114)
    auto weight_of = [&](double x, double y, double z, int kj) -> double {
        const double cx = x - px, cy = y - py, cz = z - pz;
        const double d2 = (cx * cx + cy * cy) + cz * cz;
        // 1 / (k_j d_j) = rsqrt(d2 k_j^2): v_rsq_f64 and two Newton steps (~1 ulp; the weight is a continuous
        // factor, no decision depends on its last bit) instead of sqrt + division (~35 instructions)
        const double kd = (double)kj, xx = d2 * (kd * kd);
        const double y0 = __builtin_amdgcn_rsq(xx);
        const double y1 = __builtin_fma(0.5 * y0, __builtin_fma(-(xx * y0), y0, 1.0), y0);
        const double y2 = __builtin_fma(0.5 * y1, __builtin_fma(-(xx * y1), y1, 1.0), y1);
        return d2 > 0.0 ? y2 : 0.0;
    };
    auto accumulate = [&](const uint4 &v, double ww, int u) {
        if (sizeof(CT) == 2) {
            // A 32-bit word holds two counts, x = c1 * 65536 + c0.  The even accumulator takes w * c0; the odd
            // one takes w * x WITHOUT extracting c1 (one VALU instruction less per pair) and is turned into
            // sum(w * c1) = (odd - even) / 65536 once, after the loop.  Every product is exact inside the FMA and
            // the final scaling is a power of two, so the odd bins lose nothing beyond eps * (their own sum +
            // 2^-16 of the even neighbour's).
            acc[u][0] = __builtin_fma(u2d(v.x & 0xffffu), ww, acc[u][0]);
            acc[u][1] = __builtin_fma(u2d(v.x), ww, acc[u][1]);
            acc[u][2] = __builtin_fma(u2d(v.y & 0xffffu), ww, acc[u][2]);
            acc[u][3] = __builtin_fma(u2d(v.y), ww, acc[u][3]);
            acc[u][4 % BPP] = __builtin_fma(u2d(v.z & 0xffffu), ww, acc[u][4 % BPP]);
            acc[u][5 % BPP] = __builtin_fma(u2d(v.z), ww, acc[u][5 % BPP]);
            acc[u][6 % BPP] = __builtin_fma(u2d(v.w & 0xffffu), ww, acc[u][6 % BPP]);
            acc[u][7 % BPP] = __builtin_fma(u2d(v.w), ww, acc[u][7 % BPP]);
        } else {
            acc[u][0] = __builtin_fma(u2d(v.x), ww, acc[u][0]);
            acc[u][1] = __builtin_fma(u2d(v.y), ww, acc[u][1]);
            acc[u][2] = __builtin_fma(u2d(v.z), ww, acc[u][2]);
            acc[u][3] = __builtin_fma(u2d(v.w), ww, acc[u][3]);
        }
    };
    // stream the rows of one chunk (lane t holds neighbour t's row index j and weight w; w = 0 past the end):
    // per step, lane group g takes neighbour tt + g; 4 steps' loads are issued before any is consumed
    auto stream_rows = [&](int j, double w, int cnt) {
        constexpr int UNR = 4;
        for (int tt = 0; tt < cnt; tt += RPI * UNR) {
            uint4 v[UNR][NP];
            double ww[UNR];
#pragma unroll
            for (int e = 0; e < UNR; ++e) {
                const int src = tt + e * RPI + grp; // < 64 whenever tt + e*RPI < 64; beyond cnt the weight is 0
                const int jj = __shfl(j, src & 63);
                ww[e] = (tt + e * RPI < cnt) ? __shfl(w, src & 63) : 0.0;
                const unsigned voff = (unsigned)jj * (unsigned)(LPR * 16 * NP) + (unsigned)piece * 16u; // row bytes: a shift
#pragma unroll
                for (int u = 0; u < NP; ++u) {
                    const auto r = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff + (unsigned)u * 1024u, 0, 0);
                    v[e][u] = make_uint4(r[0], r[1], r[2], r[3]);
                }
            }
#pragma unroll
            for (int e = 0; e < UNR; ++e)
#pragma unroll
                for (int u = 0; u < NP; ++u) accumulate(v[e][u], ww[e], u);
        }
    };
    if (NCH > 0) {
        constexpr int NC = NCH > 0 ? NCH : 1;
        int jv[NC];
        double wv[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int t = c * 64 + lane;
            jv[c] = t < k ? idx[s + t] : -1;
        }
        double gx[NC], gy[NC], gz[NC];
        int gk[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            if (c * 64 < k) { // wave-uniform: chunks beyond the list cost nothing
                const int j = jv[c] < 0 ? 0 : jv[c];
                sf_load_xyz(rec, j, gx[c], gy[c], gz[c]);
                gk[c] = kk[j];
            } else {
                gx[c] = gy[c] = gz[c] = 0.0;
                gk[c] = 1;
            }
        }
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            if (c * 64 < k) {
                wv[c] = jv[c] < 0 ? 0.0 : weight_of(gx[c], gy[c], gz[c], gk[c]);
                jv[c] = jv[c] < 0 ? 0 : jv[c];
                stream_rows(jv[c], wv[c], min(64, k - c * 64));
            }
        }
    } else {
        for (int t0 = 0; t0 < k; t0 += 64) {
            const int t = t0 + lane;
            int j = 0;
            double w = 0.0;
            if (t < k) {
                j = idx[s + t];
                double x, y, z;
                sf_load_xyz(rec, j, x, y, z);
                w = weight_of(x, y, z, kk[j]);
            }
            stream_rows(j, w, min(64, k - t0));
        }
    }
    // sum the lane groups (each holds a partial sum over its share of the neighbours)
#pragma unroll
    for (int off = LPR; off < 64; off <<= 1)
#pragma unroll
        for (int u = 0; u < NP; ++u)
#pragma unroll
            for (int e = 0; e < BPP; ++e) acc[u][e] += __shfl_xor(acc[u][e], off);
    if (sizeof(CT) == 2) {
#pragma unroll
        for (int u = 0; u < NP; ++u)
#pragma unroll
            for (int e = 1; e < BPP; e += 2) acc[u][e] = (acc[u][e] - acc[u][e - 1]) * (1.0 / 65536.0);
    }
    // every lane group now holds the complete sums; group g writes bins e = g, g + RPI, ... of each piece so the
    // divisions are shared out instead of being executed (predicated) by the whole wave for group 0 alone
    {
        const double kd = (double)k;
        double inv_k = __builtin_amdgcn_rcp(kd); // 1 / k to ~1 ulp for the neighbour term (the SPFH term keeps its division)
        inv_k = __builtin_fma(inv_k, __builtin_fma(-kd, inv_k, 1.0), inv_k);
        inv_k = __builtin_fma(inv_k, __builtin_fma(-kd, inv_k, 1.0), inv_k);
        const CT *own = counts + i * (int64_t)stride;
        double *o = out + q * (int64_t)nb3;
#pragma unroll
        for (int u = 0; u < NP; ++u)
#pragma unroll
            for (int e0 = 0; e0 < BPP; e0 += RPI) {
                // select acc[u][e0 + grp] without a run-time register index
                double a = acc[u][e0];
#pragma unroll
                for (int g = 1; g < RPI; ++g)
                    if (e0 + g < BPP) a = grp == g ? acc[u][e0 + g] : a;
                const int e = e0 + grp;
                const int b = (u * 64 + piece) * BPP + e; // == byte offset (u*1024 + piece*16) / sizeof(CT) + e
                if (e < BPP && b < nb3) o[b] = (double)own[b] / kd + a * inv_k;
            }
    }
}

#include "fpfh_mc.h"

// The two matrix-core forms of K7 are separate kernels and the host picks one from its copy of the table-wide block mask.
// Should that copy ever disagree with the device's, the keypoint's row is filled with NaN and word 2 of `live` raised
// (sf_fpfh reports it at its next synchronisation) -- a wrong launch is loud, never an unwritten row.
__device__ inline void fpfh_mc_wrong_form(const unsigned *__restrict__ live, double *__restrict__ out, int64_t q, int nb3)
{
    const int lane = threadIdx.x & 63;
    for (int b = lane; b < nb3; b += 64) out[q * nb3 + b] = __builtin_nan("");
    if (lane == 0) atomicOr(const_cast<unsigned *>(live) + 2, 1u);
}

// Occupancy: the full form keeps eight 4-register accumulators and wants 76 registers -- six waves per SIMD, no spill: 1.33 ms
// per 1M keypoints at C3 (asked for seven waves: 8 bytes of scratch, 1.37 ms; for eight: 24 bytes, 1.45-1.54 ms).  LDS (4.6 KB
// per wave) allows 8.5 waves per SIMD.
// HI: some point of the table has more than 255 neighbours (sf_spfh::hi).  The FULL form is always launched in its HI
// instantiation (without long neighbours its masks are zero and the correction executes nothing; `hi` may then be null): the
// 3-chunk instantiation WITHOUT it wants 96 registers and, held to eight waves as this kernel was until late in round 4,
// spilled 360 bytes into its step loop -- 24 ms per 1M keypoints, every row correct (SF_FPFH_DENSE=1 shows it; tools/check_spills.py
// lists every kernel's private segment; tests/test_hip_round4.py holds the full form to a time as well as to its rows).
template <int NKS, bool HI, bool PADC = false>
__global__ __launch_bounds__(64 * SF_MC_WPB) __attribute__((amdgpu_waves_per_eu(PADC ? 5 : 6, 8))) void k_fpfh_mc(const double *__restrict__ rec, const int64_t *__restrict__ offset,
                                                 const int32_t *__restrict__ cnt, const int32_t *__restrict__ idx,
                                                 int64_t nbrs_begin, const int32_t *__restrict__ kp_pos, int64_t m,
                                                 sf_bin_window W, const uint8_t *__restrict__ counts, unsigned table_bytes,
                                                 const double *__restrict__ p4, const unsigned *__restrict__ live,
                                                 const uint8_t *__restrict__ packed, unsigned packed_bytes,
                                                 double *__restrict__ out, const uint8_t *__restrict__ hi, int limit,
                                                 const int32_t *__restrict__ sel, int64_t nsel, int64_t view_first)
{
    __shared__ __attribute__((aligned(16))) unsigned rowbuf_all[SF_MC_WPB][32 * 32]; // 32 rows of 128 B
    __shared__ __attribute__((aligned(16))) unsigned char abuf_all[SF_MC_WPB][9 * 64];
    const int wv_id = threadIdx.x >> 6;
    int64_t q = sf_uniform64(sf_xcd_block() * SF_MC_WPB + wv_id);
    if (sel) { // (the launch of the lists that need more chunks than the bulk: sf_dispatch::mid_sel)
        if (q >= nsel) return;
        q = (int64_t)sf_uniform(sel[q]) - view_first;
        if (q < 0) return;
    }
    if (q >= m) return;
    // (the full and the sparse-block form are two kernels -- in ONE the full form's register allocation suffered, 1.45
    // instead of 1.29 ms on a table with all eight blocks live -- and the host launches the one the table-wide block mask
    // asks for; the check here only guards against a stale host copy)
    if (__popc(sf_uniform(*live) & 0xffu) <= 2) { // the host launched the wrong form: never leave the row unwritten
        fpfh_mc_wrong_form(live, out, q, W.nb3);
        return;
    }
    fpfh_mc_body<NKS, HI, PADC>(rec, offset, cnt, idx, nbrs_begin, kp_pos, W, counts, table_bytes, p4, out, q, rowbuf_all[wv_id],
                          abuf_all[wv_id], hi, limit);
}

// K7 when at most two of the table's eight 16-bin blocks hold anything at all (K6's OR over every row): the reference's
// un-normalised v keeps alpha in ONE of its bins whenever the radius is well below that bin's width, so 100 of the 125
// bins are structurally empty -- only those blocks are streamed and multiplied (fpfh_mc_body_sparse)
template <int NKS, bool HI>
__global__ __launch_bounds__(64 * SF_MC_WPB) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_fpfh_mc_sparse(
    const double *__restrict__ rec, const int64_t *__restrict__ offset, const int32_t *__restrict__ cnt,
    const int32_t *__restrict__ idx, int64_t nbrs_begin, const int32_t *__restrict__ kp_pos, int64_t m, sf_bin_window W,
    const uint8_t *__restrict__ counts, unsigned table_bytes, const double *__restrict__ p4, const unsigned *__restrict__ live,
    const uint8_t *__restrict__ packed, unsigned packed_bytes, double *__restrict__ out, const uint8_t *__restrict__ hi, int limit,
    const int32_t *__restrict__ sel, int64_t nsel, int64_t view_first)
{
    __shared__ __attribute__((aligned(16))) unsigned rowbuf_all[SF_MC_WPB][32 * 32]; // four steps of 32 rows x 32 B
    __shared__ __attribute__((aligned(16))) unsigned char abuf_all[SF_MC_WPB][9 * 64];
    const int wv_id = threadIdx.x >> 6;
    int64_t q = sf_uniform64(sf_xcd_block() * SF_MC_WPB + wv_id);
    if (sel) {
        if (q >= nsel) return;
        q = (int64_t)sf_uniform(sel[q]) - view_first;
        if (q < 0) return;
    }
    if (q >= m) return;
    const unsigned mask = sf_uniform(*live) & 0xffu;
    if (__popc(mask) > 2) { // the full kernel's case
        fpfh_mc_wrong_form(live, out, q, W.nb3);
        return;
    }
    const int b0 = mask ? __ffs(mask) - 1 : 0;
    const unsigned rest = mask & (mask - 1u);
    const int b1 = rest ? __ffs(rest) - 1 : (b0 + 1) & 7; // (a lone live block is paired with an empty one)
    // ... from the packed copy (32 bytes per row: four rows per cache line) when every row of it was written under this
    // very mask, else from the table itself
    if (sf_uniform(live[1]) == mask) {
        fpfh_mc_body_sparse<NKS, true, HI>(rec, offset, cnt, idx, nbrs_begin, kp_pos, W, counts, packed, packed_bytes, p4, out, q,
                                           b0, b1, rowbuf_all[wv_id], abuf_all[wv_id], hi, limit);
    } else {
        fpfh_mc_body_sparse<NKS, false, HI>(rec, offset, cnt, idx, nbrs_begin, kp_pos, W, counts, counts, table_bytes, p4, out, q,
                                            b0, b1, rowbuf_all[wv_id], abuf_all[wv_id], hi, limit);
    }
}

// K7 for the keypoints whose own list exceeds 255 points, on the matrix cores (fpfh_mc.h: fpfh_mcl_body / _sparse): launched over
// the selection of those keypoints (SEL) or over every keypoint (those of the main launch return at once).
#ifndef SF_MCL_SC
#define SF_MCL_SC 8 // chunks of 64 neighbours whose loads are in flight together (sparse form; the full form holds 4)
#endif
template <bool SEL, bool SPARSE, bool PADC = false>
__global__ __launch_bounds__(64 * SF_MC_WPB) void k_fpfh_mcl(const double *__restrict__ rec, const int64_t *__restrict__ offset,
                                                 const int32_t *__restrict__ cnt, const int32_t *__restrict__ idx,
                                                 int64_t nbrs_begin, const int32_t *__restrict__ kp_pos, int64_t m,
                                                 sf_bin_window W, const uint8_t *__restrict__ counts, unsigned table_bytes,
                                                 const double *__restrict__ p4, const unsigned *__restrict__ live,
                                                 const uint8_t *__restrict__ packed, unsigned packed_bytes,
                                                 double *__restrict__ out, const uint8_t *__restrict__ hi, int limit,
                                                 const int32_t *__restrict__ sel, int64_t nsel, int64_t view_first)
{
    __shared__ __attribute__((aligned(16))) unsigned rowbuf_all[SF_MC_WPB][32 * 32];
    __shared__ __attribute__((aligned(16))) unsigned char abuf_all[SF_MC_WPB][9 * 64];
    const int wv_id = threadIdx.x >> 6;
    int64_t q = sf_uniform64(sf_xcd_block() * SF_MC_WPB + wv_id);
    if (SEL) {
        if (q >= nsel) return;
        q = (int64_t)sf_uniform(sel[q]) - view_first;
        if (q < 0) return;
    }
    if (q >= m) return;
    const unsigned mask = sf_uniform(*live) & 0xffu;
    if ((__popc(mask) <= 2) != SPARSE) { // the host launched the wrong form: never leave the row unwritten
        const int64_t i = kp_pos ? (int64_t)kp_pos[q] : nbrs_begin + q;
        if (sf_uniform(cnt[i - nbrs_begin]) > limit) fpfh_mc_wrong_form(live, out, q, W.nb3);
        return;
    }
    if (SPARSE) {
        const int b0 = mask ? __ffs(mask) - 1 : 0;
        const unsigned rest = mask & (mask - 1u);
        const int b1 = rest ? __ffs(rest) - 1 : (b0 + 1) & 7; // (a lone live block is paired with an empty one)
        if (sf_uniform(live[1]) == mask) {
            fpfh_mcl_body_sparse<true, SF_MCL_SC>(rec, offset, cnt, idx, nbrs_begin, kp_pos, W, counts, packed, packed_bytes, p4, out, q, b0, b1,
                                                  rowbuf_all[wv_id], abuf_all[wv_id], hi, limit);
        } else {
            fpfh_mcl_body_sparse<false, SF_MCL_SC>(rec, offset, cnt, idx, nbrs_begin, kp_pos, W, counts, counts, table_bytes, p4, out, q, b0, b1,
                                                   rowbuf_all[wv_id], abuf_all[wv_id], hi, limit);
        }
    } else {
        fpfh_mcl_body<4, PADC>(rec, offset, cnt, idx, nbrs_begin, kp_pos, W, counts, table_bytes, p4, out, q, rowbuf_all[wv_id], abuf_all[wv_id], hi,
                         limit);
    }
}

// K7 for the keypoints whose own list exceeds the matrix-core form (more than 255 points): the vector ALU on the byte table,
// with EXACT sums.  The weights 1 / (k_j d_j) become 52-bit fixed point scaled by the keypoint's largest weight (a first pass
// over the list finds it) and are cut into two 26-bit limbs held as doubles; limb x count products (count = low byte +
// 256 x high byte < 2^16) are below 2^42, so float64 FMAs accumulate them WITHOUT rounding for 1024 neighbours at a time
// (< 2^52), across lanes and lane groups alike; after every 1024 neighbours the exact sums are folded into the running
// float64 totals, block after block.  A row therefore does not depend on which lane adds which neighbour: the 2-lanes-per-row
// form on the packed 32-byte rows (SPARSE: at most two live 16-bin blocks, 32 neighbours per load instruction) and the
// 8-lanes-per-row form on the full rows give the same bits -- as they must, since a sharded job switches the table to
// "every block live" when it borrows rows.  (The float64-accumulating version of the first round-4 commit took 1.24 ms for
// the 128 000 long lists of the clustered 1M-point cloud in its one usable form.)
// One wave per keypoint.  SEL: the keypoints are the processing slots listed in `sel` (owner numbering; a view keeps its own
// range); without it (keypoints by index) every keypoint is looked at and those of the main launch return at once.
template <bool SEL, bool SPARSE>
__global__ __launch_bounds__(256) void k_fpfh_tail(const double *__restrict__ rec, const int64_t *__restrict__ offset,
                                                   const int32_t *__restrict__ cnt, const int32_t *__restrict__ idx,
                                                   int64_t nbrs_begin, const int32_t *__restrict__ kp_pos, int64_t m, int nb3,
                                                   const uint8_t *__restrict__ counts, const uint8_t *__restrict__ hi,
                                                   const double *__restrict__ p4, double *__restrict__ out, int limit,
                                                   const int32_t *__restrict__ sel, int64_t nsel, int64_t view_first,
                                                   const uint8_t *__restrict__ packed, int b0, int b1)
{
    constexpr int LPR = SPARSE ? 2 : 8;  // lanes per row
    constexpr int RPI = 64 / LPR;        // rows per load instruction
    const int lane = threadIdx.x & 63;
    int64_t q = sf_uniform64(sf_xcd_block() * 4 + (threadIdx.x >> 6));
    if (SEL) {
        if (q >= nsel) return;
        q = (int64_t)sf_uniform(sel[q]) - view_first;
        if (q < 0) return;
    }
    if (q >= m) return;
    const int64_t i = kp_pos ? (int64_t)kp_pos[q] : nbrs_begin + q;
    const int64_t slot = i - nbrs_begin;
    const int k = sf_uniform(cnt[slot]);
    if (k <= limit) return; // (the main launch's keypoint)
    const int64_t s = offset[slot];
    const double px = rec[6 * i + 0], py = rec[6 * i + 1], pz = rec[6 * i + 2];
    const int grp = lane / LPR, piece = lane % LPR;
    const int blk = SPARSE ? (piece ? b1 : b0) : piece; // the 16-bin block this lane accumulates
    // The list is walked in SUPER-CHUNKS of SC x 64 neighbours: the SC index loads of a super-chunk are issued together, then the
    // SC record gathers, so that a super-chunk costs two memory round trips instead of 2 SC (the first version took a round trip
    // per index load, per record gather and per row load of every 64 neighbours: 15 dependent trips for a list of 300, and that --
    // not its 600 vector instructions -- was its time).  A list of at most SC x 64 = 512 points keeps its entries and weights in
    // registers from the first pass (the largest weight) to the second (the sums).
    constexpr int SC = 8;
    // entries and weights of the super-chunk starting at `base` (weight 0 past the end and at distance 0, fpfh.py:110-114);
    // bit c of `lng`: the lane's neighbour of chunk c has high bytes to add
    auto load_super = [&](int base, int (&jv)[SC], double (&wv)[SC], unsigned &lng) {
#pragma unroll
        for (int c = 0; c < SC; ++c) {
            const int t = base + 64 * c + lane;
            jv[c] = (base + 64 * c < k && t < k) ? SF_LIST_LOAD(idx + s + t) : -1;
        }
        double2 u0[SC], u1[SC];
#pragma unroll
        for (int c = 0; c < SC; ++c) {
            u0[c] = u1[c] = make_double2(0.0, 0.0);
            if (base + 64 * c < k) { // (wave-uniform)
                const double2 *pp = reinterpret_cast<const double2 *>(p4 + 4 * (size_t)(jv[c] < 0 ? 0 : jv[c]));
                u0[c] = pp[0];
                u1[c] = pp[1];
            }
        }
        lng = 0u;
#pragma unroll
        for (int c = 0; c < SC; ++c) {
            const double cx = u0[c].x - px, cy = u0[c].y - py, cz = u1[c].x - pz;
            const double d2 = (cx * cx + cy * cy) + cz * cz;
            const double kd = u1[c].y, xx = d2 * (kd * kd);
            const double y0 = __builtin_amdgcn_rsq(xx);
            const double y1 = __builtin_fma(0.5 * y0, __builtin_fma(-(xx * y0), y0, 1.0), y0);
            const double y2 = __builtin_fma(0.5 * y1, __builtin_fma(-(xx * y1), y1, 1.0), y1);
            const bool on = jv[c] >= 0;
            wv[c] = (on && d2 > 0.0) ? y2 : 0.0;
            lng |= (on && kd > 255.0) ? 1u << c : 0u;
            jv[c] = on ? jv[c] : 0; // (past the end of the list the weight is 0 and row 0 is read)
        }
    };
    // ---- pass 0: the largest weight -> the fixed-point exponent ----
    int jv[SC];
    double wv[SC];
    unsigned lng = 0u;
    double wmax = 0.0;
    for (int base = 0; base < k; base += 64 * SC) {
        load_super(base, jv, wv, lng);
#pragma unroll
        for (int c = 0; c < SC; ++c) wmax = fmax(wmax, wv[c]);
    }
    wmax = sf_wave_max_nonneg(wmax);
    const int e2 = wmax > 0.0 ? (int)((__double2hiint(wmax) >> 20) & 0x7ff) - 1023 : 0;
    const int S = 51 - e2; // W = floor(w 2^S) < 2^52
    double acc0[16], acc1[16]; // exact sums of (low / high 26-bit limb) x count over this lane's neighbours of the block
#pragma unroll
    for (int e = 0; e < 16; ++e) acc0[e] = acc1[e] = 0.0;
    constexpr int BPL = SPARSE ? 1 : 2;  // bins per lane in the end: 16 bins of a block over 16 (of 32) resp. 8 groups
    constexpr int NG = 16 / BPL;
    double tot[BPL];
#pragma unroll
    for (int u = 0; u < BPL; ++u) tot[u] = 0.0;
    const double unscale = ldexp(1.0, -S);
    auto add_words = [&](const uint4 &v, double w0, double w1) { // sixteen counts, one per byte: exact FMAs
        const unsigned wd[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const double c = (double)((wd[u] >> (8 * e)) & 0xffu);
                acc0[4 * u + e] = __builtin_fma(c, w0, acc0[4 * u + e]);
                acc1[4 * u + e] = __builtin_fma(c, w1, acc1[4 * u + e]);
            }
    };
    auto fold = [&]() { // the block's exact sums over all lane groups (still exact: < 2^52), then into the float64 totals
#pragma unroll
        for (int off = LPR; off < 64; off <<= 1)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                acc0[e] += __shfl_xor(acc0[e], off);
                acc1[e] += __shfl_xor(acc1[e], off);
            }
#pragma unroll
        for (int u = 0; u < BPL; ++u) {
            double a0 = acc0[NG * u], a1 = acc1[NG * u];
#pragma unroll
            for (int g = 1; g < NG; ++g) {
                a0 = grp == g ? acc0[NG * u + g] : a0;
                a1 = grp == g ? acc1[NG * u + g] : a1;
            }
            tot[u] += __builtin_fma(a1, 67108864.0, a0) * unscale; // (2^26; one rounding per block and bin)
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) acc0[e] = acc1[e] = 0.0;
    };
    for (int base = 0; base < k; base += 64 * SC) {
        if (k > 64 * SC) load_super(base, jv, wv, lng); // (a list of one super-chunk still holds it from pass 0)
#pragma unroll
        for (int c = 0; c < SC; ++c) {
            const int t0 = base + 64 * c;
            if (t0 < k) { // (wave-uniform)
                const double W = floor(ldexp(wv[c], S)); // < 2^52, exact
                // every row group of the chunk, whatever the list's end (weight 0 and row 0 past it): a fixed trip count, so the
                // row loads of a chunk are all in flight before the first of them is consumed
#pragma unroll
                for (int tt = 0; tt < 64; tt += RPI) {
                    const int src = (tt + grp) & 63;
                    const int jj = __shfl(jv[c], src);
                    const double Ws = __shfl(W, src);
                    const bool ll = ((unsigned)__shfl((int)lng, src) >> c) & 1u;
                    const double w1 = floor(Ws * (1.0 / 67108864.0)), w0 = __builtin_fma(-w1, 67108864.0, Ws); // two 26-bit limbs
                    uint4 v = SPARSE ? *reinterpret_cast<const uint4 *>(packed + (size_t)jj * 32 + 16 * piece)
                                     : *reinterpret_cast<const uint4 *>(counts + (size_t)jj * 128 + 16 * piece);
                    v.x ^= 0x80808080u; v.y ^= 0x80808080u; v.z ^= 0x80808080u; v.w ^= 0x80808080u; // stored as count ^ 128
                    add_words(v, w0, w1);
                    if (ll) add_words(*reinterpret_cast<const uint4 *>(hi + (size_t)jj * 128 + 16 * blk), w0 * 256.0, w1 * 256.0);
                }
                if (((t0 + 64) & 1023) == 0 || t0 + 64 >= k) fold();
            }
        }
    }
    const double kd = (double)k;
    double inv_k = __builtin_amdgcn_rcp(kd);
    inv_k = __builtin_fma(inv_k, __builtin_fma(-kd, inv_k, 1.0), inv_k);
    inv_k = __builtin_fma(inv_k, __builtin_fma(-kd, inv_k, 1.0), inv_k);
    double *o = out + q * (int64_t)nb3;
    auto own_count = [&](int b) -> double {
        unsigned own = (unsigned)counts[i * 128 + b] ^ 128u;
        if (k > 255) own += 256u * (unsigned)hi[i * 128 + b];
        return (double)own;
    };
    // every lane group holds the complete sums; group g writes bin(s) g (+ 8) of its lane's block
    if (!SPARSE || grp < 16) {
#pragma unroll
        for (int u = 0; u < BPL; ++u) {
            const int b = 16 * blk + NG * u + grp;
            if (b < nb3) o[b] = own_count(b) / kd + tot[u] * inv_k; // spfh[kp] + sum / len(neighbourhood)  (fpfh.py:109-115)
        }
    }
    if (SPARSE) // the bins of the dead blocks: no neighbour has a count there
        for (int b = lane; b < nb3; b += 64)
            if ((b >> 4) != b0 && (b >> 4) != b1) o[b] = own_count(b) / kd;
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

constexpr int FG_TILE = 512;
__global__ __launch_bounds__(256) void k_fpfh_generic(const double *__restrict__ rec, const int64_t *__restrict__ offset,
                                                      const int32_t *__restrict__ cnt, const int32_t *__restrict__ idx,
                                                      int64_t nbrs_begin, const int32_t *__restrict__ kp_pos, int64_t m, int nb3,
                                                      int stride, const unsigned *__restrict__ counts,
                                                      const int32_t *__restrict__ kk, double *__restrict__ out)
{
    __shared__ int tj[FG_TILE];
    __shared__ double tinvd[FG_TILE], tk[FG_TILE];
    const int64_t q = blockIdx.x;
    if (q >= m) return;
    const int64_t i = kp_pos ? (int64_t)kp_pos[q] : nbrs_begin + q; // cell-sorted position of the keypoint
    const int64_t lq = i - nbrs_begin, s = offset[lq];
    const int k = cnt[lq];
    double px, py, pz;
    sf_load_xyz(rec, (int)i, px, py, pz);
    double *o = out + q * (int64_t)nb3;
    for (int t0 = 0; t0 < k || t0 == 0; t0 += FG_TILE) {
        const int nt = min(FG_TILE, k - t0);
        __syncthreads();
        for (int t = threadIdx.x; t < nt; t += 256) {
            const int j = idx[s + t0 + t];
            double x, y, z;
            sf_load_xyz(rec, j, x, y, z);
            const double cx = x - px, cy = y - py, cz = z - pz;
            const double d = sqrt((cx * cx + cy * cy) + cz * cz);
            tj[t] = j;
            tinvd[t] = d > 0.0 ? d : 0.0; // 0 marks "skip" (fpfh.py:113: distances > 0)
            tk[t] = (double)kk[j];
        }
        __syncthreads();
        for (int b = threadIdx.x; b < nb3; b += 256) {
            double acc = t0 ? o[b] : 0.0;
            for (int t = 0; t < nt; ++t)
                if (tinvd[t] > 0.0) acc += ((double)counts[(int64_t)tj[t] * stride + b] / tk[t]) / tinvd[t]; // spfh[j] / d_j
            o[b] = acc;
        }
        if (k == 0) break;
    }
    __syncthreads();
    const double kd = (double)k;
    for (int b = threadIdx.x; b < nb3; b += 256)
        o[b] = (double)counts[i * (int64_t)stride + b] / kd + o[b] / kd; // spfh[kp] + sum / len(neighbourhood)  :109-115
}

__global__ void k_map_positions(const int64_t *__restrict__ kp_idx, const int32_t *__restrict__ inv_perm, int64_t m,
                                int64_t n, int32_t *__restrict__ pos, int *__restrict__ bad)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    int64_t v = kp_idx[i];
    if (v < 0) v += n; // NumPy negative indexing
    if (v < 0 || v >= n) { *bad = 1; pos[i] = 0; return; }
    pos[i] = inv_perm[v];
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

template <typename CT>
static int launch_fpfh(sf_ctx *ctx, sf_cloud *c, sf_nbrs *nb, sf_spfh *sp, const int32_t *kp_pos, int64_t m,
                       double *dout)
{
    const dim3 grid(sf_xcd_grid(sf_div_up(m, 4))), block(256);
    const size_t tb = (size_t)sp->rows_alloc * sp->stride * sizeof(CT);
    if (tb >= ((size_t)1 << 32)) {
        sf_set_error("sf_fpfh: SPFH table of %zu bytes exceeds the 4 GiB buffer-addressing limit of the K7 kernel", tb);
        return SF_ERR_UNSUPPORTED;
    }
    const unsigned table_bytes = (unsigned)tb;
    const int row_bytes = sp->stride * (int)sizeof(CT); // a multiple of 256
    int nch = 0;
    if (sizeof(CT) == 2) {
        const int64_t chunks = sf_div_up(nb->max_count > 0 ? nb->max_count : 1, 64);
        nch = chunks <= 2 ? 2 : (chunks <= 4 ? (int)chunks : 0);
    }
#define SF_FPFH_LAUNCH(LPR, NP, NCH)                                                                                 \
    SF_LAUNCH(ctx, "k7_fpfh", (k_fpfh<CT, LPR, NP, NCH>), grid, block, c->rec, nb->offset, nb->count, nb->idx,      \
              nb->self_begin, kp_pos, m, sp->nb3, sp->stride, (const CT *)sp->counts, table_bytes, sp->k, dout)
#define SF_FPFH_SHAPE(LPR, NP)                                                 \
    {                                                                          \
        if (sizeof(CT) == 2 && nch == 2) { SF_FPFH_LAUNCH(LPR, NP, 2); }       \
        else if (sizeof(CT) == 2 && nch == 3) { SF_FPFH_LAUNCH(LPR, NP, 3); }  \
        else if (sizeof(CT) == 2 && nch == 4) { SF_FPFH_LAUNCH(LPR, NP, 4); }  \
        else { SF_FPFH_LAUNCH(LPR, NP, 0); }                                   \
    }
    if (row_bytes == 256) SF_FPFH_SHAPE(16, 1)
    else if (row_bytes == 512) SF_FPFH_SHAPE(32, 1)
    else if (row_bytes == 1024) SF_FPFH_SHAPE(64, 1)
    else if (row_bytes == 2048) SF_FPFH_SHAPE(64, 2)
    else {
        sf_set_error("sf_fpfh: SPFH rows of %d bytes unsupported", row_bytes);
        return SF_ERR_UNSUPPORTED;
    }
#undef SF_FPFH_SHAPE
#undef SF_FPFH_LAUNCH
    return SF_OK;
}

static int launch_fpfh_mc(sf_ctx *ctx, sf_cloud *c, sf_nbrs *nb, sf_spfh *sp, const int32_t *kp_pos, int64_t m, double *dout)
{
    const dim3 grid(sf_xcd_grid(sf_div_up(m, SF_MC_WPB))), block(64 * SF_MC_WPB);
    const size_t tb = (size_t)sp->rows_alloc * 128;
    if (tb >= ((size_t)1 << 32) || sp->stride != 128 || (nb->max_count > 255 && !sp->hi)) {
        sf_set_error("sf_fpfh: uint8 SPFH table of %zu bytes / lists of %lld points outside the matrix-core kernel's range", tb,
                     (long long)nb->max_count);
        return SF_ERR_UNSUPPORTED;
    }
    // Dispatch by list length, per keypoint: the matrix-core form of the main launch (at most 255 points) leaves out the
    // keypoints whose own list exceeds it; k_fpfh_tail serves exactly those.
    sf_dispatch d = sf_nbrs_dispatch(nb);
    if (d.chunks == 0 || d.limit > 255) { // (lists not planned by a radius search: everything the matrix-core form holds)
        d.chunks = d.chunks == 0 ? 4 : d.chunks;
        d.limit = 255;
    }
    if (kp_pos && d.n_mid) { // keypoints by index have no selections: one launch that holds every list of at most 255 points
        d.chunks = 4;
        d.limit = 255;
        d.n_mid = 0;
    }
    if (sp->win_len > 127) {
        // (a window of 128 real bins runs the full form's 4-chunk instantiation with its padding column from a constant operand:
        // one launch for every list of at most 255 points)
        d.chunks = 4;
        d.limit = 255;
        d.n_mid = 0;
    }
    const int long_limit = 255; // lists above it: k_fpfh_tail
    const bool any_tail = nb->max_count > long_limit;
    const uint8_t *hi = sp->hi;
    const sf_bin_window W{sp->nb3, sp->win_lo, sp->win_len};
    const bool padc = sp->win_len > 127; // (a window of 128 real bins: no padding column in the table)
#define SF_MC_ARGS c->rec, nb->offset, nb->count, nb->idx, nb->self_begin, kp_pos, m, W, (const uint8_t *)sp->counts,       \
                   (unsigned)tb, (const double *)sp->p4, (const unsigned *)sp->live, (const uint8_t *)sp->packed,                \
                   (unsigned)((size_t)sp->rows_alloc * 32), dout, hi
    // Which form runs is decided here, on the table-wide block mask -- read back once per K6 (8 bytes; the one host
    // round trip of sf_fpfh: it waits for K6, so a caller that wants it hidden queues independent work first, as
    // DescriptorJob does with the frame eigen-solves on the side stream).  Both kernels re-check the mask on the device.
    if (!sp->host_live_valid) {
        void *pin = nullptr;
        SF_CHECK(sf_ctx_pinned(ctx, &pin));
        SF_HIP(hipMemcpyAsync(pin, sp->live, 3 * sizeof(unsigned), hipMemcpyDeviceToHost, ctx->stream));
        SF_HIP(hipStreamSynchronize(ctx->stream));
        memcpy(sp->host_live, pin, 2 * sizeof(unsigned));
        sp->host_live_valid = true;
        if (((const unsigned *)pin)[2]) {
            sf_set_error("sf_fpfh: an earlier launch took the wrong matrix-core form for this table (host copy of the block mask was stale); its rows are NaN");
            return SF_ERR_STATE;
        }
    }
    const bool sparse = __builtin_popcount(sp->host_live[0] & 0xffu) <= 2;
#define SF_MC_LAUNCH2(NAME, GRID, NKS, HI, LIMIT, SELP, NSEL)                                                         \
    if (sparse) { SF_LAUNCH(ctx, NAME, (k_fpfh_mc_sparse<NKS, HI>), GRID, block, SF_MC_ARGS, LIMIT, SELP, NSEL, d.view_first); } \
    else if (padc) { SF_LAUNCH(ctx, NAME, (k_fpfh_mc<4, true, true>), GRID, block, SF_MC_ARGS, LIMIT, SELP, NSEL, d.view_first); } \
    else { SF_LAUNCH(ctx, NAME, (k_fpfh_mc<NKS, true>), GRID, block, SF_MC_ARGS, LIMIT, SELP, NSEL, d.view_first); }
#define SF_MC_LAUNCH(NKS)                                                                                            \
    if (hi) { SF_MC_LAUNCH2("k7_fpfh", grid, NKS, true, d.limit, (const int32_t *)nullptr, (int64_t)0) }             \
    else { SF_MC_LAUNCH2("k7_fpfh", grid, NKS, false, d.limit, (const int32_t *)nullptr, (int64_t)0) }
    if (d.chunks <= 1) { SF_MC_LAUNCH(1); }
    else if (d.chunks == 2) { SF_MC_LAUNCH(2); }
    else if (d.chunks == 3) { SF_MC_LAUNCH(3); }
    else { SF_MC_LAUNCH(4); }
    if (d.n_mid) { // the few lists that need more chunks than the bulk: the same form, four chunks
        const dim3 grid_mid(sf_xcd_grid(sf_div_up(d.n_mid, SF_MC_WPB)));
        if (hi) { SF_MC_LAUNCH2("k7_fpfh_mid", grid_mid, 4, true, 255, d.mid_sel, d.n_mid) }
        else { SF_MC_LAUNCH2("k7_fpfh_mid", grid_mid, 4, false, 255, d.mid_sel, d.n_mid) }
    }
#undef SF_MC_LAUNCH
#undef SF_MC_LAUNCH2
    if (any_tail && (!getenv("SF_FPFH_TAIL_VECTOR") || sp->win_len != sp->nb3)) {
        // the lists above 255 points, on the matrix cores too (the vector-ALU form below stays as a cross-check: SF_FPFH_TAIL_VECTOR=1)
#define SF_MCL_LAUNCH(SEL, GRID, SELP, NSEL, VF)                                                                       \
        if (sparse) { SF_LAUNCH(ctx, "k7_fpfh_tail", (k_fpfh_mcl<SEL, true>), dim3(sf_xcd_grid(sf_div_up(GRID, SF_MC_WPB))), block, SF_MC_ARGS, long_limit, SELP, NSEL, VF); } \
        else if (padc) { SF_LAUNCH(ctx, "k7_fpfh_tail", (k_fpfh_mcl<SEL, false, true>), dim3(sf_xcd_grid(sf_div_up(GRID, SF_MC_WPB))), block, SF_MC_ARGS, long_limit, SELP, NSEL, VF); } \
        else { SF_LAUNCH(ctx, "k7_fpfh_tail", (k_fpfh_mcl<SEL, false>), dim3(sf_xcd_grid(sf_div_up(GRID, SF_MC_WPB))), block, SF_MC_ARGS, long_limit, SELP, NSEL, VF); }
        if (!kp_pos && d.n_tail) {
            SF_MCL_LAUNCH(true, d.n_tail, d.tail_sel, d.n_tail, d.view_first)
        } else { // keypoints by index (or lists without a selection): every keypoint is looked at
            SF_MCL_LAUNCH(false, m, (const int32_t *)nullptr, (int64_t)0, (int64_t)0)
        }
#undef SF_MCL_LAUNCH
    } else if (any_tail) {
        // (the packed 32-byte rows when the table has at most two live blocks and its packed copy is current: both forms sum
        // exactly, so which one runs changes no bit of a row)
        const unsigned m8 = sp->host_live[0] & 0xffu;
        const bool packed_ok = sparse && sp->host_live[1] == m8 && !getenv("SF_FPFH_TAIL_DENSE");
        const int pb0 = m8 ? __builtin_ffs((int)m8) - 1 : 0;
        const unsigned rest = m8 & (m8 - 1u);
        const int pb1 = rest ? __builtin_ffs((int)rest) - 1 : (pb0 + 1) & 7; // (same pairing as spfh_pack_row)
#define SF_TAIL_ARGS c->rec, nb->offset, nb->count, nb->idx, nb->self_begin, kp_pos, m, sp->nb3, (const uint8_t *)sp->counts, hi, \
                     (const double *)sp->p4, dout, long_limit
#define SF_TAIL_LAUNCH(SEL, GRID, SELP, NSEL, VF)                                                                        \
        if (packed_ok) { SF_LAUNCH(ctx, "k7_fpfh_tail", (k_fpfh_tail<SEL, true>), dim3(sf_xcd_grid(sf_div_up(GRID, 4))), dim3(256), SF_TAIL_ARGS, SELP, NSEL, VF, (const uint8_t *)sp->packed, pb0, pb1); } \
        else { SF_LAUNCH(ctx, "k7_fpfh_tail", (k_fpfh_tail<SEL, false>), dim3(sf_xcd_grid(sf_div_up(GRID, 4))), dim3(256), SF_TAIL_ARGS, SELP, NSEL, VF, (const uint8_t *)sp->packed, pb0, pb1); }
        if (!kp_pos && d.n_tail) {
            SF_TAIL_LAUNCH(true, d.n_tail, d.tail_sel, d.n_tail, d.view_first)
        } else { // keypoints by index (or lists without a selection): every keypoint is looked at
            SF_TAIL_LAUNCH(false, m, (const int32_t *)nullptr, (int64_t)0, (int64_t)0)
        }
#undef SF_TAIL_LAUNCH
#undef SF_TAIL_ARGS
    }
#undef SF_MC_ARGS
    return SF_OK;
}

extern "C" int sf_fpfh(sf_ctx *ctx, sf_cloud *c, sf_nbrs *nb, sf_spfh *sp, const int64_t *kp_idx, int64_t m, double *out,
                       int flags)
{
    if (!ctx || !c || !nb || !sp || !out || m < 0) { sf_set_error("sf_fpfh: bad argument"); return SF_ERR_ARG; }
    if (!nb->self) { sf_set_error("sf_fpfh: needs a sf_radius_search_self result"); return SF_ERR_ARG; }
    SF_CHECK(sf_nbrs_on_grid(nb, c, "sf_fpfh"));
    if (!kp_idx && m != nb->m) { sf_set_error("sf_fpfh: m must equal the query count when kp_idx is NULL"); return SF_ERR_ARG; }
    if (kp_idx && !(nb->self_begin == 0 && nb->m == c->n)) {
        sf_set_error("sf_fpfh: keypoints by index need neighbour lists of the whole cloud");
        return SF_ERR_ARG;
    }
    SF_HIP(hipSetDevice(ctx->device));
    sf_pool_guard tmp(ctx);
    int32_t *pos = nullptr;
    if (kp_idx && m) {
        const int64_t *src = kp_idx;
        int64_t *dkp = nullptr;
        int *dbad = nullptr;
        if (!(flags & SF_IN_DEVICE)) {
            SF_CHECK(tmp.alloc(&dkp, (size_t)m));
            SF_HIP(hipMemcpyAsync(dkp, kp_idx, (size_t)m * sizeof(int64_t), hipMemcpyHostToDevice, ctx->stream));
            src = dkp;
        }
        SF_CHECK(tmp.alloc(&pos, (size_t)m));
        SF_CHECK(tmp.alloc(&dbad, 1));
        SF_HIP(hipMemsetAsync(dbad, 0, sizeof(int), ctx->stream));
        SF_CHECK(sf_cloud_ensure_inv_perm(ctx, c));
        SF_LAUNCH(ctx, "k7_map_positions", k_map_positions, dim3((unsigned)sf_div_up(m, 256)), dim3(256), src,
                  c->inv_perm, m, c->n, pos, dbad);
        int bad = 0;
        SF_HIP(hipMemcpyAsync(&bad, dbad, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
        SF_HIP(hipStreamSynchronize(ctx->stream));
        if (bad) {
            sf_set_error("sf_fpfh: keypoint index out of range for a cloud of %lld points", (long long)c->n);
            return SF_ERR_ARG;
        }
    }
    const int64_t tot = m * sp->nb3;
    double *dout = out, *owned = nullptr;
    if (!(flags & SF_OUT_DEVICE)) {
        SF_CHECK(tmp.alloc(&owned, (size_t)tot));
        dout = owned;
    }
    int rc = SF_OK;
    if (m && sp->n_bins > SF_FAST_FPFH_BINS && sp->elem_bytes != 1) {
        if (m > 2147483000LL) { sf_set_error("sf_fpfh: too many keypoints for one launch"); return SF_ERR_UNSUPPORTED; }
        SF_LAUNCH(ctx, "k7_fpfh", k_fpfh_generic, dim3((unsigned)m), dim3(256), c->rec, nb->offset, nb->count, nb->idx,
                  nb->self_begin, (const int32_t *)pos, m, sp->nb3, sp->stride, (const unsigned *)sp->counts, sp->k, dout);
    } else if (m)
        rc = sp->elem_bytes == 1   ? launch_fpfh_mc(ctx, c, nb, sp, pos, m, dout)
             : sp->elem_bytes == 2 ? launch_fpfh<uint16_t>(ctx, c, nb, sp, pos, m, dout)
                                   : launch_fpfh<uint32_t>(ctx, c, nb, sp, pos, m, dout);
    if (rc == SF_OK && owned) {
        if (tot) SF_HIP(hipMemcpyAsync(out, owned, (size_t)tot * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        SF_HIP(hipStreamSynchronize(ctx->stream));
    }
    return rc;
}
