// fpfh.hip -- K7: the FPFH weighted reduction over the SPFH table (spfh.hip: K6 and the table).
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
    // keypoints whose own list exceeds it; k_fpfh_mcl serves exactly those.
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
    const int long_limit = 255; // lists above it: k_fpfh_mcl
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
    if (any_tail) {
        // the lists above 255 points, on the matrix cores too (k_fpfh_mcl; the vector-ALU form it replaced in round 5 -- 3.4 x the
        // cost per pair, kept for a round as a cross-check -- is gone: tests/test_hip_round5.py holds the long form to the CPU restatement)
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

