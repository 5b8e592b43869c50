// fpfh_mc.h -- body of K7 on the int8 matrix cores, as a device function: used by k_fpfh_mc (fpfh.hip) and by the
// kernel that runs K5 and K7 side by side on the same CUs (descriptors.hip).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "device_util.h"

// --------------------------------------------------------------------------------------------------
// K7 on the matrix cores (uint8 table: at most 128 bins, neighbourhoods of at most 255 points).
//
// fpfh[q][b] - spfh[q][b] = (1/k_q) sum_j w_j c_jb is, per keypoint, the product of a 1 x k row of weights with
// the k x 128 matrix of its neighbours' integer bin counts.  Done on the vector ALU it costs three instructions
// per count (extract, convert, FMA).  Here the weights are turned into 62-bit fixed point (scaled by the
// keypoint's largest weight) and cut into eight signed-byte limbs, and  R[limb][bin] = sum_j limb_j * (c_jb - 128)  is
// accumulated EXACTLY in int32 by v_mfma_i32_16x16x64_i8: A = limbs x 64 neighbours, B = 64 neighbours x 16 bins,
// the counts going from the table to the matrix unit as the bytes they are (the table stores count ^ 128, which
// read as int8 is count - 128; a padding bin holds -128, so its column is -128 * sum_j limb_j and cancels the bias).
// The sums are recombined in float64 once per keypoint:  sum_j w_j c_jb = 2^-S sum_i 2^(8i) (R[i][b] - R[i][pad]).
// The only rounding is in the fixed-point weights (2^-61 of the largest one) and in that final recombination.
//
// One wave per keypoint, 32 neighbours per step (v_mfma_i32_16x16x32_i8).  Layouts (tools/ubench: probed on the
// device): operand lane l holds row / column l % 16 and the 8 consecutive k of block l / 16, one per byte; the result
// lane holds column l % 16 and rows 4 (l / 16) .. + 3.  Per step the 32 rows (128 B each) reach LDS by LDS-DMA; the B
// operand of the MFMA for bins 16 bb .. 16 bb + 15 -- lane (a, kb): bin 16 bb + a of neighbours 8 kb .. 8 kb + 7, one
// per byte -- is exactly what ONE transposing read (ds_read_b64_tr_b8: 8 rows x 16 bytes per 16-lane group,
// delivered column-major) returns, so no byte shuffling is left to the vector pipe.
// --------------------------------------------------------------------------------------------------
typedef int v4i __attribute__((ext_vector_type(4)));

// Which bins of the descriptor the table's 128 columns are: column c = bin win_lo + c for c < win_len; the other bins of the
// nb3 are structurally empty (sf_spfh::win_lo) and the keypoint's row gets zeros there.  win_lo = 0, win_len = nb3 for tables
// of at most 128 bins.
struct sf_bin_window { int nb3, win_lo, win_len; };
__device__ __forceinline__ void fpfh_mc_zero_outside(double *__restrict__ o, const sf_bin_window &W, int lane)
{
    if (W.win_len != W.nb3) // (wave-uniform)
        for (int b = lane; b < W.nb3; b += 64)
            if (b < W.win_lo || b >= W.win_lo + W.win_len) sf_store_stream(o + b, 0.0);
}

// The keypoint's row leaves the wave as ONE run of 16-byte pieces.  A lane owns bins i0 and i1 = i0 + 16 (the layout the matrix
// core leaves the sums in): stored from there, a row is eight 128-byte runs issued by two instructions, and -- a row of 125
// doubles starting 1000 q bytes into the array -- every run straddles 32-byte sectors it shares with the run of the OTHER
// instruction: 1.37 GB of write requests for 1.0 GB of rows at C3 (WRITE_SIZE, profiles/r05_summary.md).  So the two values go
// through the wave's LDS region (free by now) and lane l stores the aligned piece l of the row, 16 bytes -- 8 at the row's two
// ends when the row starts on an odd multiple of 8: one instruction, 1000 contiguous bytes, two partial sectors per row.
// (A windowed table -- n_bins above 5 -- keeps the plain stores: its row has holes.)
__device__ __forceinline__ void fpfh_mc_store_row(double *__restrict__ o, int n, int i0, double v0, int i1, double v1,
                                                  double *stage /* this wave's, >= 128 doubles */, int lane)
{
    if (i0 < n) stage[i0] = v0;
    if (i1 < n) stage[i1] = v1;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const int shift = (int)((reinterpret_cast<uintptr_t>(o) >> 3) & 1u); // (wave-uniform) the row starts in the middle of a piece
    const int e = 2 * lane - shift;
    const bool a_ok = e >= 0 && e < n, b_ok = e + 1 < n;
    const double a = a_ok ? stage[e] : 0.0, b = b_ok ? stage[e + 1] : 0.0;
    if (a_ok && b_ok) sf_store_stream2(o + e, a, b);
    else if (a_ok) sf_store_stream(o + e, a);
    else if (b_ok) sf_store_stream(o + e + 1, b);
}

// Staged rows: 128 B each, no padding -- the image is written by LDS-DMA (buffer_load_dwordx4 ... lds: a wave
// instruction writes its 64 lanes' 16-byte chunks back to back), so the bank spread comes from the SOURCE side:
// slot s of row r holds chunk s ^ f(r), f(r) = (r >> 1) & 7.  A transposing read of one 32-lane half (8 rows x 16
// bytes of block g and of block g + 1, same chunk) then touches all 64 banks once.
typedef int v2i_t __attribute__((ext_vector_type(2)));
#ifndef SF_MC_WPB
#define SF_MC_WPB 4 // waves (= keypoints in flight) per workgroup: 4.6 KB of LDS each
#endif

typedef unsigned sf_u2 __attribute__((ext_vector_type(2)));
template <int W>
__device__ __forceinline__ void sf_lane_swap(double &a, double &b)
{
    const unsigned alo = (unsigned)__double2loint(a), ahi = (unsigned)__double2hiint(a);
    const unsigned blo = (unsigned)__double2loint(b), bhi = (unsigned)__double2hiint(b);
    const sf_u2 lo = W == 32 ? __builtin_amdgcn_permlane32_swap(alo, blo, false, false) : __builtin_amdgcn_permlane16_swap(alo, blo, false, false);
    const sf_u2 hi = W == 32 ? __builtin_amdgcn_permlane32_swap(ahi, bhi, false, false) : __builtin_amdgcn_permlane16_swap(ahi, bhi, false, false);
    a = __hiloint2double((int)hi[0], (int)lo[0]);
    b = __hiloint2double((int)hi[1], (int)lo[1]);
}

// ---- pieces shared by the full and the sparse-block form ---------------------------------------------------------
// the lane's entries of the keypoint's neighbour list, chunk by chunk (lane t of chunk c <-> neighbour 64 c + t; -1 past the
// end).  The kernel is instantiated for the longest list of the launch; the chunks past THIS keypoint's list -- the last
// one for nine keypoints in ten -- are skipped wave-uniformly: no index load, no gather, no weight.
template <int NKS>
__device__ __forceinline__ void fpfh_mc_list(const int32_t *__restrict__ idx, int64_t s, int k, int lane, int (&jv)[NKS])
{
#pragma unroll
    for (int c = 0; c < NKS; ++c) {
        const int t = c * 64 + lane;
        jv[c] = -1;
        if (c == 0 || c * 64 < k) jv[c] = t < k ? SF_LIST_LOAD(idx + s + t) : -1;
    }
}

// weights 1 / (k_j d_j) of all neighbours from their 32-byte records {x, y, z, k} (one rsqrt + two Newton steps each; d == 0
// is masked out, fpfh.py:110-114), the fixed-point exponent S = 61 - floor(log2 of the largest), and jv clamped to valid
// rows for the gathers that follow
// HI: lm[c] = the lanes of chunk c whose neighbour has more than 255 neighbours of its own (its bins are lo + 256 hi: the
// table of high bytes, fpfh_mc_hi)
template <int NKS, bool HI>
__device__ __forceinline__ int fpfh_mc_weights(const double *__restrict__ p4, double px, double py, double pz, int k,
                                               int (&jv)[NKS], double (&wv)[NKS], unsigned long long (&lm)[NKS])
{
    double gx[NKS], gy[NKS], gz[NKS], gk[NKS];
#pragma unroll
    for (int c = 0; c < NKS; ++c) {
        gx[c] = gy[c] = gz[c] = gk[c] = 0.0;
        if (c == 0 || c * 64 < k) {
            const int j = jv[c] < 0 ? 0 : jv[c];
            const double2 *pp = reinterpret_cast<const double2 *>(p4 + 4 * (size_t)j); // {x, y}, {z, k}: one 32-byte record
            const double2 u0 = pp[0], u1 = pp[1];
            gx[c] = u0.x; gy[c] = u0.y; gz[c] = u1.x; gk[c] = u1.y;
        }
    }
    double wmax = 0.0;
#pragma unroll
    for (int c = 0; c < NKS; ++c) {
        wv[c] = 0.0;
        if (c == 0 || c * 64 < k) {
            const double cx = gx[c] - px, cy = gy[c] - py, cz = gz[c] - pz;
            const double d2 = (cx * cx + cy * cy) + cz * cz;
            const double kd = gk[c], xx = d2 * (kd * kd);
            const double y0 = __builtin_amdgcn_rsq(xx);
            const double y1 = __builtin_fma(0.5 * y0, __builtin_fma(-(xx * y0), y0, 1.0), y0);
            const double y2 = __builtin_fma(0.5 * y1, __builtin_fma(-(xx * y1), y1, 1.0), y1);
            wv[c] = (jv[c] >= 0 && d2 > 0.0) ? y2 : 0.0;
            wmax = fmax(wmax, wv[c]);
            if (HI) lm[c] = __ballot(wv[c] > 0.0 && kd > 255.0);
        } else if (HI) {
            lm[c] = 0ull;
        }
        jv[c] = jv[c] < 0 ? 0 : jv[c];
    }
    wmax = sf_wave_max_nonneg(wmax);
    // fixed point: W = floor(w 2^S) < 2^62 with S = 61 - floor(log2 wmax)
    const int e2 = wmax > 0.0 ? (int)((__double2hiint(wmax) >> 20) & 0x7ff) - 1023 : 0;
    return 61 - e2;
}

// The part of sum_j w_j c_jb that the byte table does not hold: 256 sum_{j long} w_j hi_jb for the two bins this lane
// writes, in float64, neighbour after neighbour in list order (the same order on every rank and in every form, so the rows
// stay reproducible bit for bit).  Wave-uniform loop over the set bits of the masks: a keypoint without long neighbours --
// every keypoint of a cloud without long lists -- executes nothing.
template <int NKS>
__device__ __forceinline__ void fpfh_mc_hi(const uint8_t *__restrict__ hi, const int (&jv)[NKS], const double (&wv)[NKS],
                                           const unsigned long long (&lm)[NKS], int o0, int o1, double &c0, double &c1)
{
#pragma unroll
    for (int c = 0; c < NKS; ++c) {
        unsigned long long mk = lm[c];
        while (mk) { // four long neighbours per round: their eight byte loads are in flight together
            int t[4];
            bool on[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                on[u] = mk != 0ull;
                t[u] = on[u] ? __builtin_ctzll(mk) : t[0];
                mk &= mk - 1; // (0 stays 0)
            }
            unsigned b0[4], b1[4];
            double w[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int j = __builtin_amdgcn_readlane(jv[c], t[u]);
                const uint8_t *r = hi + (size_t)j * 128;
                b0[u] = r[o0];
                b1[u] = r[o1];
                const double wu = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(wv[c]), t[u]),
                                                   __builtin_amdgcn_readlane(__double2loint(wv[c]), t[u]));
                w[u] = on[u] ? wu * 256.0 : 0.0;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) { // (list order: the same sums on every rank and in every form)
                c0 = __builtin_fma(w[u], (double)b0[u], c0);
                c1 = __builtin_fma(w[u], (double)b1[u], c1);
            }
        }
    }
}

// This lane's weight as EIGHT signed-byte limbs: W = floor(w 2^S) < 2^62 is written in the signed-digit form
// W = sum_i d_i 256^i, d_i in [-128, 127] -- add 0x80 to every byte of W with carries, then flip every byte's top bit --
// which is what an int8 MFMA operand wants, 8 bits per limb instead of 7.  The limbs of neighbour t are the 8 bytes at
// abuf + 8 t: one ds_write_b64 per lane; the A operand (row = limb, k = neighbour) comes back through the same
// transposing read as the B operand (fpfh_mc_a_operand).
__device__ __forceinline__ void fpfh_mc_limbs(unsigned char *abuf, int lane, double w, int S)
{
    const double x = ldexp(w, S - 32); // < 2^30
    const unsigned hi = (unsigned)x;
    const unsigned lo = (unsigned)ldexp(x - (double)hi, 32);
    const unsigned long long W = ((unsigned long long)hi << 32) | lo;
    const unsigned long long D = (W + 0x8080808080808080ull) ^ 0x8080808080808080ull; // (hi < 2^30: no carry out)
    *reinterpret_cast<unsigned long long *>(abuf + 8 * lane) = D;
}

// A operand of a step (32 neighbours: 32 (st & 1) .. of the chunk whose limbs are in abuf): lane (a, kb) wants limb a of
// neighbours 8 kb .. 8 kb + 7, one per byte.  In its 16-lane group lane 2 q + p supplies the address of "row" q = one
// neighbour's 8 limb bytes for p = 0 and of 8 zero bytes (abuf + 512) for p = 1 -- limbs 8 .. 15 do not exist -- and
// receives column a of those eight 16-byte rows.
__device__ __forceinline__ long fpfh_mc_a_operand(const unsigned char *abuf, int a, int kb, int st)
{
    const unsigned char *ap = (a & 1) ? abuf + 512 : abuf + 8 * (32 * (st & 1) + 8 * kb + (a >> 1));
    const v2i_t t = __builtin_amdgcn_ds_read_tr8_b64_v2i32((__attribute__((address_space(3))) v2i_t *)ap);
    return (long)(((unsigned long long)(unsigned)t[1] << 32) | (unsigned)t[0]);
}

// PADC: the padding column that removes the -128 bias comes from an MFMA against a constant operand (as in the sparse form)
// instead of column 127 of the table -- for a window of 128 real bins (n_bins = 8), which leaves no padding column.
template <int NKS, bool HI, bool PADC = false>
__device__ __forceinline__ void fpfh_mc_body(const double *__restrict__ rec, const int64_t *__restrict__ offset,
                                             const int32_t *__restrict__ cnt, const int32_t *__restrict__ idx,
                                             int64_t nbrs_begin, const int32_t *__restrict__ kp_pos, sf_bin_window W,
                                             const uint8_t *__restrict__ counts, unsigned table_bytes,
                                             const double *__restrict__ p4, double *__restrict__ out, int64_t q,
                                             unsigned *rowbuf /* 4 KB */, unsigned char *abuf /* 576 B */,
                                             const uint8_t *__restrict__ hi, int limit)
{
    const int lane = threadIdx.x & 63;
    const int64_t i = kp_pos ? (int64_t)kp_pos[q] : nbrs_begin + q;
    const int64_t slot = i - nbrs_begin;
    const int64_t s = offset[slot];
    const int k = cnt[slot];
    if (sf_uniform(k) > limit) return; // (a keypoint whose own list exceeds this form: the second launch, launch_fpfh_mc)
    const double px = rec[6 * i + 0], py = rec[6 * i + 1], pz = rec[6 * i + 2];
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(counts), 0, (int)table_bytes, 0x00020000);
    const int a = lane & 15, kb = lane >> 4;
    // (the keypoint's own counts of the two bins this lane writes, requested with the list: see the sparse form)
    const unsigned own0 = counts[i * 128 + 16 * (4 * (kb >> 1) + 2 * (kb & 1)) + a], own1 = counts[i * 128 + 16 * (4 * (kb >> 1) + 2 * (kb & 1)) + a + 16];
    if (lane == 0) *reinterpret_cast<unsigned long long *>(abuf + 512) = 0ull; // the "limbs 8 .. 15" every A operand reads
    // A step covers 32 neighbours (v_mfma_i32_16x16x32_i8: 8 k per 16-lane group).  Transposing reads: in its group
    // (k block kb) lane 2 q + p supplies the address of row 8 kb + q, bytes 8 p .. + 7 of chunk bb (slot bb ^ f(row),
    // f(row) = (row >> 1) & 7), and receives bin 16 bb + a of those eight rows -- exactly its B operand.
    const int frow = ((a >> 2) & 3) | ((kb & 1) << 2);
    const int rd_base = (8 * kb + (a >> 1)) * 128 + 8 * (a & 1); // bytes
    int xoff[8];
#pragma unroll
    for (int bb = 0; bb < 8; ++bb) xoff[bb] = rd_base + 16 * (bb ^ frow);
    // DMA: lane l of instruction u fills slot l & 7 of row 8 u + (l >> 3) with chunk (l & 7) ^ f(row)
    const int dma_chunk = (lane & 7) ^ ((lane >> 4) & 3); // ^ 4 for the rows 8 .. 15 and 24 .. 31 (u = 1, 3)
    const unsigned lds_rows = (unsigned)__builtin_amdgcn_readfirstlane(
        (int)(unsigned)(size_t)(__attribute__((address_space(3))) unsigned *)rowbuf);
    // ---- stage the 32 rows of step ST (neighbours 32 ST .. + 31 = lanes 32 (ST & 1) .. of chunk ST >> 1) in LDS:
    //      4 DMA instructions of 64 x 16 bytes ----
#define SF_MC_DMA(ST)                                                                                               \
    {                                                                                                               \
        _Pragma("unroll") for (int u = 0; u < 4; ++u) {                                                             \
            const int jr0 = __shfl(jv[(ST) >> 1], 32 * ((ST) & 1) + 8 * u + (lane >> 3));                           \
            const int jr = jr0 < 0 ? 0 : jr0; /* idle slots of the last step fetch row 0 */                         \
            const unsigned voff = (unsigned)jr * 128u + 16u * (unsigned)(dma_chunk ^ ((u & 1) << 2));              \
            unsigned keep_;                                                                                         \
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"                                     \
                         "buffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0"                              \
                         : "=&s"(keep_)                                                                             \
                         : "v"(voff), "s"(rsrc), "s"(lds_rows + 1024u * u)                                          \
                         : "memory");                                                                               \
        }                                                                                                           \
    }
    int jv[NKS];
    double wv[NKS];
    unsigned long long lm[NKS];
    fpfh_mc_list<NKS>(idx, s, k, lane, jv);
    SF_MC_DMA(0) // in flight while the weights are computed
    const int S = fpfh_mc_weights<NKS, HI>(p4, px, py, pz, k, jv, wv, lm);

    v4i acc[8]; // acc[bb]: column a = bin 16 bb + a, rows = limbs 4 kb .. 4 kb + 3
#pragma unroll
    for (int bb = 0; bb < 8; ++bb) acc[bb] = v4i{0, 0, 0, 0};
    v4i accp = v4i{0, 0, 0, 0};

#pragma unroll
    for (int st = 0; st < 2 * NKS; ++st) {
        if (st * 32 < k) { // wave-uniform
            if (st > 0) SF_MC_DMA(st) // (step 0 was issued before the weights were computed)
            if ((st & 1) == 0) fpfh_mc_limbs(abuf, lane, wv[st >> 1], S); // the two steps of a chunk use the lower / upper 32 columns
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // the DMA pieces have landed
            __builtin_amdgcn_wave_barrier(); // both buffers are private to the wave; LDS operations of a wave stay in order
            const long A = fpfh_mc_a_operand(abuf, a, kb, st); // row a = limb a (rows 8 .. 15: zero)
            const unsigned char *rb = reinterpret_cast<const unsigned char *>(rowbuf);
#pragma unroll
            for (int bb = 0; bb < 8; ++bb) {
                const v2i_t t = __builtin_amdgcn_ds_read_tr8_b64_v2i32(
                    (__attribute__((address_space(3))) v2i_t *)(rb + xoff[bb]));
                const long B = (long)(((unsigned long long)(unsigned)t[1] << 32) | (unsigned)t[0]);
                acc[bb] = __builtin_amdgcn_mfma_i32_16x16x32_i8(A, B, acc[bb], 0, 0, 0);
            }
            if (PADC) accp = __builtin_amdgcn_mfma_i32_16x16x32_i8(A, (long)0x8080808080808080ull, accp, 0, 0, 0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // reads done before the next step's DMA overwrites the rows
            __builtin_amdgcn_wave_barrier();
        }
    }
    // ---- recombination: lane (a, g) holds rows (limbs) 4g .. 4g+3 of column a of acc[bb]; bin = 16 bb + a ----
    int rpad[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) rpad[r] = PADC ? accp[r] : __builtin_amdgcn_update_dpp(0, acc[7][r], 0x150 + 15, 0xf, 0xf, false); // column 15 of bb = 7: bin 127
    // A lane's four limb rows are combined in INTEGER arithmetic first: |R| <= 128 * 128 * 255 < 2^22, so
    // (R[r+1] << 8) + R[r] and the same combination of the padding column stay below 2^30.01, and their difference --
    // (sum_j d_j c_jb of limb r + 1) 2^8 + (that of limb r), each at most 128 * 255 * 255 -- is below 2^31: it fits an
    // int32 exactly (two's-complement wrap-around of an intermediate is harmless).  Two shift-adds and a subtraction per
    // pair of limbs instead of a subtraction, a conversion and a float64 FMA per limb.
    const int pad01 = (rpad[1] << 8) + rpad[0], pad23 = (rpad[3] << 8) + rpad[2];
    const double p0 = ldexp(1.0, 32 * kb - S); // 2^(8 (4 g) - S)
    const double f0 = p0, f2 = p0 * 65536.0;
    const double kd = (double)k;
    double inv_k = __builtin_amdgcn_rcp(kd);
    inv_k = __builtin_fma(inv_k, __builtin_fma(-kd, inv_k, 1.0), inv_k);
    inv_k = __builtin_fma(inv_k, __builtin_fma(-kd, inv_k, 1.0), inv_k);
    // partial sums over this lane's four limbs, for its eight bins
    double part[8];
#pragma unroll
    for (int bb = 0; bb < 8; ++bb) {
        const int e01 = ((acc[bb][1] << 8) + acc[bb][0]) - pad01;
        const int e23 = ((acc[bb][3] << 8) + acc[bb][2]) - pad23;
        part[bb] = __builtin_fma((double)e23, f2, (double)e01 * f0);
    }
    // sum over the four limb groups g = kb, transposing as we go: after the exchange with lane ^ 32 a lane keeps only
    // the blocks bb = 4 (kb >> 1) + {0..3}, after the one with lane ^ 16 only bb = 4 (kb >> 1) + 2 (kb & 1) + {0, 1}
    // -- the two bins it writes
    // gfx950's lane-swap instructions do the exchange AND the selection in one go, without the LDS: v_permlane32_swap(a, b)
    // leaves a = {a[0..31], b[0..31]}, b = {a[32..63], b[32..63]}, so a + b is "my half's block plus the partner's" in both
    // halves; v_permlane16_swap does the same between the 16-lane rows 0/1 and 2/3 (tools/ubench/permlane_swap.hip).
    double keep[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        double a_ = part[u], b_ = part[4 + u];
        sf_lane_swap<32>(a_, b_);
        keep[u] = a_ + b_;
    }
    double a0 = keep[0], b0_ = keep[2], a1 = keep[1], b1_ = keep[3];
    sf_lane_swap<16>(a0, b0_);
    sf_lane_swap<16>(a1, b1_);
    const double vsel0 = a0 + b0_, vsel1 = a1 + b1_;
    {
        const int bb0 = 4 * (kb >> 1) + 2 * (kb & 1);
        const int b0 = 16 * bb0 + a, b1 = b0 + 16;
        double *o = out + q * (int64_t)W.nb3;
        fpfh_mc_zero_outside(o, W, lane);
        o += W.win_lo;
        // count / k (the keypoint's own SPFH term, fpfh.py:88-90) through the reciprocal already at hand and one residual
        // step -- the closing step of a division: correctly rounded for these small integers at a tenth of the instructions
        const double c0 = (double)(own0 ^ 128u), c1 = (double)(own1 ^ 128u);
        double s0 = c0 * inv_k, s1 = c1 * inv_k;
        s0 = __builtin_fma(__builtin_fma(-s0, kd, c0), inv_k, s0);
        s1 = __builtin_fma(__builtin_fma(-s1, kd, c1), inv_k, s1);
        double h0 = 0.0, h1 = 0.0;
        if (HI) fpfh_mc_hi<NKS>(hi, jv, wv, lm, b0, b1, h0, h1);
        const double r0 = s0 + (HI ? vsel0 + h0 : vsel0) * inv_k, r1 = s1 + (HI ? vsel1 + h1 : vsel1) * inv_k;
        if (W.win_len == W.nb3 && W.win_len <= 126) { // (wave-uniform)
            fpfh_mc_store_row(o, W.win_len, b0, r0, b1, r1, reinterpret_cast<double *>(rowbuf), lane);
        } else {
            if (b0 < W.win_len) sf_store_stream(o + b0, r0);
            if (b1 < W.win_len) sf_store_stream(o + b1, r1);
        }
    }
}

#undef SF_MC_DMA

// --------------------------------------------------------------------------------------------------
// The same contraction when at most TWO of the table's eight 16-bin blocks hold any count (blocks b0, b1; K6 keeps the OR
// of the non-empty blocks of every row it writes).  Per step of 32 neighbours ONE LDS-DMA instruction fetches the two live
// 16-byte chunks of each row (lane l: row l >> 1 ... see the mapping below) instead of four instructions for the whole
// rows, two transposing reads + two MFMAs replace eight, and the padding column that removes the -128 bias is an MFMA
// against a constant operand (every byte 0x80) instead of a column read from the table.  Results are those of the full
// kernel bit for bit: the integer sums of the live bins are the same sums, and a dead bin's sum is exactly its bias.
// --------------------------------------------------------------------------------------------------
// PACKED: the two chunks come from K6's packed copy of the table (`rows`: 32 bytes per row = {chunk b0, chunk b1}) instead
// of the table itself (`rows` = counts, 128 bytes per row): four rows per cache line for the gather.
template <int NKS, bool PACKED, bool HI>
__device__ __forceinline__ void fpfh_mc_body_sparse(const double *__restrict__ rec, const int64_t *__restrict__ offset,
                                                    const int32_t *__restrict__ cnt, const int32_t *__restrict__ idx,
                                                    int64_t nbrs_begin, const int32_t *__restrict__ kp_pos, sf_bin_window W,
                                                    const uint8_t *__restrict__ counts, const uint8_t *__restrict__ rows,
                                                    unsigned rows_bytes, const double *__restrict__ p4,
                                                    double *__restrict__ out, int64_t q, int b0, int b1,
                                                    unsigned *rowbuf /* 4 KB: four steps of 1 KB */, unsigned char *abuf /* 576 B */,
                                                    const uint8_t *__restrict__ hi, int limit)
{
    const int lane = threadIdx.x & 63;
    const int64_t i = kp_pos ? (int64_t)kp_pos[q] : nbrs_begin + q;
    const int64_t slot = i - nbrs_begin;
    const int64_t s = offset[slot];
    const int k = cnt[slot];
    if (sf_uniform(k) > limit) return; // (the second launch's keypoint)
    const double px = rec[6 * i + 0], py = rec[6 * i + 1], pz = rec[6 * i + 2];
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(rows), 0, (int)rows_bytes, 0x00020000);
    const int a = lane & 15, kb = lane >> 4;
    // the keypoint's own counts of the two bins this lane writes: requested HERE, with the list -- the "memory" clobbers of the
    // DMA statements below pin a load where it is written, and at the end of the kernel (where the values are used) it was a
    // memory round trip of its own on every wave's critical path
    const int bb0 = 4 * (kb >> 1) + 2 * (kb & 1);
    const int o0 = 16 * bb0 + a, o1 = o0 + 16; // as in the full kernel
    const unsigned own0 = counts[i * 128 + o0], own1 = counts[i * 128 + o1];
    if (lane == 0) *reinterpret_cast<unsigned long long *>(abuf + 512) = 0ull; // the "limbs 8 .. 15" every A operand reads
    // LDS image of a step: 64 pieces of 16 bytes, piece P = 32 (kb >> 1) + 16 u + 8 (kb & 1) + q holds chunk b_u of row
    // 8 kb + q -- the DMA writes its lanes' pieces back to back, so lane P fetches exactly that; the 32 pieces a
    // transposing read of block u touches per half-wave then lie in 32 different 8-byte bank pairs.
    const int d_u = (lane >> 4) & 1, d_row = 16 * (lane >> 5) + (lane & 15); // = 8 (2 (P >> 5) + ((P >> 3) & 1)) + (P & 7)
    const unsigned dma_chunk16 = PACKED ? 16u * (unsigned)d_u : 16u * (unsigned)(d_u ? b1 : b0);
    constexpr unsigned ROW_BYTES = PACKED ? 32u : 128u;
    const int rd_piece = 32 * (kb >> 1) + 8 * (kb & 1) + (a >> 1); // + 16 u ; bytes 8 (a & 1) .. of the piece
    const int rd0 = 16 * rd_piece + 8 * (a & 1), rd1 = rd0 + 256;
    const unsigned lds_rows = (unsigned)__builtin_amdgcn_readfirstlane(
        (int)(unsigned)(size_t)(__attribute__((address_space(3))) unsigned *)rowbuf);
#define SF_MCS_DMA(ST)                                                                                              \
    {                                                                                                               \
        const int jr0 = __shfl(jv[((ST) >> 1) < NKS ? ((ST) >> 1) : 0], 32 * ((ST) & 1) + d_row);                   \
        const int jr = jr0 < 0 ? 0 : jr0; /* idle slots of the last step fetch row 0 */                             \
        const unsigned voff = (unsigned)jr * ROW_BYTES + dma_chunk16;                                               \
        unsigned keep_;                                                                                             \
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"                                         \
                     "buffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0"                                  \
                     : "=&s"(keep_)                                                                                 \
                     : "v"(voff), "s"(rsrc), "s"(lds_rows + 1024u * (unsigned)((ST) & 3))                           \
                     : "memory");                                                                                   \
    }
    int jv[NKS];
    double wv[NKS];
    unsigned long long lm[NKS];
    fpfh_mc_list<NKS>(idx, s, k, lane, jv);
    // The image of a step is 1 KB, so the 4 KB row buffer holds FOUR steps: the rows of the first two chunks (128
    // neighbours -- the whole list for nine keypoints in ten) are all requested here, in flight while the weights are
    // computed, and the step loop below never waits on memory again; only a third / fourth chunk re-uses the buffer.
    SF_MCS_DMA(0)
    if (32 < k) SF_MCS_DMA(1)
    if (NKS >= 2 && 64 < k) SF_MCS_DMA(2)
    if (NKS >= 2 && 96 < k) SF_MCS_DMA(3)
    const int S = fpfh_mc_weights<NKS, HI>(p4, px, py, pz, k, jv, wv, lm);

    v4i acc0 = v4i{0, 0, 0, 0}, acc1 = v4i{0, 0, 0, 0}, accp = v4i{0, 0, 0, 0}; // blocks b0, b1, and the padding column
    const long Bpad = (long)0x8080808080808080ull; // eight neighbours' padding bin: -128 each

#pragma unroll
    for (int st = 0; st < 2 * NKS; ++st) {
        if (st * 32 < k) { // wave-uniform
            if (st >= 4) SF_MCS_DMA(st) // (steps 0 .. 3 were requested up front; by now their regions have been consumed)
            if ((st & 1) == 0) fpfh_mc_limbs(abuf, lane, wv[st >> 1], S);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // the DMA pieces have landed
            __builtin_amdgcn_wave_barrier();
            const long A = fpfh_mc_a_operand(abuf, a, kb, st); // row a = limb a (rows 8 .. 15: zero)
            const unsigned char *rb = reinterpret_cast<const unsigned char *>(rowbuf);
            const v2i_t t0 = __builtin_amdgcn_ds_read_tr8_b64_v2i32((__attribute__((address_space(3))) v2i_t *)(rb + rd0 + 1024 * (st & 3)));
            const v2i_t t1 = __builtin_amdgcn_ds_read_tr8_b64_v2i32((__attribute__((address_space(3))) v2i_t *)(rb + rd1 + 1024 * (st & 3)));
            accp = __builtin_amdgcn_mfma_i32_16x16x32_i8(A, Bpad, accp, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_i32_16x16x32_i8(A, (long)(((unsigned long long)(unsigned)t0[1] << 32) | (unsigned)t0[0]), acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_i32_16x16x32_i8(A, (long)(((unsigned long long)(unsigned)t1[1] << 32) | (unsigned)t1[0]), acc1, 0, 0, 0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // reads done before the next step's DMA overwrites the pieces
            __builtin_amdgcn_wave_barrier();
        }
    }
    // ---- recombination: lane (a, g) holds limbs 4g .. 4g + 3 of bin 16 b_u + a in acc_u, and of the padding column in accp
    //      (every column of accp is the same sum) ----
    const int pad01 = (accp[1] << 8) + accp[0], pad23 = (accp[3] << 8) + accp[2];
    const double p0 = ldexp(1.0, 32 * kb - S); // 2^(8 (4 g) - S)
    const double f0 = p0, f2 = p0 * 65536.0;
    const double kd = (double)k;
    double inv_k = __builtin_amdgcn_rcp(kd);
    inv_k = __builtin_fma(inv_k, __builtin_fma(-kd, inv_k, 1.0), inv_k);
    inv_k = __builtin_fma(inv_k, __builtin_fma(-kd, inv_k, 1.0), inv_k);
    double part0, part1;
    {
        const int e01 = ((acc0[1] << 8) + acc0[0]) - pad01, e23 = ((acc0[3] << 8) + acc0[2]) - pad23;
        part0 = __builtin_fma((double)e23, f2, (double)e01 * f0);
        const int g01 = ((acc1[1] << 8) + acc1[0]) - pad01, g23 = ((acc1[3] << 8) + acc1[2]) - pad23;
        part1 = __builtin_fma((double)g23, f2, (double)g01 * f0);
    }
    // Sum over the four limb groups in the SAME order as the full kernel -- (g + (g ^ 2)) first, then the two halves of that
    // -- so that the rounding, hence every bit of the row, is the full kernel's: after the 32-swap the lower half of the
    // wave holds block b0's pair sums and the upper half block b1's, after the 16-swap both rows of a half hold the total.
    sf_lane_swap<32>(part0, part1);
    double v = part0 + part1;     // lanes 0..31: block b0, groups (g, g + 2) ; lanes 32..63: block b1
    double w = v;
    sf_lane_swap<16>(v, w);
    const double tot = v + w;     // all four groups; lanes 0..31: bin 16 b0 + a, lanes 32..63: bin 16 b1 + a
    // every lane needs both totals for the row it helps write: the other half's through one more 32-swap
    double mine = tot, other = tot;
    sf_lane_swap<32>(mine, other); // lower half: mine = own (b0), other = b1's ; upper half: mine = b0's, other = own (b1)
    const double tot0 = mine, tot1 = other; // (after the swap `mine` is block b0's total in both halves, `other` b1's)
    {
        double *o = out + q * (int64_t)W.nb3;
        fpfh_mc_zero_outside(o, W, lane);
        o += W.win_lo;
        const double c0 = (double)(own0 ^ 128u), c1 = (double)(own1 ^ 128u);
        double s0 = c0 * inv_k, s1 = c1 * inv_k;
        s0 = __builtin_fma(__builtin_fma(-s0, kd, c0), inv_k, s0);
        s1 = __builtin_fma(__builtin_fma(-s1, kd, c1), inv_k, s1);
        const double v0 = bb0 == b0 ? tot0 : (bb0 == b1 ? tot1 : 0.0);
        const double v1 = bb0 + 1 == b0 ? tot0 : (bb0 + 1 == b1 ? tot1 : 0.0);
        double h0 = 0.0, h1 = 0.0;
        if (HI) fpfh_mc_hi<NKS>(hi, jv, wv, lm, o0, o1, h0, h1);
        const double r0 = s0 + (HI ? v0 + h0 : v0) * inv_k, r1 = s1 + (HI ? v1 + h1 : v1) * inv_k;
        if (W.win_len == W.nb3 && W.win_len <= 126) { // (wave-uniform)
            fpfh_mc_store_row(o, W.win_len, o0, r0, o1, r1, reinterpret_cast<double *>(rowbuf), lane);
        } else {
            if (o0 < W.win_len) sf_store_stream(o + o0, r0);
            if (o1 < W.win_len) sf_store_stream(o + o1, r1);
        }
    }
}

#undef SF_MCS_DMA

// --------------------------------------------------------------------------------------------------
// The matrix-core contraction for lists of MORE than 255 points (round 5; until then the vector ALU served them at 3.4 x the
// cost per pair: k_fpfh_tail).  Nothing in the arithmetic above is tied to 255 neighbours but the registers that hold a list
// and the int32 recombination: |limb x (count - 128)| <= 2^14 per neighbour, so the int32 accumulators hold EXACT sums for any
// list a byte table can have (k <= 65 535 -> < 2^30).  So: a first pass over the list finds the largest weight (the fixed-point
// exponent), the second walks it in super-chunks of SC x 64 neighbours -- index loads, record gathers and weights of a
// super-chunk at once, then its 2 SC steps of 32 neighbours through the same LDS-DMA / transposing reads / MFMAs -- and the
// limb sums are recombined once, in float64 (the integer shift-adds of the short form would overflow; on a short list the
// two give the same bits).  A list of at most SC x 64 points keeps entries and weights in registers between the passes.
// High bytes of neighbours that have more than 255 neighbours of their own: fpfh_mc_hi per super-chunk, list order.
// The sparse and the full form mirror each other operation by operation, as the short forms do: same bits.
// --------------------------------------------------------------------------------------------------
template <int SC>
__device__ __forceinline__ void fpfh_mcl_load(const int32_t *__restrict__ idx, const double *__restrict__ p4, int64_t s, int k,
                                              int base, int lane, double px, double py, double pz, int (&jv)[SC], double (&wv)[SC],
                                              unsigned long long (&lm)[SC])
{
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
#pragma unroll
    for (int c = 0; c < SC; ++c) {
        const double cx = u0[c].x - px, cy = u0[c].y - py, cz = u1[c].x - pz;
        const double d2 = (cx * cx + cy * cy) + cz * cz;
        const double kd = u1[c].y, xx = d2 * (kd * kd);
        const double y0 = __builtin_amdgcn_rsq(xx);
        const double y1 = __builtin_fma(0.5 * y0, __builtin_fma(-(xx * y0), y0, 1.0), y0);
        const double y2 = __builtin_fma(0.5 * y1, __builtin_fma(-(xx * y1), y1, 1.0), y1);
        const bool on = jv[c] >= 0;
        wv[c] = (on && d2 > 0.0) ? y2 : 0.0; // (weight 0 past the end and at distance 0, fpfh.py:110-114: all limbs 0)
        lm[c] = __ballot(wv[c] > 0.0 && kd > 255.0);
        jv[c] = on ? jv[c] : 0; // (past the end of the list row 0 is fetched, under weight 0)
    }
}

// limb sums -> float64, for one pair of accumulator registers against the padding column's
__device__ __forceinline__ double fpfh_mcl_pair(int hi_limb, int lo_limb, int pad_hi, int pad_lo)
{
    return __builtin_fma((double)hi_limb - (double)pad_hi, 256.0, (double)lo_limb - (double)pad_lo); // exact: < 2^41
}

template <bool PACKED, int SC>
__device__ __forceinline__ void fpfh_mcl_body_sparse(const double *__restrict__ rec, const int64_t *__restrict__ offset,
                                                     const int32_t *__restrict__ cnt, const int32_t *__restrict__ idx,
                                                     int64_t nbrs_begin, const int32_t *__restrict__ kp_pos, sf_bin_window W,
                                                     const uint8_t *__restrict__ counts, const uint8_t *__restrict__ rows,
                                                     unsigned rows_bytes, const double *__restrict__ p4,
                                                     double *__restrict__ out, int64_t q, int b0, int b1,
                                                     unsigned *rowbuf /* 4 KB: four steps of 1 KB */, unsigned char *abuf /* 576 B */,
                                                     const uint8_t *__restrict__ hi, int limit)
{
    const int lane = threadIdx.x & 63;
    const int64_t i = kp_pos ? (int64_t)kp_pos[q] : nbrs_begin + q;
    const int64_t slot = i - nbrs_begin;
    const int k = sf_uniform(cnt[slot]);
    if (k <= limit) return; // (the main launch's keypoint)
    const int64_t s = offset[slot];
    const double px = rec[6 * i + 0], py = rec[6 * i + 1], pz = rec[6 * i + 2];
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(rows), 0, (int)rows_bytes, 0x00020000);
    const int a = lane & 15, kb = lane >> 4;
    const int bb0 = 4 * (kb >> 1) + 2 * (kb & 1);
    const int o0 = 16 * bb0 + a, o1 = o0 + 16; // the two bins this lane writes, as in the short forms
    unsigned own0 = (unsigned)counts[i * 128 + o0] ^ 128u, own1 = (unsigned)counts[i * 128 + o1] ^ 128u;
    if (k > 255) { own0 += 256u * (unsigned)hi[i * 128 + o0]; own1 += 256u * (unsigned)hi[i * 128 + o1]; }
    if (lane == 0) *reinterpret_cast<unsigned long long *>(abuf + 512) = 0ull; // the "limbs 8 .. 15" every A operand reads
    // LDS image of a step and the DMA's lane mapping: fpfh_mc_body_sparse
    const int d_u = (lane >> 4) & 1, d_row = 16 * (lane >> 5) + (lane & 15);
    const unsigned dma_chunk16 = PACKED ? 16u * (unsigned)d_u : 16u * (unsigned)(d_u ? b1 : b0);
    constexpr unsigned ROW_BYTES = PACKED ? 32u : 128u;
    const int rd_piece = 32 * (kb >> 1) + 8 * (kb & 1) + (a >> 1);
    const int rd0 = 16 * rd_piece + 8 * (a & 1), rd1 = rd0 + 256;
    const unsigned lds_rows = (unsigned)__builtin_amdgcn_readfirstlane(
        (int)(unsigned)(size_t)(__attribute__((address_space(3))) unsigned *)rowbuf);
#define SF_MCL_DMA(ST)                                                                                              \
    {                                                                                                               \
        const int jr = __shfl(jv[((ST) >> 1) < SC ? ((ST) >> 1) : 0], 32 * ((ST) & 1) + d_row);                     \
        const unsigned voff = (unsigned)jr * ROW_BYTES + dma_chunk16;                                               \
        unsigned keep_;                                                                                             \
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"                                         \
                     "buffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0"                                  \
                     : "=&s"(keep_)                                                                                 \
                     : "v"(voff), "s"(rsrc), "s"(lds_rows + 1024u * (unsigned)((ST) & 3))                           \
                     : "memory");                                                                                   \
    }
    int jv[SC];
    double wv[SC];
    unsigned long long lm[SC];
    // ---- pass 0: the largest weight -> the fixed-point exponent ----
    double wmax = 0.0;
    for (int base = 0; base < k; base += 64 * SC) {
        fpfh_mcl_load<SC>(idx, p4, s, k, base, lane, px, py, pz, jv, wv, lm);
#pragma unroll
        for (int c = 0; c < SC; ++c) wmax = fmax(wmax, wv[c]);
    }
    wmax = sf_wave_max_nonneg(wmax);
    const int e2 = wmax > 0.0 ? (int)((__double2hiint(wmax) >> 20) & 0x7ff) - 1023 : 0;
    const int S = 61 - e2; // W = floor(w 2^S) < 2^62

    v4i acc0 = v4i{0, 0, 0, 0}, acc1 = v4i{0, 0, 0, 0}, accp = v4i{0, 0, 0, 0}; // blocks b0, b1, and the padding column
    const long Bpad = (long)0x8080808080808080ull;
    double h0 = 0.0, h1 = 0.0;
    for (int base = 0; base < k; base += 64 * SC) {
        if (k > 64 * SC) fpfh_mcl_load<SC>(idx, p4, s, k, base, lane, px, py, pz, jv, wv, lm); // (else: still held from pass 0)
        const int left = k - base;
        const int nst = left >= 64 * SC ? 2 * SC : (left + 31) >> 5; // steps of this super-chunk (wave-uniform)
        // The 4 KB row buffer holds four steps: up to four are in flight, and a step's slot is refilled as soon as its reads
        // have returned -- the loop waits for the OLDEST request only (vector-memory requests complete in order).
#pragma unroll
        for (int st = 0; st < 4; ++st)
            if (st < nst) SF_MCL_DMA(st)
#pragma unroll
        for (int st = 0; st < 2 * SC; ++st) {
            if (st < nst) { // wave-uniform
                if ((st & 1) == 0) fpfh_mc_limbs(abuf, lane, wv[st >> 1], S);
                if (st + 3 < nst) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
                else if (st + 2 < nst) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
                else if (st + 1 < nst) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_wave_barrier();
                const long A = fpfh_mc_a_operand(abuf, a, kb, st);
                const unsigned char *rb = reinterpret_cast<const unsigned char *>(rowbuf);
                const v2i_t t0 = __builtin_amdgcn_ds_read_tr8_b64_v2i32((__attribute__((address_space(3))) v2i_t *)(rb + rd0 + 1024 * (st & 3)));
                const v2i_t t1 = __builtin_amdgcn_ds_read_tr8_b64_v2i32((__attribute__((address_space(3))) v2i_t *)(rb + rd1 + 1024 * (st & 3)));
                accp = __builtin_amdgcn_mfma_i32_16x16x32_i8(A, Bpad, accp, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_i32_16x16x32_i8(A, (long)(((unsigned long long)(unsigned)t0[1] << 32) | (unsigned)t0[0]), acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_i32_16x16x32_i8(A, (long)(((unsigned long long)(unsigned)t1[1] << 32) | (unsigned)t1[0]), acc1, 0, 0, 0);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // reads done before the slot is refilled
                __builtin_amdgcn_wave_barrier();
                if (st + 4 < nst) SF_MCL_DMA(st + 4)
            }
        }
        fpfh_mc_hi<SC>(hi, jv, wv, lm, o0, o1, h0, h1);
    }
    // ---- recombination: fpfh_mc_body_sparse's, the limb pairs formed in float64 ----
    const double p0 = ldexp(1.0, 32 * kb - S);
    const double f0 = p0, f2 = p0 * 65536.0;
    const double kd = (double)k;
    double inv_k = __builtin_amdgcn_rcp(kd);
    inv_k = __builtin_fma(inv_k, __builtin_fma(-kd, inv_k, 1.0), inv_k);
    inv_k = __builtin_fma(inv_k, __builtin_fma(-kd, inv_k, 1.0), inv_k);
    double part0 = __builtin_fma(fpfh_mcl_pair(acc0[3], acc0[2], accp[3], accp[2]), f2, fpfh_mcl_pair(acc0[1], acc0[0], accp[1], accp[0]) * f0);
    double part1 = __builtin_fma(fpfh_mcl_pair(acc1[3], acc1[2], accp[3], accp[2]), f2, fpfh_mcl_pair(acc1[1], acc1[0], accp[1], accp[0]) * f0);
    sf_lane_swap<32>(part0, part1);
    double v = part0 + part1;
    double w = v;
    sf_lane_swap<16>(v, w);
    const double tot = v + w;
    double mine = tot, other = tot;
    sf_lane_swap<32>(mine, other);
    const double tot0 = mine, tot1 = other;
    {
        double *o = out + q * (int64_t)W.nb3;
        fpfh_mc_zero_outside(o, W, lane);
        o += W.win_lo;
        const double v0 = bb0 == b0 ? tot0 : (bb0 == b1 ? tot1 : 0.0);
        const double v1 = bb0 + 1 == b0 ? tot0 : (bb0 + 1 == b1 ? tot1 : 0.0);
        const double r0 = (double)own0 / kd + (v0 + h0) * inv_k, r1 = (double)own1 / kd + (v1 + h1) * inv_k; // spfh[kp] + sum / len(neighbourhood)  (fpfh.py:109-115)
        if (W.win_len == W.nb3 && W.win_len <= 126) { // (wave-uniform)
            fpfh_mc_store_row(o, W.win_len, o0, r0, o1, r1, reinterpret_cast<double *>(rowbuf), lane);
        } else {
            if (o0 < W.win_len) sf_store_stream(o + o0, r0);
            if (o1 < W.win_len) sf_store_stream(o + o1, r1);
        }
    }
#undef SF_MCL_DMA
}

// ... and on a table with more than two live blocks: whole 128-byte rows, one step (4 KB) in LDS at a time, eight MFMAs per step
template <int SC, bool PADC = false>
__device__ __forceinline__ void fpfh_mcl_body(const double *__restrict__ rec, const int64_t *__restrict__ offset,
                                              const int32_t *__restrict__ cnt, const int32_t *__restrict__ idx,
                                              int64_t nbrs_begin, const int32_t *__restrict__ kp_pos, sf_bin_window W,
                                              const uint8_t *__restrict__ counts, unsigned table_bytes,
                                              const double *__restrict__ p4, double *__restrict__ out, int64_t q,
                                              unsigned *rowbuf /* 4 KB */, unsigned char *abuf /* 576 B */,
                                              const uint8_t *__restrict__ hi, int limit)
{
    const int lane = threadIdx.x & 63;
    const int64_t i = kp_pos ? (int64_t)kp_pos[q] : nbrs_begin + q;
    const int64_t slot = i - nbrs_begin;
    const int k = sf_uniform(cnt[slot]);
    if (k <= limit) return; // (the main launch's keypoint)
    const int64_t s = offset[slot];
    const double px = rec[6 * i + 0], py = rec[6 * i + 1], pz = rec[6 * i + 2];
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(counts), 0, (int)table_bytes, 0x00020000);
    const int a = lane & 15, kb = lane >> 4;
    const int bb0 = 4 * (kb >> 1) + 2 * (kb & 1);
    const int o0 = 16 * bb0 + a, o1 = o0 + 16;
    unsigned own0 = (unsigned)counts[i * 128 + o0] ^ 128u, own1 = (unsigned)counts[i * 128 + o1] ^ 128u;
    if (k > 255) { own0 += 256u * (unsigned)hi[i * 128 + o0]; own1 += 256u * (unsigned)hi[i * 128 + o1]; }
    if (lane == 0) *reinterpret_cast<unsigned long long *>(abuf + 512) = 0ull;
    const int frow = ((a >> 2) & 3) | ((kb & 1) << 2);
    const int rd_base = (8 * kb + (a >> 1)) * 128 + 8 * (a & 1);
    int xoff[8];
#pragma unroll
    for (int bb = 0; bb < 8; ++bb) xoff[bb] = rd_base + 16 * (bb ^ frow);
    const int dma_chunk = (lane & 7) ^ ((lane >> 4) & 3);
    const unsigned lds_rows = (unsigned)__builtin_amdgcn_readfirstlane(
        (int)(unsigned)(size_t)(__attribute__((address_space(3))) unsigned *)rowbuf);
#define SF_MCLF_DMA(ST)                                                                                             \
    {                                                                                                               \
        _Pragma("unroll") for (int u = 0; u < 4; ++u) {                                                             \
            const int jr = __shfl(jv[(ST) >> 1], 32 * ((ST) & 1) + 8 * u + (lane >> 3));                            \
            const unsigned voff = (unsigned)jr * 128u + 16u * (unsigned)(dma_chunk ^ ((u & 1) << 2));              \
            unsigned keep_;                                                                                         \
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"                                     \
                         "buffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0"                              \
                         : "=&s"(keep_)                                                                             \
                         : "v"(voff), "s"(rsrc), "s"(lds_rows + 1024u * u)                                          \
                         : "memory");                                                                               \
        }                                                                                                           \
    }
    int jv[SC];
    double wv[SC];
    unsigned long long lm[SC];
    double wmax = 0.0;
    for (int base = 0; base < k; base += 64 * SC) {
        fpfh_mcl_load<SC>(idx, p4, s, k, base, lane, px, py, pz, jv, wv, lm);
#pragma unroll
        for (int c = 0; c < SC; ++c) wmax = fmax(wmax, wv[c]);
    }
    wmax = sf_wave_max_nonneg(wmax);
    const int e2 = wmax > 0.0 ? (int)((__double2hiint(wmax) >> 20) & 0x7ff) - 1023 : 0;
    const int S = 61 - e2;
    v4i acc[8];
#pragma unroll
    for (int bb = 0; bb < 8; ++bb) acc[bb] = v4i{0, 0, 0, 0};
    v4i accp = v4i{0, 0, 0, 0};
    double h0 = 0.0, h1 = 0.0;
    for (int base = 0; base < k; base += 64 * SC) {
        if (k > 64 * SC) fpfh_mcl_load<SC>(idx, p4, s, k, base, lane, px, py, pz, jv, wv, lm);
        const int left = k - base;
        const int nst = left >= 64 * SC ? 2 * SC : (left + 31) >> 5;
#pragma unroll
        for (int st = 0; st < 2 * SC; ++st) {
            if (st < nst) { // wave-uniform
                SF_MCLF_DMA(st)
                if ((st & 1) == 0) fpfh_mc_limbs(abuf, lane, wv[st >> 1], S);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_wave_barrier();
                const long A = fpfh_mc_a_operand(abuf, a, kb, st);
                const unsigned char *rb = reinterpret_cast<const unsigned char *>(rowbuf);
#pragma unroll
                for (int bb = 0; bb < 8; ++bb) {
                    const v2i_t t = __builtin_amdgcn_ds_read_tr8_b64_v2i32((__attribute__((address_space(3))) v2i_t *)(rb + xoff[bb]));
                    const long B = (long)(((unsigned long long)(unsigned)t[1] << 32) | (unsigned)t[0]);
                    acc[bb] = __builtin_amdgcn_mfma_i32_16x16x32_i8(A, B, acc[bb], 0, 0, 0);
                }
                if (PADC) accp = __builtin_amdgcn_mfma_i32_16x16x32_i8(A, (long)0x8080808080808080ull, accp, 0, 0, 0);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_wave_barrier();
            }
        }
        fpfh_mc_hi<SC>(hi, jv, wv, lm, o0, o1, h0, h1);
    }
    // ---- recombination: fpfh_mc_body's, the limb pairs formed in float64 ----
    int rpad[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) rpad[r] = PADC ? accp[r] : __builtin_amdgcn_update_dpp(0, acc[7][r], 0x150 + 15, 0xf, 0xf, false); // bin 127: the padding column
    const double p0 = ldexp(1.0, 32 * kb - S);
    const double f0 = p0, f2 = p0 * 65536.0;
    const double kd = (double)k;
    double inv_k = __builtin_amdgcn_rcp(kd);
    inv_k = __builtin_fma(inv_k, __builtin_fma(-kd, inv_k, 1.0), inv_k);
    inv_k = __builtin_fma(inv_k, __builtin_fma(-kd, inv_k, 1.0), inv_k);
    double part[8];
#pragma unroll
    for (int bb = 0; bb < 8; ++bb)
        part[bb] = __builtin_fma(fpfh_mcl_pair(acc[bb][3], acc[bb][2], rpad[3], rpad[2]), f2, fpfh_mcl_pair(acc[bb][1], acc[bb][0], rpad[1], rpad[0]) * f0);
    double keep[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        double a_ = part[u], b_ = part[4 + u];
        sf_lane_swap<32>(a_, b_);
        keep[u] = a_ + b_;
    }
    double a0 = keep[0], b0_ = keep[2], a1 = keep[1], b1_ = keep[3];
    sf_lane_swap<16>(a0, b0_);
    sf_lane_swap<16>(a1, b1_);
    const double vsel0 = a0 + b0_, vsel1 = a1 + b1_;
    {
        double *o = out + q * (int64_t)W.nb3;
        fpfh_mc_zero_outside(o, W, lane);
        o += W.win_lo;
        const double r0 = (double)own0 / kd + (vsel0 + h0) * inv_k, r1 = (double)own1 / kd + (vsel1 + h1) * inv_k;
        if (W.win_len == W.nb3 && W.win_len <= 126) { // (wave-uniform)
            fpfh_mc_store_row(o, W.win_len, o0, r0, o1, r1, reinterpret_cast<double *>(rowbuf), lane);
        } else {
            if (o0 < W.win_len) sf_store_stream(o + o0, r0);
            if (o1 < W.win_len) sf_store_stream(o + o1, r1);
        }
    }
#undef SF_MCLF_DMA
}
