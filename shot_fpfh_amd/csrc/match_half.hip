// match_half.hip -- K8 pre-filter: the arg-min of cdist(a, b) with an FP16 matrix-core pass that only PRUNES,
// followed by the reference's own float64 arithmetic on the few pairs that survive.
//
// Replaces scipy cdist + argmin (matching.py:47-52, 164-168) for large problems, in front of the FP64 GEMM of
// match_gemm.hip (whose rate, 2 m1 m2 d flop at <= 78.6 TFLOP/s, is what bounds config 4: 1M x 1M x 352).
//
//   1. both descriptor sets are scaled by a power of two and rounded to FP16 (k_half_convert); per row the
//      EXACT quantisation error ||a_i - a'_i||, the quantised norm ||a'_i|| and ||a_i||^2 are kept in float64.
//   2. k_match_half: one pass of v_mfma_f32_32x32x16_f16 over all pairs gives approximate ranking keys
//          k(i, j) = ||b_j||^2 - 2 a'_i . b'_j
//      Every wave owns 32 rows of `a` (fragments resident in registers for the whole pass) and keeps, per row, a
//      threshold thr_i = (smallest key seen so far) + W_i.  A key above the threshold is dropped; one below it
//      is appended to the row's candidate list and may lower the threshold.  After t column tiles that happens
//      with probability ~ 1/t, so the pass is MFMA + one v_max3 per 32 pairs (the threshold is the start value of
//      the accumulator, see k_match_half).
//   3. k_half_final: per row, the candidates still within W_i of the final minimum (typically 1-3) get the
//      reference's distance -- sequential float64 sum, square root -- and the smallest (lowest column on
//      ties) wins, exactly scipy's first-minimum rule.
//
// Exactness.  Let eps_i bound |k(i, j) - key(i, j)| over j, with key the exact ||b_j||^2 - 2 a_i . b_j:
//     eps_i = 2 (ea_i Bmax + qa_i EBmax + gamma qa_i QBmax) + gamma (nbmax + 4 qa_i QBmax)
//     ea_i = ||a_i - a'_i||, qa_i = ||a'_i||, Bmax = max ||b_j||, EBmax = max ||b_j - b'_j||, QBmax = max ||b'_j||,
//     nbmax = Bmax^2, gamma = (d + 32) 2^-22
// (Cauchy-Schwarz on (a - a').b + a'.(b - b'); FP16 x FP16 products are exact in FP32; gamma covers the FP32
// accumulation of the matrix core however it rounds, the second gamma term because the accumulator starts at
// the threshold rather than at zero).  The reference's arg-min j* has an exact key no larger than that of the
// column j1 that set the final threshold, plus float64 rounding (1e-12 relative, folded into eps).  Hence
// k(i, j*) <= k(i, j1) + 2 eps_i, and with W_i = 2 eps_i + 12 eta_i (eta_i: the FP32 rounding of the key /
// threshold arithmetic in the epilogue) j* passes the threshold test at the moment it is scanned and the final
// filter.  So the candidate set always contains the reference's arg-min and everything that ties with
// it; step 3 then decides in the reference's arithmetic.  Rows whose list overflows (more than `cap` near-
// minimal columns, e.g. many duplicated descriptors), or that hold non-finite values, go to the FP64 path
// (match_gemm.hip, which has its own exact slow path).  The result equals the exact kernel's for every
// input; only the amount of work depends on the data.
// Roofline: FP16 matrix cores (dense peak ~2.5 PFLOP/s), 2 m1 m2 dpad flop.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <vector>

#include "common.h"
#include "device_util.h"

int sf_match_gemm_f64(sf_ctx *ctx, const double *da, int64_t m1, const double *db, int64_t m2, int64_t d, int64_t *didx,
                      double *ddist, const char *name, int64_t *n_slow, const unsigned char *a_ok,
                      const unsigned char *b_ok); // match_gemm.hip

namespace {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

constexpr int HM = 256; // rows of `a` per workgroup: 8 waves x 32
constexpr int HN = 64;  // columns per LDS tile
constexpr int HCAP = 32; // candidate slots per row

// One wave per row: FP16 image of scale * row (zero padded to dp), and the row's float64 bookkeeping.
__global__ __launch_bounds__(256) void k_half_convert(const double *__restrict__ a, int64_t m, int64_t m_pad, int64_t d,
                                                      int dp, double scale, const unsigned char *__restrict__ ok,
                                                      _Float16 *__restrict__ out, double *__restrict__ err,
                                                      double *__restrict__ qn, double *__restrict__ n2,
                                                      float *__restrict__ n2f, double n2f_scale)
{
    const int lane = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= m_pad) return;
    const bool real = i < m;
    const double inv = 1.0 / scale; // power of two: exact
    double se = 0.0, sq = 0.0, sn = 0.0;
    for (int t = lane; t < dp; t += 64) {
        const double v = (real && t < d) ? a[i * d + t] : 0.0;
        const _Float16 hv = (_Float16)(v * scale);
        if (out) out[i * dp + t] = hv;
        const double back = (double)hv * inv, e = v - back;
        se += e * e;
        sq += back * back;
        sn += v * v;
    }
    se = sf_wave_sum(se); sq = sf_wave_sum(sq); sn = sf_wave_sum(sn);
    if (lane == 0) {
        const bool masked = !real || (ok && !ok[i]);
        err[i] = real ? sqrt(se) : 0.0;
        qn[i] = real ? sqrt(sq) : 0.0;
        n2[i] = real ? sn : 0.0;
        if (n2f) { // +inf keeps masked / padding columns out of every candidate list; rounded to nearest otherwise
            n2f[i] = masked ? INFINITY : (float)(sn * n2f_scale); // power-of-two scale: one rounding
        }
    }
}

// max over i of v[i] (v >= 0; non-finite entries propagate so that the host can refuse them) -> partial[blockIdx]
__global__ void k_half_max(const double *__restrict__ v, int64_t n, double *__restrict__ partial)
{
    double mx = 0.0;
    bool bad = false;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const double x = v[i];
        bad |= !(x <= 1.7976931348623157e308) || !(x >= 0.0); // inf or NaN
        mx = fmax(mx, x);
    }
    if (bad) mx = INFINITY;
    for (int off = 32; off > 0; off >>= 1) mx = fmax(mx, __shfl_xor(mx, off));
    __shared__ double s[4];
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = fmax(fmax(s[0], s[1]), fmax(s[2], s[3]));
}

// W_i of the header, rounded up to float; rows beyond m get 0 (their thresholds start at -inf and never move)
__global__ void k_half_window(const double *__restrict__ ea, const double *__restrict__ qa, const double *__restrict__ na2,
                              int64_t m, int64_t m_pad, double bmax, double ebmax, double qbmax, double nbmax,
                              double gamma, double unit, float *__restrict__ win)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m_pad) return;
    if (i >= m) { win[i] = 0.0f; return; }
    // the accumulator starts at the threshold (|T| <= nbmax + 2 qa qbmax + W), hence the second gamma term
    const double eps = (2.0 * (ea[i] * bmax + qa[i] * ebmax + gamma * qa[i] * qbmax) + gamma * (nbmax + 4.0 * qa[i] * qbmax)) *
                           (1.0 + 1e-6) + 1e-12 * (na2[i] + nbmax);
    const double eta = 1.1920928955078125e-07 * (nbmax + 2.0 * qa[i] * qbmax); // 2^-23 (|key| terms)
    const double w = (2.0 * eps + 12.0 * eta) * (1.0 + 1e-6) * unit; // in accumulator units (unit = 1 / 2s)
    float wf = (float)w;
    if ((double)wf < w) wf = nextafterf(wf, INFINITY);
    win[i] = wf;
}

template <int CTRL>
__device__ __forceinline__ float dpp_f32(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}

// The pass.  ah: m1_pad x DP (m1_pad a multiple of 256), bh: m2_pad x DP (m2_pad a multiple of 64), row-major FP16.
// All keys, thresholds and windows are in "accumulator units": divided by 2s = 2 / (scale_a scale_b), a power of
// two, so that  key(i, j) = nbs_j - dot  with dot the raw MFMA result and nbs_j = ||b_j||^2 / 2s.
// Epilogue: a warm row's accumulator STARTS at its threshold T_i (the C operand of the first MFMA), so after the
// K loop it holds T_i + dot and  key <= T_i  <=>  acc >= nbs_j : the test is a max over the lane's 16 rows and
// one compare per column block.  Rows without any finite key yet (T = +inf: "cold", normally the first tile
// only) start from 0 and always take the slow path.
// LDS: two column tiles, filled by LDS-DMA (global_load_lds_dwordx4: no staging registers, no ds_write pass).  A
// DMA instruction writes its 64 lanes' 16-byte chunks back to back, so the image is lane-linear: 64 columns x PC
// chunk slots (PC = chunks per row rounded up to 16; 768-byte rows for d <= 352).  The bank spread comes from the
// SOURCE side instead: slot p of column col holds chunk p ^ (col & 15), so the 16 lanes of a fragment read
// (same chunk, 16 consecutive columns, pitch = 0 mod 64 banks) hit 16 different slots = all 64 banks once.
// Work order (XCD-aware).  A workgroup is (row block, column split); the reference columns are cut into splits of at
// most ~2 MB of FP16 rows, so that the split an XCD is working on stays in its own 4 MB L2 while all of that XCD's
// workgroups -- consecutive row blocks -- stream it again and again.  Workgroups are dealt round-robin over the 8 XCDs
// (blockIdx % 8) and XCD x takes a contiguous eighth of the splits, one after the other, every row block of one split before
// the next split starts (with fewer than eight splits several XCDs share one, each with its own range of row blocks): each tile of `bh` crosses the fabric about once per launch instead of once per row block (round 2:
// one split, the 256 resident workgroups drifted apart along the 184 MB panel and 72 % of their tile reads missed L2).
// A row's threshold is shared between the splits through `thr_best` (racy reads, atomic minimum at the end): any
// threshold that some column has reached is a valid start value, so later splits start warm and append next to nothing;
// the final result does not depend on who saw what when (k_half_final decides in float64 among a superset).
__device__ inline void atomic_min_f32(float *addr, float v)
{
    if (v >= 0.0f) atomicMin(reinterpret_cast<int *>(addr), __float_as_int(v));
    else atomicMax(reinterpret_cast<unsigned *>(addr), (unsigned)__float_as_int(v));
}

template <int KS>
__global__ __launch_bounds__(512, 1) void k_match_half(const _Float16 *__restrict__ ah, int64_t m1,
                                                        const _Float16 *__restrict__ bh, int64_t m2_pad,
                                                        const float *__restrict__ nbs, const float *__restrict__ win,
                                                        int64_t tiles_per_split, int64_t m1_pad, int64_t row_blocks,
                                                        int64_t nsplit, float *__restrict__ thr_best,
                                                        int *__restrict__ cnt, int32_t *__restrict__ cand_j,
                                                        float *__restrict__ cand_k, float *__restrict__ thr_out)
{
    // XCD x (= blockIdx % 8) walks its own contiguous eighth of the split-major list of (split, row block) pairs
    const int64_t wv = sf_xcd_block(), split = wv / row_blocks, rb = wv - split * row_blocks;
    if (split >= nsplit) return;
    constexpr int DP = 16 * KS;
    constexpr int CPR = 2 * KS;                  // 16-byte chunks per row
    constexpr int PC = (CPR + 15) / 16 * 16;     // chunk slots per row in LDS
    constexpr int NI = PC / 8;                   // DMA instructions per wave per tile (64 PC slots / 64 lanes / 8 waves)
    constexpr int TILE_BYTES = HN * PC * 16 + 256; // + the tile's 64 column norms (floats), DMA'd like the rest
    __shared__ __attribute__((aligned(16))) unsigned char Bs[2 * TILE_BYTES];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r31 = lane & 31, h = lane >> 5;
    const int64_t row0 = rb * HM + 32 * wave;

    h8 af[KS];
    {
        const _Float16 *ap = ah + (row0 + r31) * DP + 8 * h;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) af[ks] = *reinterpret_cast<const h8 *>(ap + 16 * ks);
        // have the fragments land here, once: otherwise every k-step of the main loop carries a vmcnt wait for
        // "its" fragment, which would also drain the DMA of the next tile in the middle of the MFMAs
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) asm volatile("" : "+v"(af[ks]));
    }
    // thresholds of the 16 rows this lane sees in an accumulator: row (r & 3) + 8 (r >> 2) + 4 h
    f16v T;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int64_t row = row0 + (r & 3) + 8 * (r >> 2) + 4 * h;
        T[r] = row < m1 ? (thr_best ? __builtin_nontemporal_load(thr_best + row) : INFINITY) : -INFINITY;
    }
    bool cold; // wave-uniform: some real row still has T = +inf
    {
        bool c = false;
#pragma unroll
        for (int r = 0; r < 16; ++r) c |= T[r] == INFINITY;
        cold = __ballot(c) != 0;
    }
    f16v zero;
#pragma unroll
    for (int r = 0; r < 16; ++r) zero[r] = 0.0f;

    // DMA source offsets (bytes inside a tile of bh) of this lane's NI slots; slots whose chunk index falls into
    // the row padding (>= CPR) are never read: they fetch chunk 0 of their column
    unsigned soff[NI];
#pragma unroll
    for (int u = 0; u < NI; ++u) {
        const int P = 64 * (wave + 8 * u) + lane, col = P / PC, p = P - col * PC;
        int c = p ^ (col & 15);
        if (c >= CPR) c = 0;
        soff[u] = (unsigned)(col * (CPR * 16) + c * 16);
    }
    // fragment read addresses: column r31 (+32 for the second block), chunk 2 ks + h -> slot (2 ks + h) ^ (r31 & 15);
    // the low nibble of 2 ks + h takes 8 values per lane, the rest is an immediate offset
    unsigned roff[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) roff[k] = (unsigned)(r31 * (PC * 16) + (((2 * k + h) ^ (r31 & 15)) * 16));

    // column split blockIdx.y scans its own range of tiles with its own thresholds and candidate lists (used when
    // there are too few 256-row blocks to fill the chip); k_half_final merges the splits
    const int64_t jt0 = split * tiles_per_split;
    const int64_t ntiles = (m2_pad / HN < jt0 + tiles_per_split) ? m2_pad / HN : jt0 + tiles_per_split; // end tile
    cnt += split * m1_pad;
    cand_j += split * m1_pad * HCAP;
    cand_k += split * m1_pad * HCAP;
    thr_out += split * m1_pad;
    const unsigned char *bbytes = reinterpret_cast<const unsigned char *>(bh);
    // The DMA is issued from inline assembly: the compiler then does not know about it and puts no vmcnt(0) in
    // front of the fragment reads of the OTHER buffer (it cannot tell the two halves of Bs apart); completion is
    // waited for explicitly before the barrier that publishes the tile.  M0 = LDS destination (wave-uniform).
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)Bs;
    const unsigned wave_u = (unsigned)__builtin_amdgcn_readfirstlane(wave);
#define SF_H_DMA16(GPTR, LDS_DST)                                                                                   \
    {                                                                                                               \
        unsigned keep_;                                                                                             \
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\t"       \
                     "s_mov_b32 m0, %0"                                                                             \
                     : "=&s"(keep_)                                                                                 \
                     : "v"(GPTR), "s"(LDS_DST)                                                                      \
                     : "memory");                                                                                   \
    }
#define SF_H_DMA4(GPTR, LDS_DST)                                                                                    \
    {                                                                                                               \
        unsigned keep_;                                                                                             \
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\t"         \
                     "s_mov_b32 m0, %0"                                                                             \
                     : "=&s"(keep_)                                                                                 \
                     : "v"(GPTR), "s"(LDS_DST)                                                                      \
                     : "memory");                                                                                   \
    }
#define SF_H_DMA(JT, BUF)                                                                                           \
    {                                                                                                               \
        const unsigned char *tile_ = bbytes + (JT) * (int64_t)(HN * DP * 2);                                        \
        const unsigned dst_ = lds_base + (unsigned)(BUF) * TILE_BYTES + 1024u * wave_u;                             \
        _Pragma("unroll") for (int u = 0; u < NI; ++u) SF_H_DMA16(tile_ + soff[u], dst_ + 8192u * u)                \
        if (wave_u == 0) SF_H_DMA4(nbs + (JT) * HN + lane, lds_base + (unsigned)(BUF) * TILE_BYTES + HN * PC * 16)  \
    }
    SF_H_DMA(jt0, 0)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int64_t jt = jt0; jt < ntiles; ++jt) {
        const int buf = (int)((jt - jt0) & 1);
        if (jt + 1 < ntiles) SF_H_DMA(jt + 1, buf ^ 1)
        const unsigned char *bp = Bs + buf * TILE_BYTES;
#define SF_H_FRAG(KSTEP, CB) \
    (*reinterpret_cast<const h8 *>(bp + roff[(KSTEP) & 7] + ((KSTEP) >> 3) * 256 + (CB) * (32 * PC * 16)))
        // One basic block: fragments are read PF k-steps ahead of the MFMAs that use them, and the scheduler is
        // told to keep that interleave (2 MFMA : 2 LDS reads) instead of sinking every read next to its use.
        constexpr int PF = 3;
        h8 q0[PF], q1[PF];
#pragma unroll
        for (int i = 0; i < PF; ++i) {
            q0[i] = SF_H_FRAG(i, 0);
            q1[i] = SF_H_FRAG(i, 1);
        }
        f16v acc[2];
        acc[0] = cold ? zero : T;
        acc[1] = acc[0];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[ks], q0[ks % PF], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[ks], q1[ks % PF], acc[1], 0, 0, 0);
            if (ks + PF < KS) {
                q0[ks % PF] = SF_H_FRAG(ks + PF, 0);
                q1[ks % PF] = SF_H_FRAG(ks + PF, 1);
            }
        }
        __builtin_amdgcn_sched_group_barrier(0x100, 2 * PF, 0); // DS reads
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0); // MFMA
            if (ks + PF < KS) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
        }
        const float *nbl = reinterpret_cast<const float *>(bp + HN * PC * 16);
        const float nbv2[2] = {nbl[r31], nbl[32 + r31]};
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
            const int64_t j = jt * HN + 32 * cb + r31;
            const float nbv = nbv2[cb];
            float mx = acc[cb][0];
#pragma unroll
            for (int r = 1; r < 16; ++r) mx = fmaxf(mx, acc[cb][r]);
            if (cold || __ballot(mx >= nbv)) {
                const bool was_cold = cold;
                // rare path: keep its address arithmetic out of the loop-invariant registers of the MFMA loop
                int rowb = (int)(row0 - rb * HM) + 4 * h;
                asm volatile("" : "+v"(rowb));
                const int64_t rowbase = rb * HM + rowb;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float dot = was_cold ? acc[cb][r] : acc[cb][r] - T[r];
                    const float key = nbv - dot;
                    const bool hit = (key <= T[r]) & (key < INFINITY);
                    if (__ballot(hit)) {
                        float v = hit ? key : INFINITY;
                        v = fminf(v, dpp_f32<0xB1>(v));  // quad_perm [1,0,3,2]
                        v = fminf(v, dpp_f32<0x4E>(v));  // quad_perm [2,3,0,1]
                        v = fminf(v, dpp_f32<0x141>(v)); // row_half_mirror
                        v = fminf(v, dpp_f32<0x140>(v)); // row_mirror: min of the 16-lane row in every lane
                        v = fminf(v, __shfl_xor(v, 16)); // the two DPP rows of this 32-lane half
                        const int64_t row = rowbase + (r & 3) + 8 * (r >> 2);
                        const float tn = fminf(T[r], v + win[row]);
                        if (hit && key <= tn) {
                            const int s = atomicAdd(&cnt[row], 1);
                            if (s < HCAP) {
                                cand_j[row * HCAP + s] = (int32_t)j;
                                cand_k[row * HCAP + s] = key;
                            }
                        }
                        // the second column block of this tile was accumulated from the OLD threshold: keep its
                        // dots recoverable by moving the difference into its accumulator
                        if (cb == 0 && !was_cold) acc[1][r] += tn - T[r];
                        T[r] = tn;
                    }
                }
                if (was_cold && cb == 1) {
                    bool c = false;
#pragma unroll
                    for (int r = 0; r < 16; ++r) c |= T[r] == INFINITY;
                    cold = __ballot(c) != 0;
                }
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // this wave's DMA pieces of the next tile have landed
        __syncthreads();
    }
    if (r31 == 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int64_t row = row0 + (r & 3) + 8 * (r >> 2) + 4 * h;
            thr_out[row] = T[r];
            if (thr_best && row < m1 && T[r] < INFINITY) atomic_min_f32(thr_best + row, T[r]);
        }
    }
#undef SF_H_DMA
#undef SF_H_DMA16
#undef SF_H_DMA4
#undef SF_H_FRAG
}

// Step 3: the reference's arithmetic on the surviving candidates (scipy's loop order: every distance is one lane's sequential
// sum).  LPR lanes share a scan row -- lane u takes the candidates u, u + LPR, ... of every split's list -- and fold their
// (distance, column) pairs with the reference's tie rule (smaller column); with one lane per row a launch of 10^4 rows is 40
// workgroups of serial 352-term sums.
template <int LPR>
__global__ void k_half_final(const double *__restrict__ a, int64_t m1, const double *__restrict__ b, int64_t d,
                             const unsigned char *__restrict__ a_ok, const int *__restrict__ cnt,
                             const int32_t *__restrict__ cand_j, const float *__restrict__ cand_k,
                             const float *__restrict__ thr, const float *__restrict__ win,
                             const double *__restrict__ na2, double unit, int nsplit, int64_t m1_pad,
                             int64_t *__restrict__ idx,
                             double *__restrict__ dist, int *__restrict__ flag, int *__restrict__ n_flagged)
{
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t i = gid / LPR;
    const int sub = (int)(gid % LPR);
    if (i >= m1) return; // (whole groups of LPR lanes: LPR divides the wave)
    if (a_ok && !a_ok[i]) { // masked scan row: +inf from everything, first column (shotfpfh.h, sf_match_argmin_multiscale)
        if (sub == 0) {
            idx[i] = 0;
            if (dist) dist[i] = INFINITY;
            flag[i] = 0;
        }
        return;
    }
    // final threshold: the smallest over the column splits (each = that split's smallest key + W_i)
    float t = INFINITY;
    for (int s = 0; s < nsplit; ++s) t = fminf(t, thr[(int64_t)s * m1_pad + i]);
    // a split whose list overflowed lost columns with keys >= its own minimum = thr_s - W_i; they matter only if
    // that minimum is within the final threshold
    bool overflow = false;
    const float w = win[i] * 1.001f;
    for (int s = 0; s < nsplit; ++s)
        overflow |= cnt[(int64_t)s * m1_pad + i] > HCAP && !(thr[(int64_t)s * m1_pad + i] - w > t);
    double best = INFINITY;
    int64_t bj = -1;
    bool model_ok = true;
    if (!overflow) {
        const double *ai = a + i * d;
        const double half_w = 0.5 * (double)win[i], na = na2[i];
        for (int s = 0; s < nsplit; ++s) {
            const int64_t base = ((int64_t)s * m1_pad + i) * HCAP;
            const int n = min(cnt[(int64_t)s * m1_pad + i], HCAP);
            for (int c = sub; c < n; c += LPR) {
                if (!(cand_k[base + c] <= t)) continue;
                const int64_t j = cand_j[base + c];
                const double *bjp = b + j * d;
                double acc = 0.0;
                for (int64_t u = 0; u < d; ++u) {
                    const double df = ai[u] - bjp[u];
                    acc += df * df; // left to right, no FMA: scipy's euclidean loop
                }
                // safety net for the error model: the pre-filter's key of this pair must be within eps_i (< W_i / 2)
                // of the float64 one, ||a - b||^2 - ||a||^2; a row where it is not is handed to the FP64 path
                model_ok &= fabs((acc - na) * unit - (double)cand_k[base + c]) <= half_w;
                const double dj = sqrt(acc);
                if (dj < best || (dj == best && j < bj) || bj < 0) {
                    if (!(dj == dj)) continue; // NaN: leave the row to the float64 path
                    best = dj;
                    bj = j;
                }
            }
        }
    }
#pragma unroll
    for (int off = LPR / 2; off > 0; off >>= 1) { // (minimum with the smaller column on ties: the order of the fold does not matter)
        const double ob = __shfl_xor(best, off);
        const int64_t oj = __shfl_xor(bj, off);
        const int om = __shfl_xor((int)model_ok, off);
        if (oj >= 0 && (bj < 0 || ob < best || (ob == best && oj < bj))) {
            best = ob;
            bj = oj;
        }
        model_ok = model_ok && om;
    }
    if (sub != 0) return;
    const bool decided = bj >= 0 && model_ok;
    idx[i] = decided ? bj : 0;
    if (dist) dist[i] = best;
    flag[i] = decided ? 0 : 1;
    if (!decided) atomicAdd(n_flagged, 1);
}

__global__ void k_half_gather_rows(const double *__restrict__ a, int64_t d, const int64_t *__restrict__ rows, int64_t nr,
                                   double *__restrict__ out)
{
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= nr * d) return;
    const int64_t r = g / d, t = g - r * d;
    out[g] = a[rows[r] * d + t];
}

__global__ void k_half_scatter(const int64_t *__restrict__ rows, int64_t nr, const int64_t *__restrict__ sidx,
                               const double *__restrict__ sdist, int64_t *__restrict__ idx, double *__restrict__ dist)
{
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= nr) return;
    idx[rows[g]] = sidx[g];
    if (dist) dist[rows[g]] = sdist[g];
}

int host_max(sf_ctx *ctx, const double *v, int64_t n, double *part, double *out)
{
    SF_LAUNCH(ctx, "k8_half_max", k_half_max, dim3(256), dim3(256), v, n, part);
    std::vector<double> h(256);
    SF_HIP(hipMemcpyAsync(h.data(), part, 256 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    SF_HIP(hipStreamSynchronize(ctx->stream));
    double mx = 0.0;
    for (double x : h) mx = std::max(mx, x);
    *out = mx;
    return SF_OK;
}

// power of two p with p * sqrt(n2max) in [2^13, 2^14]: FP16 keeps 11 significant bits down to 2^-14, so entries
// 2^-27 below the largest row norm still round relatively; whatever is lost is measured, not assumed
bool pick_scale(double n2max, double *scale)
{
    if (!(n2max > 0.0) || !std::isfinite(n2max)) return false;
    int e = 0;
    std::frexp(std::sqrt(n2max), &e); // sqrt = f * 2^e, f in [0.5, 1)
    const int k = 14 - e;
    if (k < -100 || k > 100) return false;
    *scale = std::ldexp(1.0, k);
    return true;
}

} // namespace

// SF_MATCH_HALF=0 disables the pre-filter, =1 forces it for every problem the FP64 GEMM path would take.
int sf_match_half_mode()
{
    const char *e = getenv("SF_MATCH_HALF");
    if (!e || !e[0]) return -1;
    return e[0] == '0' ? 0 : 1;
}

// rc SF_OK and *used = 1 when the pre-filter produced the result; *used = 0 (nothing written) when the input is
// not suitable (d > 352, zero / non-finite norms) and the caller must take the FP64 path.
int sf_match_half(sf_ctx *ctx, const double *da, int64_t m1, const double *db, int64_t m2, int64_t d, int64_t *didx,
                  double *ddist, const char *name, int64_t *n_slow, const unsigned char *a_ok, const unsigned char *b_ok,
                  int *used)
{
    *used = 0;
    if (d > 352 || m1 <= 0 || m2 <= 0 || m2 > 0x7fffffff) return SF_OK;
    const int ks = d <= 128 ? 8 : 22, dp = 16 * ks;
    const int64_t m1p = sf_div_up(m1, HM) * HM, m2p = sf_div_up(m2, HN) * HN;
    _Float16 *ah = nullptr, *bh = nullptr;
    double *ea = nullptr, *qa = nullptr, *na2 = nullptr, *eb = nullptr, *qb = nullptr, *nb2 = nullptr, *part = nullptr;
    float *nbf = nullptr, *win = nullptr, *candk = nullptr, *thr = nullptr;
    int *cnt = nullptr, *flag = nullptr, *nflag = nullptr;
    int32_t *candj = nullptr;
    sf_pool_guard tmp(ctx); // every block returns to the pool on any exit, error returns of SF_LAUNCH / SF_HIP included
#define SF_HALLOC(ptr, count) SF_CHECK(tmp.alloc(&ptr, (size_t)(count)))
    SF_HALLOC(part, 256);
    SF_HALLOC(nb2, m2p); SF_HALLOC(eb, m2p); SF_HALLOC(qb, m2p); SF_HALLOC(nbf, m2p);
    SF_HALLOC(na2, m1p); SF_HALLOC(ea, m1p); SF_HALLOC(qa, m1p);
    SF_HALLOC(ah, m1p * dp); SF_HALLOC(bh, m2p * dp);
    // the scales come from the largest row norms: a first pass of the converter that only produces ||row||^2
    double namax = 0.0, nbmax = 0.0, sa = 1.0, sb = 1.0;
    {
        SF_LAUNCH(ctx, "k8_half_convert", k_half_convert, dim3((unsigned)sf_div_up(m1p, 4)), dim3(256), da, m1, m1p, d, dp,
                  1.0, (const unsigned char *)nullptr, (_Float16 *)nullptr, ea, qa, na2, (float *)nullptr, 1.0);
        SF_LAUNCH(ctx, "k8_half_convert", k_half_convert, dim3((unsigned)sf_div_up(m2p, 4)), dim3(256), db, m2, m2p, d, dp,
                  1.0, b_ok, (_Float16 *)nullptr, eb, qb, nb2, (float *)nullptr, 1.0);
        int rc = host_max(ctx, na2, m1, part, &namax);
        if (rc == SF_OK) rc = host_max(ctx, nb2, m2, part, &nbmax);
        if (rc != SF_OK) return rc;
    }
    if (!pick_scale(namax, &sa) || !pick_scale(nbmax, &sb)) return SF_OK;
    // accumulator units: 1 / 2s = sa sb / 2.  ||b||^2 / 2s ~ 2^27 ||b||max / ||a||max must stay a normal float
    const double unit = 0.5 * sa * sb;
    if (!(nbmax * unit < 1e30) || !(nbmax * unit > 1e-20)) return SF_OK;
    SF_LAUNCH(ctx, "k8_half_convert", k_half_convert, dim3((unsigned)sf_div_up(m1p, 4)), dim3(256), da, m1, m1p, d, dp, sa,
              (const unsigned char *)nullptr, ah, ea, qa, na2, (float *)nullptr, 1.0);
    SF_LAUNCH(ctx, "k8_half_convert", k_half_convert, dim3((unsigned)sf_div_up(m2p, 4)), dim3(256), db, m2, m2p, d, dp, sb,
              b_ok, bh, eb, qb, nb2, nbf, unit);
    double ebmax = 0.0, qbmax = 0.0;
    {
        int rc = host_max(ctx, eb, m2, part, &ebmax);
        if (rc == SF_OK) rc = host_max(ctx, qb, m2, part, &qbmax);
        if (rc != SF_OK) return rc;
    }
    if (!std::isfinite(ebmax) || !std::isfinite(qbmax)) return SF_OK;
    // column splits: (1) each split is short enough for an XCD's workgroups to share its tiles through their L2 (8 MB: hit rate
    // 0.83 against 0.90 / 0.89 at 2 / 4 MB with fewer lists to walk, profiles/r03_match_summary.md) -- see the note above k_match_half; (2) with few row blocks,
    // enough workgroups for two per CU's worth of the chip, each with at least 32 tiles to scan
    const int64_t col_tiles = m2p / HN;
    const int64_t chunk_kb = 8192;
    const int64_t tiles_in_l2 = std::max<int64_t>(8, chunk_kb * 1024 / ((int64_t)HN * dp * 2));
    int64_t nsplit = sf_div_up(col_tiles, tiles_in_l2);
    if ((m1p / HM) * nsplit < 512) nsplit = std::max<int64_t>(nsplit, std::min<int64_t>(sf_div_up(512, m1p / HM), std::max<int64_t>(col_tiles / 32, 1)));
    if (const char *e = getenv("SF_MATCH_HALF_SPLITS")) nsplit = std::max<int64_t>(1, std::min<int64_t>(atoll(e), col_tiles));
    const int64_t tiles_per_split = sf_div_up(col_tiles, nsplit);
    nsplit = sf_div_up(col_tiles, tiles_per_split);
    const bool seed = true; // (a row's threshold shared between its splits: without it the pass runs 2.3 x longer, same result)
    // scan rows go through in slabs: every (row, split) pair owns HCAP candidate slots, and 1M rows x 344 splits of them
    // would be 90 GB -- a slab keeps the lists within ~8 GB (at least 64 row blocks, so a slab still fills the chip)
    const int64_t slab_rows = std::min<int64_t>(m1p, HM * std::max<int64_t>(64, ((int64_t)8 << 30) / (nsplit * HCAP * 8 * HM)));
    float *tbest = nullptr;
    SF_HALLOC(win, slab_rows); SF_HALLOC(thr, nsplit * slab_rows); SF_HALLOC(cnt, nsplit * slab_rows); SF_HALLOC(tbest, slab_rows);
    SF_HALLOC(candj, nsplit * slab_rows * HCAP); SF_HALLOC(candk, nsplit * slab_rows * HCAP);
    SF_HALLOC(flag, m1); SF_HALLOC(nflag, 1);
    const double gamma = (double)(dp + 32) * 2.384185791015625e-07; // 2^-22
    SF_HIP(hipMemsetAsync(nflag, 0, sizeof(int), ctx->stream));
    for (int64_t r0 = 0; r0 < m1; r0 += slab_rows) {
        const int64_t ms = std::min(m1 - r0, slab_rows), msp = sf_div_up(ms, HM) * HM, row_blocks = msp / HM;
        SF_LAUNCH(ctx, "k8_half_window", k_half_window, dim3((unsigned)sf_div_up(msp, 256)), dim3(256), (const double *)(ea + r0),
                  (const double *)(qa + r0), (const double *)(na2 + r0), ms, msp, std::sqrt(nbmax), ebmax, qbmax, nbmax, gamma, unit, win);
        SF_HIP(hipMemsetAsync(cnt, 0, (size_t)(nsplit * msp) * sizeof(int), ctx->stream));
        SF_HIP(hipMemsetD32Async((hipDeviceptr_t)tbest, 0x7f800000, (size_t)msp, ctx->stream)); // +inf
        const int64_t wgs = sf_xcd_grid(nsplit * row_blocks); // XCD x: the x-th eighth of the split-major (split, row block) list
        if (wgs > 0x7fffffffLL) { sf_set_error("sf_match_half: %lld workgroups exceed a launch", (long long)wgs); return SF_ERR_UNSUPPORTED; }
        if (ks == 8) {
            SF_LAUNCH(ctx, name, k_match_half<8>, dim3((unsigned)wgs), dim3(512), (const _Float16 *)(ah + r0 * dp),
                      ms, (const _Float16 *)bh, m2p, (const float *)nbf, (const float *)win, tiles_per_split, msp, row_blocks, nsplit,
                      seed ? tbest : (float *)nullptr, cnt, candj, candk, thr);
        } else {
            SF_LAUNCH(ctx, name, k_match_half<22>, dim3((unsigned)wgs), dim3(512), (const _Float16 *)(ah + r0 * dp),
                      ms, (const _Float16 *)bh, m2p, (const float *)nbf, (const float *)win, tiles_per_split, msp, row_blocks, nsplit,
                      seed ? tbest : (float *)nullptr, cnt, candj, candk, thr);
        }
#define SF_HALF_FINAL(LPR)                                                                                            \
        SF_LAUNCH(ctx, "k8_half_final", k_half_final<LPR>, dim3((unsigned)sf_div_up(ms * LPR, 256)), dim3(256), da + r0 * d, ms, db, d, \
                  a_ok ? a_ok + r0 : a_ok, (const int *)cnt, (const int32_t *)candj, (const float *)candk, (const float *)thr, \
                  (const float *)win, (const double *)(na2 + r0), unit, (int)nsplit, msp, didx + r0, ddist ? ddist + r0 : ddist, \
                  flag + r0, nflag);
        // (10^4 rows: 1.53 / 0.48 / 0.26 ms with 1 / 4 / 16 lanes per row; 262 144 rows: 3.9 / 4.5 / 5.5 ms -- the chip is full with one)
        // (75 000 rows -- what the integer pass hands on of config 4's 10^6 -- with one lane per row: 1 172 waves, one per SIMD, every
        // one of them a serial chain of 2 x 352 scattered loads: 4.7 ms; sixteen lanes per row there: round 6)
        const int lpr = ms <= 160000 ? 16 : 1;
        if (lpr == 1) { SF_HALF_FINAL(1) } else { SF_HALF_FINAL(16) }
#undef SF_HALF_FINAL
    }
    int nf = 0;
    SF_HIP(hipMemcpyAsync(&nf, nflag, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    SF_HIP(hipStreamSynchronize(ctx->stream));
    int rc = SF_OK;
    if (n_slow) *n_slow = 0;
    if (nf > 0) { // overflowing / non-finite rows: the FP64 path on the gathered rows
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
        SF_HALLOC(drows, nr); SF_HALLOC(sidx, nr); SF_HALLOC(sdist, nr); SF_HALLOC(sub, nr * d);
        SF_HIP(hipMemcpyAsync(drows, rows.data(), (size_t)nr * sizeof(int64_t), hipMemcpyHostToDevice, ctx->stream));
        SF_LAUNCH(ctx, "k8_gather_rows", k_half_gather_rows, dim3((unsigned)sf_div_up(nr * d, 256)), dim3(256), da, d,
                  (const int64_t *)drows, nr, sub);
        int64_t slow2 = 0;
        rc = sf_match_gemm_f64(ctx, sub, nr, db, m2, d, sidx, sdist, "k8_match_gemm_overflow", &slow2, nullptr, b_ok);
        if (rc == SF_OK) {
            SF_LAUNCH(ctx, "k8_scatter_results", k_half_scatter, dim3((unsigned)sf_div_up(nr, 256)), dim3(256),
                      (const int64_t *)drows, nr, (const int64_t *)sidx, (const double *)sdist, didx, ddist);
        }
        SF_HIP(hipStreamSynchronize(ctx->stream)); // rows.data() is a host buffer
        if (n_slow) *n_slow = nr;
    }
#undef SF_HALLOC
    *used = 1;
    return rc;
}
