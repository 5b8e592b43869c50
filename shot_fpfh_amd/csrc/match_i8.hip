// match_i8.hip -- K8 pre-filter on the INT8 matrix cores: the arg-min of cdist(a, b) (matching.py:47-52, 164-168) with an
// integer pass that only PRUNES, in front of the FP16 pre-filter (match_half.hip) and the float64 decision.
//
// Why.  On random operands this chip sustains 1.7 PFLOP/s with v_mfma_f32_32x32x16_f16 and 3.8 Pop/s with
// v_mfma_i32_32x32x32_i8 (tools/ubench/mfma_rates.hip: the clock it holds under matrix load sets both; 2.2 x per pair of
// descriptors).  The FP16 pass of match_half.hip runs at 0.77 of the former: the operand type is what is left.
//
//   1. both descriptor sets are scaled by 127 / (largest |entry|) and rounded to int8 (k_i8_convert); per row the EXACT
//      quantisation error ||a_i - a'_i||, the quantised norm ||a'_i|| and ||a_i||^2 are kept in float64.
//   2. k_i8_min: one pass of v_mfma_i32_32x32x32_i8 over all pairs gives INTEGER ranking keys
//          K(i, j) = NB_j - D(i, j),   D = sum_k qa_ik qb_jk (exact),  NB_j = round(||b_j||^2 sa sb / 2)
//      -- the key ||b_j||^2 - 2 a'_i . b'_j in units of u = 2 / (sa sb), with one rounding (of NB) -- and keeps of them only
//      the MINIMUM per (row, column split): the accumulator starts at -NB_j, so the epilogue of a 32 x 32 block is sixteen
//      v_max_i32, no branch, no list.  (A first version kept candidate lists in this pass, like the FP16 pass does: with a window
//      as wide as eight bits leave it, a row meets a key inside "smallest so far + W" hundreds of times before it meets its nearest
//      descriptor, most 32 x 32 blocks took the list path and the pass ran at half the FP16 pass's speed.)
//   3. k_i8_live: a row's global minimum Kmin_i = min over the splits; a split is LIVE for the row when its own minimum is within
//      W_i of Kmin_i -- only there can the reference's arg-min be.  A row with a clear nearest descriptor has ONE live split.
//   4. k_i8_collect: the (row, live split) pairs, sorted by split, 256 to a workgroup, scan their split again -- 1/nsplit of the
//      first pass's work per pair -- with the FINAL threshold Kmin_i + W_i known from the start: what passes is the candidate set.
//   5. k_i8_final: the candidates get the reference's float64 distance -- sequential sum, square root -- and the smallest (lowest
//      column on ties) wins, scipy's first-minimum rule.
//
// Exactness.  With key(i, j) = ||b_j||^2 - 2 a_i . b_j exact:  |u K(i, j) - key(i, j)| <= eps_i + u / 2,
//     eps_i = 2 (ea_i Bmax + qa_i EBmax)     (Cauchy-Schwarz on (a - a').b + a'.(b - b'); integer products and sums are exact,
//                                             so none of the FP16 pass's accumulation terms)
// The reference's arg-min j* has key(i, j*) <= key(i, j1) for the column j1 that attains Kmin_i, hence
// K(i, j*) <= Kmin_i + 2 eps_i / u + 1: with W_i = ceil(2 eps_i / u) + 2 its split is live and it passes the collect pass's
// test.  The candidate set therefore always contains the reference's arg-min and everything that ties with it.
//
// The window is WIDE: eight bits leave ea ~ 0.01 on a unit SHOT row, W ~ 0.09 in squared distance, against 0.002 for FP16.
// A row whose nearest descriptor stands clear of the rest (a true correspondence: 92 % of config 4's rows) has one live split
// and one candidate; a row without one has a column within W of its minimum in every split.  Rows with more than I_LIVE live
// splits (or an overflowing candidate list) are FLAGGED and re-done by the FP16 pass (sf_match_half on the gathered rows; its
// own leftovers go to float64): the result equals the exact kernel's for every input, only the work depends on the data.  A
// pilot slab decides whether the integer pass pays at all: when more than SF_I8_MAX_FLAGGED (0.35) of its rows are flagged the
// whole problem takes the FP16 pass.
// Roofline: int8 matrix cores (dense peak ~5 Pop/s), 2 m1 m2 dpad op.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <vector>

#include "common.h"
#include "device_util.h"

int sf_match_half(sf_ctx *ctx, const double *da, int64_t m1, const double *db, int64_t m2, int64_t d, int64_t *didx,
                  double *ddist, const char *name, int64_t *n_slow, const unsigned char *a_ok, const unsigned char *b_ok,
                  int *used); // match_half.hip
int sf_match_gemm_f64(sf_ctx *ctx, const double *da, int64_t m1, const double *db, int64_t m2, int64_t d, int64_t *didx,
                      double *ddist, const char *name, int64_t *n_slow, const unsigned char *a_ok,
                      const unsigned char *b_ok); // match_gemm.hip

namespace {

typedef int i4v __attribute__((ext_vector_type(4)));
typedef int i16v __attribute__((ext_vector_type(16)));

constexpr int IN = 64;    // columns per LDS tile
constexpr int ICAP = 32;  // candidate slots per (row, live column split) pair
constexpr int I_LIVE = 4; // live splits a row may have and still be served here
constexpr int I_BIG = 1 << 30;     // NB of a masked / padding column; threshold of a row that has seen no key yet ("cold")
constexpr int I_REAL = 1 << 29;    // every real NB and every warm threshold is below this

// ||row||^2 and the largest |entry| of every row (non-finite entries -> +inf in both: the caller then leaves the problem
// to the float64 path)
__global__ __launch_bounds__(256) void k_i8_rowstat(const double *__restrict__ a, int64_t m, int64_t d, double *__restrict__ n2,
                                                    double *__restrict__ amax)
{
    const int lane = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= m) return;
    double sn = 0.0, mx = 0.0;
    bool bad = false;
    for (int64_t t = lane; t < d; t += 64) {
        const double v = a[i * d + t];
        bad |= !(fabs(v) <= 1.7976931348623157e308);
        sn += v * v;
        mx = fmax(mx, fabs(v));
    }
    sn = sf_wave_sum(sn);
    for (int off = 32; off > 0; off >>= 1) mx = fmax(mx, __shfl_xor(mx, off));
    bad = __ballot(bad) != 0;
    if (lane == 0) {
        n2[i] = bad ? INFINITY : sn;
        amax[i] = bad ? INFINITY : mx;
    }
}

// One wave per row: int8 image of round(scale * row) (zero padded to dp bytes, dp a multiple of 32), the row's exact
// quantisation error and quantised norm, and -- for the reference side -- its integer norm term NB.
__global__ __launch_bounds__(256) void k_i8_convert(const double *__restrict__ a, int64_t m, int64_t m_pad, int64_t d, int dp,
                                                    double scale, const unsigned char *__restrict__ ok,
                                                    unsigned *__restrict__ out, double *__restrict__ err,
                                                    double *__restrict__ qn, const double *__restrict__ n2,
                                                    int *__restrict__ nbi, double unit)
{
    const int lane = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= m_pad) return;
    const bool real = i < m;
    double se = 0.0, sq = 0.0;
    for (int w = lane; w < dp / 4; w += 64) { // one dword = four entries per lane and step
        unsigned packed = 0;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int t = 4 * w + u;
            const double v = (real && t < d) ? a[i * d + t] : 0.0;
            double q = rint(v * scale);
            q = fmin(fmax(q, -127.0), 127.0);
            const double back = q / scale, e = v - back;
            se += e * e;
            sq += back * back;
            packed |= ((unsigned)(int)q & 0xffu) << (8 * u);
        }
        out[i * (dp / 4) + w] = packed;
    }
    se = sf_wave_sum(se);
    sq = sf_wave_sum(sq);
    if (lane == 0) {
        err[i] = real ? sqrt(se) : 0.0;
        qn[i] = real ? sqrt(sq) : 0.0;
        if (nbi) {
            const bool masked = !real || (ok && !ok[i]);
            nbi[i] = masked ? I_BIG : (int)llrint(n2[i] * unit); // (< 2^23 for d <= 352)
        }
    }
}

// max over i of v[i] (v >= 0; non-finite entries propagate) -> partial[blockIdx]
__global__ void k_i8_max(const double *__restrict__ v, int64_t n, double *__restrict__ partial)
{
    double mx = 0.0;
    bool bad = false;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const double x = v[i];
        bad |= !(x <= 1.7976931348623157e308) || !(x >= 0.0);
        mx = fmax(mx, x);
    }
    if (bad) mx = INFINITY;
    for (int off = 32; off > 0; off >>= 1) mx = fmax(mx, __shfl_xor(mx, off));
    __shared__ double s[4];
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = fmax(fmax(s[0], s[1]), fmax(s[2], s[3]));
}

// W_i of the header in integer key units (unit = 1 / u = sa sb / 2); rows beyond m get 0 (their thresholds never move)
__global__ void k_i8_window(const double *__restrict__ ea, const double *__restrict__ qa, const double *__restrict__ na2, int64_t m,
                            int64_t m_pad, double bmax, double ebmax, double nbmax, double unit, int *__restrict__ win)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m_pad) return;
    if (i >= m) { win[i] = 0; return; }
    // (1e-6: the float64 rounding of ea / qa / the maxima, generously; 1e-12 (...): that of the exact keys themselves)
    const double eps = 2.0 * (ea[i] * bmax + qa[i] * ebmax) * (1.0 + 1e-6) + 1e-12 * (na2[i] + nbmax);
    const double w = ceil(2.0 * eps * unit * (1.0 + 1e-9)) + 2.0;
    win[i] = w < (double)(1 << 27) ? (int)w : (1 << 27); // (an absurd window only floods the row's lists: it is then flagged)
}

template <int CTRL>
__device__ __forceinline__ int dpp_i32(int v)
{
    return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, false);
}

// LDS image and fragment addressing shared by the two passes: two column tiles, filled by LDS-DMA (global_load_lds_dwordx4),
// lane-linear image of 64 columns x PC chunk slots of 16 bytes (PC = chunks per row rounded up to 16); slot p of column col holds
// chunk p ^ (col & 15), so the 16 lanes of a fragment read (same chunk, 16 consecutive columns) hit all 64 banks once.  The
// tile's 64 norm terms NB follow it.  The DMA is issued from inline assembly (see k_match_half): the compiler then puts no
// vmcnt(0) in front of the fragment reads of the OTHER buffer; completion is waited for explicitly before the barrier that
// publishes the tile.
#define SF_I_DMA16(GPTR, LDS_DST)                                                                                   \
    {                                                                                                               \
        unsigned keep_;                                                                                             \
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\t"       \
                     "s_mov_b32 m0, %0"                                                                             \
                     : "=&s"(keep_)                                                                                 \
                     : "v"(GPTR), "s"(LDS_DST)                                                                      \
                     : "memory");                                                                                   \
    }
#define SF_I_DMA4(GPTR, LDS_DST)                                                                                    \
    {                                                                                                               \
        unsigned keep_;                                                                                             \
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\t"         \
                     "s_mov_b32 m0, %0"                                                                             \
                     : "=&s"(keep_)                                                                                 \
                     : "v"(GPTR), "s"(LDS_DST)                                                                      \
                     : "memory");                                                                                   \
    }
#define SF_I_DMA(JT, BUF)                                                                                           \
    {                                                                                                               \
        const unsigned char *tile_ = bi + (JT) * (int64_t)(IN * DP);                                                \
        const unsigned dst_ = lds_base + (unsigned)(BUF) * TILE_BYTES + 1024u * wave_u;                             \
        _Pragma("unroll") for (int u = 0; u < NI; ++u) SF_I_DMA16(tile_ + soff[u], dst_ + 8192u * u)                \
        if (wave_u == 0) SF_I_DMA4(nbs + (JT) * IN + lane, lds_base + (unsigned)(BUF) * TILE_BYTES + IN * PC * 16)  \
    }
#define SF_I_FRAG(KSTEP, CB) \
    (*reinterpret_cast<const i4v *>(bp + roff[(KSTEP) & 7] + ((KSTEP) >> 3) * 256 + (CB) * (32 * PC * 16)))
#define SF_I_GEOMETRY                                                                                               \
    constexpr int DP = 32 * KS;                  /* bytes per row */                                                \
    constexpr int CPR = 2 * KS;                  /* 16-byte chunks per row */                                       \
    constexpr int PC = (CPR + 15) / 16 * 16;     /* chunk slots per row in LDS */                                   \
    constexpr int NI = PC / 8;                   /* DMA instructions per wave per tile */                           \
    constexpr int TILE_BYTES = IN * PC * 16 + 256;                                                                  \
    __shared__ __attribute__((aligned(16))) unsigned char Bs[2 * TILE_BYTES];                                       \
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;                                                  \
    const int r31 = lane & 31, h = lane >> 5;                                                                       \
    unsigned soff[NI];                                                                                              \
    _Pragma("unroll") for (int u = 0; u < NI; ++u) {                                                                \
        const int P = 64 * (wave + 8 * u) + lane, col = P / PC, p = P - col * PC;                                   \
        int c = p ^ (col & 15);                                                                                     \
        if (c >= CPR) c = 0;                                                                                        \
        soff[u] = (unsigned)(col * (CPR * 16) + c * 16);                                                            \
    }                                                                                                               \
    unsigned roff[8];                                                                                               \
    _Pragma("unroll") for (int k = 0; k < 8; ++k) roff[k] = (unsigned)(r31 * (PC * 16) + (((2 * k + h) ^ (r31 & 15)) * 16)); \
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)Bs;              \
    const unsigned wave_u = (unsigned)__builtin_amdgcn_readfirstlane(wave);

// Pass 1.  ai: m1_pad x DP bytes (m1_pad a multiple of 256 RB), bi: m2_pad x DP bytes (m2_pad a multiple of 64), row-major int8,
// DP = 32 KS.  RB: 32-row blocks per wave (2: one B fragment from LDS feeds two MFMAs, half the LDS reads and half the L2 -> LDS
// traffic per MFMA).  smin[split * m1_pad + row] = min over the split's columns of K(row, .).
// Work order: as k_match_half -- XCD x (= blockIdx % 8) walks its own contiguous eighth of the split-major list of (split, row
// block) pairs, so the column split an XCD is working on stays in its L2 while its workgroups stream it.
template <int KS, int RB>
__global__ __launch_bounds__(512, 1) void k_i8_min(const unsigned char *__restrict__ ai, const unsigned char *__restrict__ bi,
                                                    const int *__restrict__ nbs, const int2 *__restrict__ split_tiles,
                                                    int64_t split_first, int64_t m1_pad, int64_t row_blocks, int64_t nsplit,
                                                    int *__restrict__ smin)
{
    // (a launch serves the splits split_first .. split_first + nsplit - 1 of the table: all of them, or a chunk's)
    const int64_t wv = sf_xcd_block(), sl = wv / row_blocks, rb = wv - sl * row_blocks, split = split_first + sl;
    if (sl >= nsplit) return;
    SF_I_GEOMETRY
    constexpr int HMB = 256 * RB; // rows per workgroup
    const int64_t row0 = rb * HMB + 32 * RB * wave;
    i4v af[RB][KS];
#pragma unroll
    for (int b = 0; b < RB; ++b) {
        const unsigned char *ap = ai + (row0 + 32 * b + r31) * DP + 16 * h;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) af[b][ks] = *reinterpret_cast<const i4v *>(ap + 32 * ks);
    }
#pragma unroll
    for (int b = 0; b < RB; ++b)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) asm volatile("" : "+v"(af[b][ks])); // (land here, once: see k_match_half)
    i16v best[RB]; // max over this lane's columns of D - NB = -K, rows 32 b + (r & 3) + 8 (r >> 2) + 4 h
#pragma unroll
    for (int b = 0; b < RB; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) best[b][r] = (int)0x80000000;
    const int2 st = split_tiles[split]; // tiles [x, y) of 64 columns
    const int64_t jt0 = st.x, ntiles = st.y;
    SF_I_DMA(jt0, 0)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int64_t jt = jt0; jt < ntiles; ++jt) {
        const int buf = (int)((jt - jt0) & 1);
        if (jt + 1 < ntiles) SF_I_DMA(jt + 1, buf ^ 1)
        const unsigned char *bp = Bs + buf * TILE_BYTES;
        constexpr int PF = 3;
        const int *nbl = reinterpret_cast<const int *>(bp + IN * PC * 16);
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
            i4v q[PF];
#pragma unroll
            for (int i = 0; i < PF; ++i) q[i] = SF_I_FRAG(i, cb);
            const int neg_nb = -nbl[32 * cb + r31];
            i16v acc[RB];
#pragma unroll
            for (int b = 0; b < RB; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[b][r] = neg_nb;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
                for (int b = 0; b < RB; ++b) acc[b] = __builtin_amdgcn_mfma_i32_32x32x32_i8(af[b][ks], q[ks % PF], acc[b], 0, 0, 0);
                if (ks + PF < KS) q[ks % PF] = SF_I_FRAG(ks + PF, cb);
            }
#pragma unroll
            for (int b = 0; b < RB; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) best[b][r] = max(best[b][r], acc[b][r]);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // this wave's DMA pieces of the next tile have landed
        __syncthreads();
    }
#pragma unroll
    for (int b = 0; b < RB; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            int v = best[b][r];
            v = max(v, dpp_i32<0xB1>(v));  // quad_perm [1,0,3,2]
            v = max(v, dpp_i32<0x4E>(v));  // quad_perm [2,3,0,1]
            v = max(v, dpp_i32<0x141>(v)); // row_half_mirror
            v = max(v, dpp_i32<0x140>(v)); // row_mirror: max of the 16-lane row in every lane
            v = max(v, __shfl_xor(v, 16)); // the two DPP rows of this 32-lane half
            if (r31 == 0) smin[split * m1_pad + row0 + 32 * b + (r & 3) + 8 * (r >> 2) + 4 * h] = -v;
        }
}

// Step 3: which splits can hold a row's arg-min.  One thread per row.  The per-split pair counts (and, in k_i8_place, the pair
// slots) are taken per WORKGROUP from a histogram in LDS, one global atomic per split and workgroup: 10^6 rows adding to a few
// dozen global counters one by one serialise (3.4 ms per 200 000 rows, measured).
constexpr int I_HIST = 4096; // splits the LDS histogram holds; beyond: global atomics (many counters, little contention)

__global__ __launch_bounds__(256) void k_i8_live(const int *__restrict__ smin, const int *__restrict__ win,
                                                 const unsigned char *__restrict__ a_ok, int64_t m1, int64_t m1_pad, int nsplit,
                                                 int *__restrict__ kmin, int *__restrict__ live /* m1 x I_LIVE */,
                                                 int *__restrict__ split_count, int64_t *__restrict__ idx,
                                                 double *__restrict__ dist, int *__restrict__ flag, int *__restrict__ n_flagged)
{
    __shared__ int hist[I_HIST];
    __shared__ int nfl;
    const bool use_lds = nsplit <= I_HIST;
    if (use_lds)
        for (int s = threadIdx.x; s < nsplit; s += blockDim.x) hist[s] = 0;
    if (threadIdx.x == 0) nfl = 0;
    __syncthreads();
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < m1) {
        int sl[I_LIVE];
#pragma unroll
        for (int l = 0; l < I_LIVE; ++l) sl[l] = -1;
        if (a_ok && !a_ok[i]) { // masked scan row: +inf from everything, first column (shotfpfh.h, sf_match_argmin_multiscale)
            idx[i] = 0;
            if (dist) dist[i] = INFINITY;
            flag[i] = 0;
            kmin[i] = I_BIG;
#pragma unroll
            for (int l = 0; l < I_LIVE; ++l) live[i * I_LIVE + l] = -1;
        } else {
            int t = I_BIG;
            for (int s = 0; s < nsplit; ++s) t = min(t, smin[(int64_t)s * m1_pad + i]);
            int n = 0;
            if (t < I_REAL) {
                const int cut = t + win[i];
                for (int s = 0; s < nsplit; ++s)
                    if (smin[(int64_t)s * m1_pad + i] <= cut) {
#pragma unroll
                        for (int l = 0; l < I_LIVE; ++l)
                            if (n == l) sl[l] = s;
                        ++n;
                    }
            }
            const bool served = n >= 1 && n <= I_LIVE;
            kmin[i] = t;
#pragma unroll
            for (int l = 0; l < I_LIVE; ++l) {
                live[i * I_LIVE + l] = served ? sl[l] : -1;
                if (served && sl[l] >= 0) atomicAdd(use_lds ? &hist[sl[l]] : &split_count[sl[l]], 1);
            }
            flag[i] = served ? 0 : 1;
            if (!served) atomicAdd(&nfl, 1);
        }
    }
    __syncthreads();
    if (use_lds)
        for (int s = threadIdx.x; s < nsplit; s += blockDim.x)
            if (hist[s]) atomicAdd(&split_count[s], hist[s]);
    if (threadIdx.x == 0 && nfl) atomicAdd(n_flagged, nfl);
}

// ... and the pairs, split by split: pair slot = split_base[s] + (arrival order within the split; any order gives the same result)
__global__ __launch_bounds__(256) void k_i8_place(const int *__restrict__ kmin, const int *__restrict__ win, int *__restrict__ live,
                                                  int64_t m1, int nsplit, const int *__restrict__ split_base,
                                                  int *__restrict__ split_cursor, int *__restrict__ pair_row,
                                                  int *__restrict__ pair_thr)
{
    __shared__ int hist[I_HIST]; // first: this workgroup's pairs per split; then: its first slot in the split's run
    const bool use_lds = nsplit <= I_HIST;
    if (use_lds)
        for (int s = threadIdx.x; s < nsplit; s += blockDim.x) hist[s] = 0;
    __syncthreads();
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int sl[I_LIVE], rank[I_LIVE];
#pragma unroll
    for (int l = 0; l < I_LIVE; ++l) {
        sl[l] = i < m1 ? live[i * I_LIVE + l] : -1;
        rank[l] = 0;
        if (sl[l] >= 0) rank[l] = use_lds ? atomicAdd(&hist[sl[l]], 1) : atomicAdd(&split_cursor[sl[l]], 1);
    }
    __syncthreads();
    if (use_lds) {
        for (int s = threadIdx.x; s < nsplit; s += blockDim.x)
            if (hist[s]) hist[s] = atomicAdd(&split_cursor[s], hist[s]);
        __syncthreads();
    }
#pragma unroll
    for (int l = 0; l < I_LIVE; ++l) {
        if (sl[l] < 0) continue;
        const int p = split_base[sl[l]] + (use_lds ? hist[sl[l]] : 0) + rank[l];
        pair_row[p] = (int)i;
        pair_thr[p] = kmin[i] + win[i];
        live[i * I_LIVE + l] = p; // (from here on: the row's pair slots)
    }
}

// Step 4.  A workgroup = 256 pairs of ONE split (blk_split[block]; pair slots 256 block ..; pair_row < 0: padding), scanning that
// split's tiles with the final thresholds.  Keys within the threshold are appended to the pair's list.
template <int KS>
__global__ __launch_bounds__(512, 1) void k_i8_collect(const unsigned char *__restrict__ ai, const unsigned char *__restrict__ bi,
                                                        const int *__restrict__ nbs, const int2 *__restrict__ split_tiles,
                                                        int64_t n_blocks, const int *__restrict__ blk_split,
                                                        const int *__restrict__ pair_row, const int *__restrict__ pair_thr,
                                                        int *__restrict__ cnt, int32_t *__restrict__ cand_j,
                                                        int *__restrict__ cand_k)
{
    const int64_t vb = sf_xcd_block();
    if (vb >= n_blocks) return;
    const int64_t split = blk_split[vb];
    SF_I_GEOMETRY
    const int64_t p0 = vb * 256 + 32 * wave;
    i4v af[KS];
    {
        const int row = pair_row[p0 + r31];
        const unsigned char *ap = ai + (int64_t)(row < 0 ? 0 : row) * DP + 16 * h;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) af[ks] = *reinterpret_cast<const i4v *>(ap + 32 * ks);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) asm volatile("" : "+v"(af[ks]));
    }
    i16v T; // thresholds of the 16 pairs this lane sees in an accumulator: pair p0 + (r & 3) + 8 (r >> 2) + 4 h
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int64_t p = p0 + (r & 3) + 8 * (r >> 2) + 4 * h;
        T[r] = pair_row[p] < 0 ? -I_BIG : pair_thr[p];
    }
    const int2 st = split_tiles[split];
    const int64_t jt0 = st.x, ntiles = st.y;
    SF_I_DMA(jt0, 0)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int64_t jt = jt0; jt < ntiles; ++jt) {
        const int buf = (int)((jt - jt0) & 1);
        if (jt + 1 < ntiles) SF_I_DMA(jt + 1, buf ^ 1)
        const unsigned char *bp = Bs + buf * TILE_BYTES;
        constexpr int PF = 3;
        const int *nbl = reinterpret_cast<const int *>(bp + IN * PC * 16);
#pragma unroll 1
        for (int cb = 0; cb < 2; ++cb) {
            i4v q[PF];
#pragma unroll
            for (int i = 0; i < PF; ++i) q[i] = SF_I_FRAG(i, cb);
            i16v acc = T; // the accumulator starts at the threshold: key <= T  <=>  T + D >= NB
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(af[ks], q[ks % PF], acc, 0, 0, 0);
                if (ks + PF < KS) q[ks % PF] = SF_I_FRAG(ks + PF, cb);
            }
            const int nbv = nbl[32 * cb + r31];
            int mx = acc[0];
#pragma unroll
            for (int r = 1; r < 16; ++r) mx = max(mx, acc[r]);
            if (__ballot(mx >= nbv && nbv < I_REAL)) {
                const int64_t j = jt * IN + 32 * cb + r31;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    if (acc[r] >= nbv && nbv < I_REAL) {
                        const int64_t p = p0 + (r & 3) + 8 * (r >> 2) + 4 * h;
                        const int s = atomicAdd(&cnt[p], 1);
                        if (s < ICAP) {
                            cand_j[p * ICAP + s] = (int32_t)j;
                            cand_k[p * ICAP + s] = nbv - (acc[r] - T[r]);
                        }
                    }
                }
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
}
#undef SF_I_DMA
#undef SF_I_DMA16
#undef SF_I_DMA4
#undef SF_I_FRAG
#undef SF_I_GEOMETRY

// Step 5: the reference's arithmetic on the candidates of a row's pairs (as k_half_final: LPR lanes share a scan row).
template <int LPR>
__global__ void k_i8_final(const double *__restrict__ a, int64_t m1, const double *__restrict__ b, int64_t d,
                           const unsigned char *__restrict__ a_ok, const int *__restrict__ live, const int *__restrict__ cnt,
                           const int32_t *__restrict__ cand_j, const int *__restrict__ cand_k, const int *__restrict__ win,
                           const double *__restrict__ na2, double unit, int64_t *__restrict__ idx, double *__restrict__ dist,
                           int *__restrict__ flag, int *__restrict__ n_flagged)
{
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t i = gid / LPR;
    const int sub = (int)(gid % LPR);
    if (i >= m1) return; // (whole groups of LPR lanes: LPR divides the wave)
    if ((a_ok && !a_ok[i]) || flag[i]) return; // (masked: written by k_i8_live; flagged there: the FP16 pass's)
    double best = INFINITY;
    int64_t bj = -1;
    bool ok = true;
    const double *ai = a + i * d;
    const double slack = (double)win[i], na = na2[i];
    for (int l = 0; l < I_LIVE; ++l) {
        const int p = live[i * I_LIVE + l];
        if (p < 0) continue;
        const int n = cnt[p];
        if (n > ICAP) { ok = false; continue; } // more columns within the window than the list holds: the FP16 pass
        const int64_t base = (int64_t)p * ICAP;
        for (int c = sub; c < n; c += LPR) {
            const int64_t j = cand_j[base + c];
            const double *bjp = b + j * d;
            double acc = 0.0;
            for (int64_t u = 0; u < d; ++u) {
                const double df = ai[u] - bjp[u];
                acc += df * df; // left to right, no FMA: scipy's euclidean loop
            }
            // safety net for the error model: the integer key of this pair must be within eps_i / u + 1/2 (< W_i / 2) of the
            // float64 one, (||a - b||^2 - ||a||^2) / u; a row where it is not is handed on
            ok &= fabs((acc - na) * unit - (double)cand_k[base + c]) <= 0.5 * slack;
            const double dj = sqrt(acc);
            if (dj < best || (dj == best && j < bj) || bj < 0) {
                if (!(dj == dj)) continue; // NaN: leave the row to the float64 path
                best = dj;
                bj = j;
            }
        }
    }
#pragma unroll
    for (int off = LPR / 2; off > 0; off >>= 1) { // (minimum with the smaller column on ties: the order of the fold does not matter)
        const double ob = __shfl_xor(best, off);
        const int64_t oj = __shfl_xor(bj, off);
        const int om = __shfl_xor((int)ok, off);
        if (oj >= 0 && (bj < 0 || ob < best || (ob == best && oj < bj))) {
            best = ob;
            bj = oj;
        }
        ok = ok && om;
    }
    if (sub != 0) return;
    const bool decided = bj >= 0 && ok;
    idx[i] = decided ? bj : 0;
    if (dist) dist[i] = best;
    if (!decided) {
        flag[i] = 1;
        atomicAdd(n_flagged, 1);
    }
}

__global__ void k_i8_gather_rows(const double *__restrict__ a, int64_t d, const int64_t *__restrict__ rows, int64_t nr,
                                 double *__restrict__ out)
{
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= nr * d) return;
    const int64_t r = g / d, t = g - r * d;
    out[g] = a[rows[r] * d + t];
}

__global__ void k_i8_scatter(const int64_t *__restrict__ rows, int64_t nr, const int64_t *__restrict__ sidx,
                             const double *__restrict__ sdist, int64_t *__restrict__ idx, double *__restrict__ dist)
{
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= nr) return;
    idx[rows[g]] = sidx[g];
    if (dist) dist[rows[g]] = sdist[g];
}

int i8_host_max(sf_ctx *ctx, const double *v, int64_t n, double *part, double *out)
{
    SF_LAUNCH(ctx, "k8_i8_max", k_i8_max, dim3(256), dim3(256), v, n, part);
    std::vector<double> h(256);
    SF_HIP(hipMemcpyAsync(h.data(), part, 256 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    SF_HIP(hipStreamSynchronize(ctx->stream));
    double mx = 0.0;
    for (double x : h) mx = std::max(mx, x);
    *out = mx;
    return SF_OK;
}

} // namespace

// SF_MATCH_I8=0 disables the integer pre-filter, =1 forces it for every problem that reaches the matrix-core paths.
int sf_match_i8_mode()
{
    const char *e = getenv("SF_MATCH_I8");
    if (!e || !e[0]) return -1;
    return e[0] == '0' ? 0 : 1;
}

// ---- host side ------------------------------------------------------------------------------------------------------------------
// One implementation behind two entry points: sf_match_i8 (every reference row present: convert, pass 1, decide -- with a pilot
// slab of scan rows in front) and the STREAMED form sf_match_stream_* (match.hip's caller: the reference rows arrive in chunks --
// an all-gather in flight -- and every chunk gets its conversion and its share of pass 1 while the next one travels; the decision
// steps run once, over the minima of all chunks).  A column split is a run of 64-column tiles inside ONE fed range; the table of
// splits grows with the feeds.
struct sf_match_stream {
    sf_ctx *ctx = nullptr;
    const double *da = nullptr, *db = nullptr;
    const unsigned char *a_ok = nullptr, *b_ok = nullptr;
    int64_t m1 = 0, m2 = 0, d = 0, m1p = 0, m2p = 0;
    int ks = 11, dp = 352, RB = 2, IM = 512;
    double sa = 0.0, sb = 0.0, unit = 0.0;
    bool integer = false;   // the integer pass serves this problem (else: sf_match_stream_end runs the generic path)
    bool windowed = false;
    unsigned char *ai = nullptr, *bi = nullptr;
    double *ea = nullptr, *qa = nullptr, *na2 = nullptr, *eb = nullptr, *qb = nullptr, *nb2 = nullptr, *part = nullptr, *amx = nullptr;
    int *nbi = nullptr, *win = nullptr, *smin = nullptr, *kmin = nullptr, *live = nullptr, *flag = nullptr, *counters = nullptr;
    int2 *split_tiles = nullptr;
    int64_t split_cap = 0, tiles_target = 0;
    std::vector<int2> splits;
    std::vector<void *> owned;
    double nbmax = 0.0, ebmax = 0.0;
};

namespace {

template <typename T>
int st_alloc(sf_match_stream *st, T **p, size_t count)
{
    SF_CHECK(sf_palloc(st->ctx, p, count ? count : 1));
    st->owned.push_back(*p);
    return SF_OK;
}

void st_free(sf_match_stream *st)
{
    if (!st) return;
    for (void *p : st->owned) sf_pool_release(st->ctx, p);
    delete st;
}

// scales, the scan side's image, the buffers of the reference side.  *suitable = false: leave the problem to the other paths.
int st_begin(sf_match_stream *st, double b_entry_max, int64_t max_ranges, bool *suitable)
{
    sf_ctx *ctx = st->ctx;
    *suitable = false;
    const int64_t m1 = st->m1, m2 = st->m2, d = st->d;
    if (d > 352 || m1 <= 0 || m2 <= 0 || m2 > 0x7fffffff || m1 > 0x3fffffff) return SF_OK;
    st->RB = 2; // 32-row blocks per wave of the first pass (one per wave: + 11 % on the pass, measured; the instantiation is gone)
    st->IM = 256 * st->RB;
    st->ks = d <= 128 ? 4 : 11;
    st->dp = 32 * st->ks;
    st->m1p = sf_div_up(m1, 512) * 512;
    st->m2p = sf_div_up(m2, IN) * IN;
    const int64_t m1p = st->m1p, m2p = st->m2p;
    SF_CHECK(st_alloc(st, &st->part, 256));
    SF_CHECK(st_alloc(st, &st->na2, m1p)); SF_CHECK(st_alloc(st, &st->ea, m1p)); SF_CHECK(st_alloc(st, &st->qa, m1p));
    SF_CHECK(st_alloc(st, &st->amx, std::max(m1, m2)));
    double namax = 0.0, aamax = 0.0;
    SF_LAUNCH(ctx, "k8_i8_convert", k_i8_rowstat, dim3((unsigned)sf_div_up(m1, 4)), dim3(256), st->da, m1, d, st->na2, st->amx);
    SF_CHECK(i8_host_max(ctx, st->na2, m1, st->part, &namax));
    SF_CHECK(i8_host_max(ctx, st->amx, m1, st->part, &aamax));
    if (!(aamax > 0.0) || !(b_entry_max > 0.0) || !std::isfinite(namax) || !std::isfinite(aamax) || !std::isfinite(b_entry_max)) return SF_OK;
    st->sa = 127.0 / aamax;
    st->sb = 127.0 / b_entry_max;
    st->unit = 0.5 * st->sa * st->sb;
    if (!std::isfinite(st->sa) || !std::isfinite(st->sb) || !std::isfinite(st->unit) || !(st->unit > 0.0)) return SF_OK;
    SF_CHECK(st_alloc(st, &st->ai, m1p * st->dp)); SF_CHECK(st_alloc(st, &st->bi, m2p * st->dp));
    SF_CHECK(st_alloc(st, &st->nb2, m2p)); SF_CHECK(st_alloc(st, &st->eb, m2p)); SF_CHECK(st_alloc(st, &st->qb, m2p)); SF_CHECK(st_alloc(st, &st->nbi, m2p));
    if (m1p > m1) SF_HIP(hipMemsetAsync(st->na2 + m1, 0, (size_t)(m1p - m1) * sizeof(double), ctx->stream));
    SF_LAUNCH(ctx, "k8_i8_convert", k_i8_convert, dim3((unsigned)sf_div_up(m1p, 4)), dim3(256), st->da, m1, m1p, d, st->dp, st->sa,
              (const unsigned char *)nullptr, reinterpret_cast<unsigned *>(st->ai), st->ea, st->qa, (const double *)st->na2, (int *)nullptr, st->unit);
    // column splits: 8 MB of reference rows each, so that an XCD's workgroups share the split they stream in their L2 (as
    // sf_match_half); with few row blocks, enough splits to fill the chip
    const int64_t col_tiles = m2p / IN;
    const int64_t chunk_kb = 8192;
    const int64_t tiles_in_l2 = std::max<int64_t>(8, chunk_kb * 1024 / ((int64_t)IN * st->dp));
    int64_t nsplit = sf_div_up(col_tiles, tiles_in_l2);
    if ((m1p / st->IM) * nsplit < 512) nsplit = std::max<int64_t>(nsplit, std::min<int64_t>(sf_div_up(512, m1p / st->IM), std::max<int64_t>(col_tiles / 32, 1)));
    if (const char *e = getenv("SF_MATCH_I8_SPLITS")) nsplit = std::max<int64_t>(1, std::min<int64_t>(atoll(e), col_tiles));
    st->tiles_target = sf_div_up(col_tiles, nsplit);
    st->split_cap = sf_div_up(col_tiles, st->tiles_target) + std::max<int64_t>(max_ranges, 1);
    if (st->split_cap > 65536) return SF_OK;
    SF_CHECK(st_alloc(st, &st->win, m1p)); SF_CHECK(st_alloc(st, &st->smin, st->split_cap * m1p)); SF_CHECK(st_alloc(st, &st->kmin, m1p));
    SF_CHECK(st_alloc(st, &st->live, m1p * I_LIVE)); SF_CHECK(st_alloc(st, &st->flag, m1));
    SF_CHECK(st_alloc(st, &st->counters, 2 * st->split_cap + 2)); // [0]: flagged rows, [1]: unused, then split_count, split_cursor
    SF_CHECK(st_alloc(st, &st->split_tiles, st->split_cap));
    SF_HIP(hipMemsetAsync(st->counters, 0, (size_t)(2 * st->split_cap + 2) * sizeof(int), ctx->stream));
    *suitable = true;
    st->integer = true;
    return SF_OK;
}

// reference rows [rb, re) (rb a multiple of 64; re a multiple of 64 or m2) are in place: their int8 image, their norm terms,
// their splits.  *first / *count: the new splits.
int st_convert_cols(sf_match_stream *st, int64_t rb, int64_t re, int64_t *first, int64_t *count)
{
    sf_ctx *ctx = st->ctx;
    const int64_t rows = re - rb, rows_p = sf_div_up(rows, IN) * IN; // (the last range is padded to whole tiles: masked columns)
    if (rb % IN || rows <= 0 || re > st->m2 || (re % IN && re != st->m2)) { sf_set_error("sf_match_stream_feed: rows [%lld, %lld) are not whole 64-row tiles", (long long)rb, (long long)re); return SF_ERR_ARG; }
    SF_LAUNCH(ctx, "k8_i8_convert", k_i8_rowstat, dim3((unsigned)sf_div_up(rows, 4)), dim3(256), st->db + rb * st->d, rows, st->d, st->nb2 + rb, st->amx);
    if (rows_p > rows) SF_HIP(hipMemsetAsync(st->nb2 + re, 0, (size_t)(rows_p - rows) * sizeof(double), ctx->stream));
    SF_LAUNCH(ctx, "k8_i8_convert", k_i8_convert, dim3((unsigned)sf_div_up(rows_p, 4)), dim3(256), st->db + rb * st->d, rows, rows_p, st->d, st->dp, st->sb,
              st->b_ok ? st->b_ok + rb : st->b_ok, reinterpret_cast<unsigned *>(st->bi + rb * st->dp), st->eb + rb, st->qb + rb,
              (const double *)(st->nb2 + rb), st->nbi + rb, st->unit);
    const int64_t t0 = rb / IN, t1 = t0 + rows_p / IN, n = sf_div_up(t1 - t0, st->tiles_target), per = sf_div_up(t1 - t0, n);
    *first = (int64_t)st->splits.size();
    for (int64_t t = t0; t < t1; t += per) st->splits.push_back(make_int2((int)t, (int)std::min(t + per, t1)));
    *count = (int64_t)st->splits.size() - *first;
    if ((int64_t)st->splits.size() > st->split_cap) { sf_set_error("sf_match_stream_feed: more ranges than announced"); return SF_ERR_ARG; }
    SF_HIP(hipMemcpyAsync(st->split_tiles + *first, st->splits.data() + *first, (size_t)*count * sizeof(int2), hipMemcpyHostToDevice, ctx->stream));
    SF_HIP(hipStreamSynchronize(ctx->stream)); // (the vector may grow, and move, with the next feed)
    return SF_OK;
}

// pass 1 of the scan rows [r0, r0 + ms) over the splits [first, first + count)
int st_pass1(sf_match_stream *st, const char *name, int64_t first, int64_t count, int64_t r0, int64_t ms)
{
    sf_ctx *ctx = st->ctx;
    const int64_t msp = sf_div_up(ms, st->IM) * st->IM, row_blocks = msp / st->IM;
    const int64_t wgs = sf_xcd_grid(count * row_blocks);
    if (wgs > 0x7fffffffLL) { sf_set_error("sf_match_i8: %lld workgroups exceed a launch", (long long)wgs); return SF_ERR_UNSUPPORTED; }
    if (!count || !ms) return SF_OK;
    // (smin rows are addressed split * m1p + row: a slab of scan rows writes its own rows of every split's stripe)
#define SF_I8_MIN(KS_, RB_)                                                                                             \
    SF_LAUNCH(ctx, name, (k_i8_min<KS_, RB_>), dim3((unsigned)wgs), dim3(512), (const unsigned char *)(st->ai + r0 * st->dp), \
              (const unsigned char *)st->bi, (const int *)st->nbi, (const int2 *)st->split_tiles, first, st->m1p, row_blocks, count, st->smin + r0)
    if (st->ks == 4) { SF_I8_MIN(4, 2); } else { SF_I8_MIN(11, 2); }
#undef SF_I8_MIN
    return SF_OK;
}

// every reference row has been fed: the maxima the windows need.  *suitable = false: norms the integer keys cannot hold.
int st_window(sf_match_stream *st, bool *suitable)
{
    sf_ctx *ctx = st->ctx;
    *suitable = false;
    SF_CHECK(i8_host_max(ctx, st->nb2, st->m2, st->part, &st->nbmax));
    SF_CHECK(i8_host_max(ctx, st->eb, st->m2, st->part, &st->ebmax));
    if (!std::isfinite(st->nbmax) || !std::isfinite(st->ebmax) || !(st->nbmax * st->unit < 4e6)) return SF_OK;
    SF_LAUNCH(ctx, "k8_i8_window", k_i8_window, dim3((unsigned)sf_div_up(st->m1p, 256)), dim3(256), (const double *)st->ea, (const double *)st->qa,
              (const double *)st->na2, st->m1, st->m1p, std::sqrt(st->nbmax), st->ebmax, st->nbmax, st->unit, st->win);
    st->windowed = true;
    *suitable = true;
    return SF_OK;
}

// steps 3-5 for the scan rows [r0, r0 + ms): live splits, pairs, collect pass, decision.  Rows that end flagged (counted in
// counters[0]) are left to the caller.
int st_decide(sf_match_stream *st, int64_t r0, int64_t ms, int64_t *didx, double *ddist)
{
    sf_ctx *ctx = st->ctx;
    const int64_t nsplit = (int64_t)st->splits.size(), m1p = st->m1p;
    int *nflag = st->counters, *split_count = st->counters + 2, *split_cursor = st->counters + 2 + st->split_cap;
    SF_HIP(hipMemsetAsync(split_count, 0, (size_t)(2 * st->split_cap) * sizeof(int), ctx->stream)); // (counts and cursors)
    SF_LAUNCH(ctx, "k8_i8_live", k_i8_live, dim3((unsigned)sf_div_up(ms, 256)), dim3(256), (const int *)(st->smin + r0), (const int *)(st->win + r0),
              st->a_ok ? st->a_ok + r0 : st->a_ok, ms, m1p, (int)nsplit, st->kmin + r0, st->live + r0 * I_LIVE, split_count, didx + r0,
              ddist ? ddist + r0 : ddist, st->flag + r0, nflag);
    // the pairs, split by split, each split's run padded to whole workgroups of 256
    std::vector<int> hcount((size_t)nsplit), hbase((size_t)nsplit), hblk;
    SF_HIP(hipMemcpyAsync(hcount.data(), split_count, (size_t)nsplit * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    SF_HIP(hipStreamSynchronize(ctx->stream));
    int64_t n_pairs_pad = 0;
    for (int64_t s = 0; s < nsplit; ++s) {
        hbase[(size_t)s] = (int)n_pairs_pad;
        const int64_t nb = sf_div_up(hcount[(size_t)s], 256);
        for (int64_t k = 0; k < nb; ++k) hblk.push_back((int)s);
        n_pairs_pad += nb * 256;
        if (n_pairs_pad > 0x7fffff00LL) { sf_set_error("sf_match_i8: too many (row, split) pairs"); return SF_ERR_UNSUPPORTED; }
    }
    const int64_t n_blocks = (int64_t)hblk.size();
    sf_pool_guard ptmp(ctx);
    int *split_base = nullptr, *blk_split = nullptr, *pair_row = nullptr, *pair_thr = nullptr, *cnt = nullptr, *candk = nullptr;
    int32_t *candj = nullptr;
    const size_t np1 = (size_t)std::max<int64_t>(n_pairs_pad, 1);
    SF_CHECK(ptmp.alloc(&split_base, (size_t)nsplit)); SF_CHECK(ptmp.alloc(&blk_split, (size_t)std::max<int64_t>(n_blocks, 1)));
    SF_CHECK(ptmp.alloc(&pair_row, np1)); SF_CHECK(ptmp.alloc(&pair_thr, np1)); SF_CHECK(ptmp.alloc(&cnt, np1));
    SF_CHECK(ptmp.alloc(&candj, np1 * ICAP)); SF_CHECK(ptmp.alloc(&candk, np1 * ICAP));
    SF_HIP(hipMemcpyAsync(split_base, hbase.data(), (size_t)nsplit * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
    if (n_blocks) {
        SF_HIP(hipMemcpyAsync(blk_split, hblk.data(), (size_t)n_blocks * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
        SF_HIP(hipMemsetAsync(pair_row, 0xff, (size_t)n_pairs_pad * sizeof(int), ctx->stream)); // -1: padding
        SF_HIP(hipMemsetAsync(cnt, 0, (size_t)n_pairs_pad * sizeof(int), ctx->stream));
        SF_LAUNCH(ctx, "k8_i8_live", k_i8_place, dim3((unsigned)sf_div_up(ms, 256)), dim3(256), (const int *)(st->kmin + r0), (const int *)(st->win + r0),
                  st->live + r0 * I_LIVE, ms, (int)nsplit, (const int *)split_base, split_cursor, pair_row, pair_thr);
        const int64_t cw = sf_xcd_grid(n_blocks);
        if (st->ks == 4) {
            SF_LAUNCH(ctx, "k8_i8_collect", k_i8_collect<4>, dim3((unsigned)cw), dim3(512), (const unsigned char *)(st->ai + r0 * st->dp),
                      (const unsigned char *)st->bi, (const int *)st->nbi, (const int2 *)st->split_tiles, n_blocks, (const int *)blk_split,
                      (const int *)pair_row, (const int *)pair_thr, cnt, candj, candk);
        } else {
            SF_LAUNCH(ctx, "k8_i8_collect", k_i8_collect<11>, dim3((unsigned)cw), dim3(512), (const unsigned char *)(st->ai + r0 * st->dp),
                      (const unsigned char *)st->bi, (const int *)st->nbi, (const int2 *)st->split_tiles, n_blocks, (const int *)blk_split,
                      (const int *)pair_row, (const int *)pair_thr, cnt, candj, candk);
        }
    }
#define SF_I8_FINAL(LPR)                                                                                               \
    SF_LAUNCH(ctx, "k8_i8_final", k_i8_final<LPR>, dim3((unsigned)sf_div_up(ms * LPR, 256)), dim3(256), st->da + r0 * st->d, ms, st->db, st->d, \
              st->a_ok ? st->a_ok + r0 : st->a_ok, (const int *)(st->live + r0 * I_LIVE), (const int *)cnt, (const int32_t *)candj, \
              (const int *)candk, (const int *)(st->win + r0), (const double *)(st->na2 + r0), st->unit, didx + r0,          \
              ddist ? ddist + r0 : ddist, st->flag + r0, nflag)
    if (ms <= 65536) { SF_I8_FINAL(16); } else { SF_I8_FINAL(1); }
#undef SF_I8_FINAL
    SF_HIP(hipStreamSynchronize(ctx->stream)); // (hbase / hblk are host buffers of the copies above)
    return SF_OK;
}

int st_flagged(sf_match_stream *st, int *nf)
{
    SF_HIP(hipMemcpyAsync(nf, st->counters, sizeof(int), hipMemcpyDeviceToHost, st->ctx->stream));
    SF_HIP(hipStreamSynchronize(st->ctx->stream));
    return SF_OK;
}

// the flagged rows -- no clear nearest descriptor, or overflowing lists -- through the FP16 pass on the gathered rows
int st_fallback(sf_match_stream *st, int nf, int64_t *didx, double *ddist, int64_t *n_slow)
{
    sf_ctx *ctx = st->ctx;
    const int64_t m1 = st->m1, d = st->d;
    if (n_slow) *n_slow = 0;
    if (nf <= 0) return SF_OK;
    std::vector<int> hflag((size_t)m1);
    SF_HIP(hipMemcpyAsync(hflag.data(), st->flag, (size_t)m1 * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    SF_HIP(hipStreamSynchronize(ctx->stream));
    std::vector<int64_t> rows;
    rows.reserve((size_t)nf);
    for (int64_t i = 0; i < m1; ++i)
        if (hflag[(size_t)i]) rows.push_back(i);
    const int64_t nr = (int64_t)rows.size();
    sf_pool_guard tmp(ctx);
    int64_t *drows = nullptr, *sidx = nullptr;
    double *sub = nullptr, *sdist = nullptr;
    SF_CHECK(tmp.alloc(&drows, (size_t)nr)); SF_CHECK(tmp.alloc(&sidx, (size_t)nr)); SF_CHECK(tmp.alloc(&sdist, (size_t)nr)); SF_CHECK(tmp.alloc(&sub, (size_t)(nr * d)));
    SF_HIP(hipMemcpyAsync(drows, rows.data(), (size_t)nr * sizeof(int64_t), hipMemcpyHostToDevice, ctx->stream));
    SF_LAUNCH(ctx, "k8_gather_rows", k_i8_gather_rows, dim3((unsigned)sf_div_up(nr * d, 256)), dim3(256), st->da, d, (const int64_t *)drows, nr, sub);
    SF_HIP(hipStreamSynchronize(ctx->stream)); // rows.data() is a host buffer
    int64_t slow2 = 0;
    int used2 = 0;
    int rc = sf_match_half(ctx, sub, nr, st->db, st->m2, d, sidx, sdist, "k8_match_half", &slow2, nullptr, st->b_ok, &used2);
    if (rc == SF_OK && !used2) // (norms the FP16 image cannot hold: float64 all the way)
        rc = sf_match_gemm_f64(ctx, sub, nr, st->db, st->m2, d, sidx, sdist, "k8_match_gemm_overflow", &slow2, nullptr, st->b_ok);
    if (rc == SF_OK) {
        SF_LAUNCH(ctx, "k8_scatter_results", k_i8_scatter, dim3((unsigned)sf_div_up(nr, 256)), dim3(256), (const int64_t *)drows, nr,
                  (const int64_t *)sidx, (const double *)sdist, didx, ddist);
    }
    if (n_slow) *n_slow = nr;
    return rc;
}

} // namespace

// rc SF_OK and *used = 1 when the result has been produced (flagged rows through the FP16 pass included); *used = 0
// (nothing the caller relies on written) when the input is not suitable -- d > 352, zero / non-finite entries, or a pilot slab
// that says the rows have no clear nearest descriptor -- and the caller must take the FP16 path.
int sf_match_i8(sf_ctx *ctx, const double *da, int64_t m1, const double *db, int64_t m2, int64_t d, int64_t *didx, double *ddist,
                const char *name, int64_t *n_slow, const unsigned char *a_ok, const unsigned char *b_ok, int *used)
{
    *used = 0;
    if (d > 352 || m1 <= 0 || m2 <= 0 || m2 > 0x7fffffff || m1 > 0x3fffffff) return SF_OK;
    sf_match_stream *st = new sf_match_stream();
    st->ctx = ctx; st->da = da; st->db = db; st->a_ok = a_ok; st->b_ok = b_ok; st->m1 = m1; st->m2 = m2; st->d = d;
    struct guard { sf_match_stream *s; ~guard() { st_free(s); } } g{st};
    // the largest |entry| of the reference side: the scale of its image
    double abmax = 0.0;
    {
        sf_pool_guard tmp(ctx);
        double *n2 = nullptr, *mx = nullptr, *part = nullptr;
        SF_CHECK(tmp.alloc(&n2, (size_t)m2)); SF_CHECK(tmp.alloc(&mx, (size_t)m2)); SF_CHECK(tmp.alloc(&part, 256));
        SF_LAUNCH(ctx, "k8_i8_convert", k_i8_rowstat, dim3((unsigned)sf_div_up(m2, 4)), dim3(256), db, m2, d, n2, mx);
        SF_CHECK(i8_host_max(ctx, mx, m2, part, &abmax));
    }
    bool ok = false;
    SF_CHECK(st_begin(st, abmax, 1, &ok));
    if (!ok) return SF_OK;
    int64_t first = 0, count = 0;
    SF_CHECK(st_convert_cols(st, 0, m2, &first, &count));
    SF_CHECK(st_window(st, &ok));
    if (!ok) return SF_OK;
    // the first slab of scan rows is the PILOT when the problem is many times its size: its flagged share decides whether the
    // integer pass pays
    static const double max_flagged = [] { const char *e = getenv("SF_I8_MAX_FLAGGED"); const double v = e ? atof(e) : 0.35; return v > 0.0 ? v : 0.35; }();
    // (SF_I8_PILOT_ROWS: the pilot's size, for tests -- it then runs in the forced mode too)
    const char *pilot_env = getenv("SF_I8_PILOT_ROWS");
    const int64_t pilot_rows = pilot_env ? std::max<int64_t>(st->IM, sf_div_up(atoll(pilot_env), st->IM) * st->IM) : (int64_t)st->IM * 64;
    const bool pilot = pilot_env ? m1 > pilot_rows : (m1 >= 4 * pilot_rows && sf_match_i8_mode() != 1);
    int64_t done = 0;
    int nf = 0;
    if (pilot) {
        SF_CHECK(st_pass1(st, name, first, count, 0, pilot_rows));
        SF_CHECK(st_decide(st, 0, pilot_rows, didx, ddist));
        SF_CHECK(st_flagged(st, &nf));
        if ((double)nf > max_flagged * (double)pilot_rows) return SF_OK; // (*used = 0: the caller's pass overwrites what was written)
        done = pilot_rows;
    }
    SF_CHECK(st_pass1(st, name, first, count, done, m1 - done));
    SF_CHECK(st_decide(st, done, m1 - done, didx, ddist));
    SF_CHECK(st_flagged(st, &nf));
    const int rc = st_fallback(st, nf, didx, ddist, n_slow);
    *used = 1;
    return rc;
}

// ---- the streamed form (include/shotfpfh.h: sf_match_stream_*) ---------------------------------------------------------------------
int sf_match_argmin_masked_generic(sf_ctx *ctx, const double *a, const double *b, int64_t m1, int64_t m2, int64_t d,
                                   const unsigned char *a_ok, const unsigned char *b_ok, int64_t *idx, double *dist); // match.hip

extern "C" sf_match_stream *sf_match_stream_begin(sf_ctx *ctx, const double *a_dev, const unsigned char *a_ok_dev, int64_t m1,
                                                  const double *b_dev, const unsigned char *b_ok_dev, int64_t m2, int64_t d,
                                                  double b_entry_max, int64_t max_ranges)
{
    if (!ctx || !a_dev || !b_dev || !a_ok_dev || !b_ok_dev || m1 < 0 || m2 <= 0 || d <= 0) { sf_set_error("sf_match_stream_begin: bad argument"); return nullptr; }
    if (hipSetDevice(ctx->device) != hipSuccess) { sf_set_error("hipSetDevice failed"); return nullptr; }
    sf_match_stream *st = new sf_match_stream();
    st->ctx = ctx; st->da = a_dev; st->db = b_dev; st->a_ok = a_ok_dev; st->b_ok = b_ok_dev; st->m1 = m1; st->m2 = m2; st->d = d;
    bool ok = false;
    const bool wanted = sf_match_i8_mode() != 0 && m1 > 0;
    if (wanted && st_begin(st, b_entry_max, max_ranges, &ok) != SF_OK) { st_free(st); return nullptr; }
    // (not suitable -- or switched off: the handle stays, feeds do nothing, sf_match_stream_end runs the one-shot paths)
    return st;
}

extern "C" int sf_match_stream_feed(sf_ctx *ctx, sf_match_stream *st, int64_t row_begin, int64_t row_end)
{
    if (!ctx || !st || st->ctx != ctx) { sf_set_error("sf_match_stream_feed: bad argument"); return SF_ERR_ARG; }
    if (!st->integer) return SF_OK;
    SF_HIP(hipSetDevice(ctx->device));
    int64_t first = 0, count = 0;
    SF_CHECK(st_convert_cols(st, row_begin, row_end, &first, &count));
    return st_pass1(st, "k8_match_i8", first, count, 0, st->m1);
}

extern "C" int sf_match_stream_end(sf_ctx *ctx, sf_match_stream *st, int64_t *idx_dev, double *dist_dev)
{
    if (!ctx || !st || st->ctx != ctx || !idx_dev) { sf_set_error("sf_match_stream_end: bad argument"); return SF_ERR_ARG; }
    struct guard { sf_match_stream *s; ~guard() { st_free(s); } } g{st};
    SF_HIP(hipSetDevice(ctx->device));
    if (st->integer) {
        // every reference row must have been fed exactly once: the splits' tiles add up to the padded row count
        int64_t tiles = 0;
        for (const int2 &sp : st->splits) tiles += sp.y - sp.x;
        if (tiles != st->m2p / IN) { sf_set_error("sf_match_stream_end: %lld of %lld column tiles were fed", (long long)tiles, (long long)(st->m2p / IN)); return SF_ERR_STATE; }
        bool ok = false;
        SF_CHECK(st_window(st, &ok));
        if (ok) {
            static const double max_flagged = [] { const char *e = getenv("SF_I8_MAX_FLAGGED"); const double v = e ? atof(e) : 0.35; return v > 0.0 ? v : 0.35; }();
            int nf = 0;
            SF_CHECK(st_decide(st, 0, st->m1, idx_dev, dist_dev));
            SF_CHECK(st_flagged(st, &nf));
            if ((double)nf <= max_flagged * (double)st->m1 || sf_match_i8_mode() == 1) return st_fallback(st, nf, idx_dev, dist_dev, nullptr);
            // (most rows without a clear nearest descriptor: the one-shot paths for everything)
        }
    }
    return sf_match_argmin_masked_generic(ctx, st->da, st->db, st->m1, st->m2, st->d, st->a_ok, st->b_ok, idx_dev, dist_dev);
}

extern "C" void sf_match_stream_abort(sf_ctx *ctx, sf_match_stream *st)
{
    (void)ctx;
    st_free(st);
}
