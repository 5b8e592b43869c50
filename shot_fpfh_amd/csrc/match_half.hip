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
//      with probability ~ 1/t, so the pass is MFMA + two vector instructions per pair.
//   3. k_half_final: per row, the candidates still within W_i of the final minimum (typically 1-3) get the
//      reference's distance -- sequential float64 sum, square root -- and the smallest (lowest column on
//      ties) wins, exactly scipy's first-minimum rule.
//
// Exactness.  Let eps_i bound |k(i, j) - key(i, j)| over j, with key the exact ||b_j||^2 - 2 a_i . b_j:
//     eps_i = 2 (ea_i Bmax + qa_i EBmax + gamma qa_i QBmax)        ea_i = ||a_i - a'_i||, qa_i = ||a'_i||,
//     Bmax = max ||b_j||, EBmax = max ||b_j - b'_j||, QBmax = max ||b'_j||, gamma = (d + 32) 2^-22
// (Cauchy-Schwarz on (a - a').b + a'.(b - b'); FP16 x FP16 products are exact in FP32, gamma covers the FP32
// accumulation of the matrix core however it rounds).  The reference's arg-min j* has an exact key no larger
// than that of the column j1 that set the final threshold, plus float64 rounding (1e-12 relative, folded into
// eps).  Hence k(i, j*) <= k(i, j1) + 2 eps_i, and with W_i = 2 eps_i + 8 eta_i (eta_i: the FP32 rounding of the
// key / threshold arithmetic in the epilogue) j* passes the threshold test at the moment it is scanned and the
// final filter.  So the candidate set always contains the reference's arg-min and everything that ties with
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
                                                      float *__restrict__ n2f)
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
        out[i * dp + t] = hv;
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
            n2f[i] = masked ? INFINITY : (float)sn;
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
                              double gamma, float *__restrict__ win)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m_pad) return;
    if (i >= m) { win[i] = 0.0f; return; }
    const double eps = 2.0 * (ea[i] * bmax + qa[i] * ebmax + gamma * qa[i] * qbmax) * (1.0 + 1e-6) + 1e-12 * (na2[i] + nbmax);
    const double eta = 1.1920928955078125e-07 * (nbmax + 2.0 * qa[i] * qbmax); // 2^-23 (|key| terms)
    const double w = (2.0 * eps + 8.0 * eta) * (1.0 + 1e-6);
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
// LDS: two column tiles of 64 x (DP + 8) halfs; the 16-byte pad makes the fragment read (32 lanes = 32 columns,
// 16 bytes each, pitch = 4 (mod 64) banks... see PITCH below) conflict-free.
template <int KS>
__global__ __launch_bounds__(512, 1) void k_match_half(const _Float16 *__restrict__ ah, int64_t m1,
                                                        const _Float16 *__restrict__ bh, int64_t m2_pad,
                                                        const float *__restrict__ nbf, const float *__restrict__ win,
                                                        float two_s, int *__restrict__ cnt,
                                                        int32_t *__restrict__ cand_j, float *__restrict__ cand_k,
                                                        float *__restrict__ thr_out)
{
    constexpr int DP = 16 * KS;
    // pitch in halfs: (DP + 8) * 2 bytes = 4 * (8 KS + 4) -> 8 KS + 4 dwords, = 52 (mod 64) for KS = 22 and = 4 for
    // KS = 8: 16 consecutive columns x 4 dwords land on 64 distinct banks either way
    constexpr int PITCH = DP + 8;
    constexpr int NCHUNK = HN * DP / 8;               // 16-byte chunks of one tile
    constexpr int NST = (NCHUNK + 511) / 512;         // staging rounds per thread
    __shared__ __attribute__((aligned(16))) _Float16 Bs[2][HN * PITCH];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r31 = lane & 31, h = lane >> 5;
    const int64_t row0 = (int64_t)blockIdx.x * HM + 32 * wave;

    h8 af[KS];
    {
        const _Float16 *ap = ah + (row0 + r31) * DP + 8 * h;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) af[ks] = *reinterpret_cast<const h8 *>(ap + 16 * ks);
    }
    // thresholds of the 16 rows this lane sees in an accumulator: row (r & 3) + 8 (r >> 2) + 4 h
    float thr[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) thr[r] = (row0 + (r & 3) + 8 * (r >> 2) + 4 * h) < m1 ? INFINITY : -INFINITY;

    const int64_t ntiles = m2_pad / HN;
    // staging registers (plain variables and macros: a lambda-captured array is demoted to LDS by the compiler)
    uint4 st0, st1, st2, st3, st4, st5;
    st0 = st1 = st2 = st3 = st4 = st5 = make_uint4(0, 0, 0, 0);
#define SF_H_FETCH1(U, V)                                                                  \
    if ((U) < NST) {                                                                       \
        const int q = tid + 512 * (U);                                                     \
        if (NCHUNK % 512 == 0 || q < NCHUNK) V = src_[q];                                  \
    }
#define SF_H_FETCH(JT)                                                                     \
    {                                                                                      \
        const uint4 *src_ = reinterpret_cast<const uint4 *>(bh + (JT) * HN * DP);          \
        SF_H_FETCH1(0, st0) SF_H_FETCH1(1, st1) SF_H_FETCH1(2, st2) SF_H_FETCH1(3, st3)    \
        SF_H_FETCH1(4, st4) SF_H_FETCH1(5, st5)                                            \
    }
#define SF_H_STASH1(U, V, BUF)                                                             \
    if ((U) < NST) {                                                                       \
        const int q = tid + 512 * (U);                                                     \
        if (NCHUNK % 512 == 0 || q < NCHUNK) {                                             \
            const int col = q / (2 * KS), c16 = q - col * (2 * KS);                        \
            *reinterpret_cast<uint4 *>(&Bs[BUF][col * PITCH + 8 * c16]) = V;               \
        }                                                                                  \
    }
#define SF_H_STASH(BUF)                                                                    \
    {                                                                                      \
        SF_H_STASH1(0, st0, BUF) SF_H_STASH1(1, st1, BUF) SF_H_STASH1(2, st2, BUF)         \
        SF_H_STASH1(3, st3, BUF) SF_H_STASH1(4, st4, BUF) SF_H_STASH1(5, st5, BUF)         \
    }
    static_assert(NST <= 6, "staging registers");
    SF_H_FETCH((int64_t)0)
    SF_H_STASH(0)
    __syncthreads();
    for (int64_t jt = 0; jt < ntiles; ++jt) {
        const int buf = (int)(jt & 1);
        if (jt + 1 < ntiles) SF_H_FETCH(jt + 1)
        f16v acc[2];
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[0][r] = 0.0f; acc[1][r] = 0.0f; }
        const _Float16 *bp = &Bs[buf][r31 * PITCH + 8 * h];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const h8 b0 = *reinterpret_cast<const h8 *>(bp + 16 * ks);
            const h8 b1 = *reinterpret_cast<const h8 *>(bp + 32 * PITCH + 16 * ks);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[ks], b0, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[ks], b1, acc[1], 0, 0, 0);
        }
        if (jt + 1 < ntiles) SF_H_STASH(buf ^ 1)
        // epilogue: key <= thr  <=>  2s acc + thr >= ||b_j||^2 ; the fast test takes the max over the lane's 16 rows
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
            const int64_t j = jt * HN + 32 * cb + r31;
            const float nbv = nbf[j];
            float mx = -INFINITY;
#pragma unroll
            for (int r = 0; r < 16; ++r) mx = fmaxf(mx, fmaf(acc[cb][r], two_s, thr[r]));
            if (__ballot(mx >= nbv)) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float key = fmaf(-two_s, acc[cb][r], nbv);
                    const bool hit = (key <= thr[r]) & (key < INFINITY);
                    if (__ballot(hit)) {
                        float v = hit ? key : INFINITY;
                        v = fminf(v, dpp_f32<0xB1>(v));  // quad_perm [1,0,3,2]
                        v = fminf(v, dpp_f32<0x4E>(v));  // quad_perm [2,3,0,1]
                        v = fminf(v, dpp_f32<0x141>(v)); // row_half_mirror
                        v = fminf(v, dpp_f32<0x140>(v)); // row_mirror: min of the 16-lane row in every lane
                        v = fminf(v, __shfl_xor(v, 16)); // the two DPP rows of this 32-lane half
                        const int64_t row = row0 + (r & 3) + 8 * (r >> 2) + 4 * h;
                        const float tn = fminf(thr[r], v + win[row]);
                        thr[r] = tn;
                        if (hit && key <= tn) {
                            const int s = atomicAdd(&cnt[row], 1);
                            if (s < HCAP) {
                                cand_j[row * HCAP + s] = (int32_t)j;
                                cand_k[row * HCAP + s] = key;
                            }
                        }
                    }
                }
            }
        }
        __syncthreads();
    }
    if (r31 == 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) thr_out[row0 + (r & 3) + 8 * (r >> 2) + 4 * h] = thr[r];
    }
}

// Step 3: the reference's arithmetic on the surviving candidates (one lane per row; scipy's loop order).
__global__ void k_half_final(const double *__restrict__ a, int64_t m1, const double *__restrict__ b, int64_t d,
                             const unsigned char *__restrict__ a_ok, const int *__restrict__ cnt,
                             const int32_t *__restrict__ cand_j, const float *__restrict__ cand_k,
                             const float *__restrict__ thr, int64_t *__restrict__ idx, double *__restrict__ dist,
                             int *__restrict__ flag, int *__restrict__ n_flagged)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m1) return;
    if (a_ok && !a_ok[i]) { // masked scan row: +inf from everything, first column (shotfpfh.h, sf_match_argmin_multiscale)
        idx[i] = 0;
        if (dist) dist[i] = INFINITY;
        flag[i] = 0;
        return;
    }
    const int n = cnt[i];
    const float t = thr[i];
    double best = INFINITY;
    int64_t bj = -1;
    if (n <= HCAP) {
        const double *ai = a + i * d;
        for (int c = 0; c < n; ++c) {
            if (!(cand_k[i * HCAP + c] <= t)) continue;
            const int64_t j = cand_j[i * HCAP + c];
            const double *bjp = b + j * d;
            double acc = 0.0;
            for (int64_t u = 0; u < d; ++u) {
                const double df = ai[u] - bjp[u];
                acc += df * df; // left to right, no FMA: scipy's euclidean loop
            }
            const double dj = sqrt(acc);
            if (dj < best || (dj == best && j < bj) || bj < 0) {
                if (!(dj == dj)) continue; // NaN: leave the row to the float64 path
                best = dj;
                bj = j;
            }
        }
    }
    const bool decided = bj >= 0;
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
    std::vector<void *> held;
    auto release = [&]() { for (void *p : held) sf_pool_release(ctx, p); held.clear(); };
#define SF_HALLOC(ptr, count)                                   \
    {                                                           \
        int rc_ = sf_palloc(ctx, &ptr, (size_t)(count));        \
        if (rc_ != SF_OK) { release(); return rc_; }            \
        held.push_back(ptr);                                    \
    }
    SF_HALLOC(part, 256);
    SF_HALLOC(nb2, m2p); SF_HALLOC(eb, m2p); SF_HALLOC(qb, m2p); SF_HALLOC(nbf, m2p);
    SF_HALLOC(na2, m1p); SF_HALLOC(ea, m1p); SF_HALLOC(qa, m1p);
    SF_HALLOC(ah, m1p * dp); SF_HALLOC(bh, m2p * dp);
    // scales from the largest row norms (first conversion pass with scale 1 only for the norms would cost another
    // read of both matrices; the norms are cheap on their own)
    double namax = 0.0, nbmax = 0.0, sa = 1.0, sb = 1.0;
    {
        // ||row||^2 through the converter with scale 1 into the same buffers (overwritten below)
        SF_LAUNCH(ctx, "k8_half_convert", k_half_convert, dim3((unsigned)sf_div_up(m1p, 4)), dim3(256), da, m1, m1p, d, dp,
                  1.0, (const unsigned char *)nullptr, ah, ea, qa, na2, (float *)nullptr);
        SF_LAUNCH(ctx, "k8_half_convert", k_half_convert, dim3((unsigned)sf_div_up(m2p, 4)), dim3(256), db, m2, m2p, d, dp,
                  1.0, b_ok, bh, eb, qb, nb2, nbf);
        int rc = host_max(ctx, na2, m1, part, &namax);
        if (rc == SF_OK) rc = host_max(ctx, nb2, m2, part, &nbmax);
        if (rc != SF_OK) { release(); return rc; }
    }
    if (!pick_scale(namax, &sa) || !pick_scale(nbmax, &sb) || !(nbmax < 1e37)) { release(); return SF_OK; }
    SF_LAUNCH(ctx, "k8_half_convert", k_half_convert, dim3((unsigned)sf_div_up(m1p, 4)), dim3(256), da, m1, m1p, d, dp, sa,
              (const unsigned char *)nullptr, ah, ea, qa, na2, (float *)nullptr);
    SF_LAUNCH(ctx, "k8_half_convert", k_half_convert, dim3((unsigned)sf_div_up(m2p, 4)), dim3(256), db, m2, m2p, d, dp, sb,
              b_ok, bh, eb, qb, nb2, nbf);
    double ebmax = 0.0, qbmax = 0.0;
    {
        int rc = host_max(ctx, eb, m2, part, &ebmax);
        if (rc == SF_OK) rc = host_max(ctx, qb, m2, part, &qbmax);
        if (rc != SF_OK) { release(); return rc; }
    }
    if (!std::isfinite(ebmax) || !std::isfinite(qbmax)) { release(); return SF_OK; }
    SF_HALLOC(win, m1p); SF_HALLOC(thr, m1p); SF_HALLOC(cnt, m1p);
    SF_HALLOC(candj, m1p * HCAP); SF_HALLOC(candk, m1p * HCAP);
    SF_HALLOC(flag, m1); SF_HALLOC(nflag, 1);
    const double gamma = (double)(dp + 32) * 2.384185791015625e-07; // 2^-22
    SF_LAUNCH(ctx, "k8_half_window", k_half_window, dim3((unsigned)sf_div_up(m1p, 256)), dim3(256), (const double *)ea,
              (const double *)qa, (const double *)na2, m1, m1p, std::sqrt(nbmax), ebmax, qbmax, nbmax, gamma, win);
    SF_HIP(hipMemsetAsync(cnt, 0, (size_t)m1p * sizeof(int), ctx->stream));
    SF_HIP(hipMemsetAsync(nflag, 0, sizeof(int), ctx->stream));
    const float two_s = (float)(2.0 / (sa * sb)); // a power of two
    if (!(two_s > 0.0f) || !std::isfinite(two_s)) { release(); return SF_OK; }
    if (ks == 8) {
        SF_LAUNCH(ctx, name, k_match_half<8>, dim3((unsigned)(m1p / HM)), dim3(512), (const _Float16 *)ah, m1,
                  (const _Float16 *)bh, m2p, (const float *)nbf, (const float *)win, two_s, cnt, candj, candk, thr);
    } else {
        SF_LAUNCH(ctx, name, k_match_half<22>, dim3((unsigned)(m1p / HM)), dim3(512), (const _Float16 *)ah, m1,
                  (const _Float16 *)bh, m2p, (const float *)nbf, (const float *)win, two_s, cnt, candj, candk, thr);
    }
    SF_LAUNCH(ctx, "k8_half_final", k_half_final, dim3((unsigned)sf_div_up(m1, 256)), dim3(256), da, m1, db, d, a_ok,
              (const int *)cnt, (const int32_t *)candj, (const float *)candk, (const float *)thr, didx, ddist, flag, nflag);
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
    release();
#undef SF_HALLOC
    *used = 1;
    return rc;
}
