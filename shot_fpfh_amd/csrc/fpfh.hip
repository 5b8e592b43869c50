// fpfh.hip -- K6 (SPFH integer histograms for every cloud point) and K7 (FPFH weighted reduction).
//
// Replaces: compute_fpfh_descriptor, fpfh.py:16-117 (decorrelated=False):
//   K6  fpfh.py:38-90   per point i, per neighbour j with d > 0:  u = n_i, v = (p_j-p_i) x u (NOT
//       normalised), w = u x v, alpha = v.n_j, phi = (p_j-p_i).u / d, theta = atan2(n_j.w, n_j.u);
//       np.histogramdd over (-1,1) x (-1,1) x (-pi/2,pi/2) with np.linspace edges -- samples outside
//       any range are DROPPED while the normaliser stays k = len(neighbourhood), self included.
//   K7  fpfh.py:101-116 fpfh[kp] = spfh[kp] + (sum_{j in nbrs(kp), d_j > 0} spfh[j] / d_j) / k_kp.
// Data layout in HBM: the SPFH table is kept as INTEGER bin counts (uint16, or uint32 when a
// neighbourhood exceeds 65535) plus the per-point k, by cell-sorted position: spfh[j][b] is
// reconstructed as (double)count/k exactly as the reference computed it, but a row costs 256 B instead
// of 1000 B in the K7 gather, which is that kernel's dominant traffic (k x row per keypoint).
// Mapping: one wave per point.  K6 bins with per-wave LDS atomics; K7 puts two bins on each lane and
// streams the neighbour rows (one coalesced 256-B load per neighbour).
// HBM roofline, algorithmic bytes (float64 API widths): 48 in + 1000 SPFH write + 1000 SPFH
// read + 1000 FPFH write = 3048 B per descriptor when every point is a keypoint.
#include <algorithm>

#include "common.h"
#include "device_util.h"

namespace {

struct fpfh_edges {
    double a[SF_MAX_FPFH_BINS + 1], p[SF_MAX_FPFH_BINS + 1], t[SF_MAX_FPFH_BINS + 1];
};

// np.histogramdd bin of x: searchsorted(edges, x, 'right') - 1, x == last edge -> last bin, out of
// range / NaN -> -1 (dropped).
__device__ inline int hist_bin(const double *e, int nb, double x)
{
    if (!(x >= e[0]) || x > e[nb]) return -1;
    int b = 0;
#pragma unroll
    for (int i = 1; i < SF_MAX_FPFH_BINS; ++i)
        if (i < nb && x >= e[i]) b = i;
    return b;
}

template <typename CT>
__global__ __launch_bounds__(256) void k_spfh(const double *__restrict__ xs, const double *__restrict__ ys,
                                              const double *__restrict__ zs, const double *__restrict__ nxs,
                                              const double *__restrict__ nys, const double *__restrict__ nzs,
                                              const int64_t *__restrict__ offset, const int32_t *__restrict__ idx,
                                              int64_t m, int64_t self_begin, fpfh_edges ed, int nb, int nb3, int stride,
                                              CT *__restrict__ counts, int32_t *__restrict__ kout)
{
    __shared__ unsigned int hist[4][SF_MAX_FPFH_BINS * SF_MAX_FPFH_BINS * SF_MAX_FPFH_BINS];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t q = sf_uniform64(sf_xcd_block() * 4 + wave);
    if (q >= m) return; // whole wave exits together; no block-wide barrier below
    unsigned int *h = hist[wave];
    for (int b = lane; b < nb3; b += 64) h[b] = 0;
    const int64_t i = self_begin + q; // cell-sorted position of this point
    const int64_t s = offset[q];
    const int k = (int)(offset[q + 1] - s);
    const double px = xs[i], py = ys[i], pz = zs[i];
    const double ux = nxs[i], uy = nys[i], uz = nzs[i];
    __builtin_amdgcn_wave_barrier();
    for (int t = lane; t < k; t += 64) {
        const int j = idx[s + t];
        const double cx = xs[j] - px, cy = ys[j] - py, cz = zs[j] - pz;
        const double d2 = (cx * cx + cy * cy) + cz * cz;
        if (d2 > 0.0) { // dist > 0 (fpfh.py:50-57)
            const double dist = sqrt(d2);
            const double njx = nxs[j], njy = nys[j], njz = nzs[j];
            const double vx = cy * uz - cz * uy, vy = cz * ux - cx * uz, vz = cx * uy - cy * ux; // cross(c, u)
            const double wx = uy * vz - uz * vy, wy = uz * vx - ux * vz, wz = ux * vy - uy * vx; // cross(u, v)
            const double alpha = (vx * njx + vy * njy) + vz * njz;
            const double phi = ((cx * ux + cy * uy) + cz * uz) / dist;
            const double theta = atan2((njx * wx + njy * wy) + njz * wz, (njx * ux + njy * uy) + njz * uz);
            const int ba = hist_bin(ed.a, nb, alpha), bp = hist_bin(ed.p, nb, phi), bt = hist_bin(ed.t, nb, theta);
            if ((ba | bp | bt) >= 0) atomicAdd(&h[(ba * nb + bp) * nb + bt], 1u);
        }
    }
    __builtin_amdgcn_wave_barrier();
    __threadfence_block();
    CT *row = counts + i * (int64_t)stride;
    for (int b = lane; b < stride; b += 64) row[b] = b < nb3 ? (CT)h[b] : (CT)0;
    if (lane == 0) kout[i] = k;
}

// K7.  NB2 = number of (bin pair)s per lane: lane l owns bins 2*(l + 64*u) and +1, u < NB2.
template <typename CT, int NB2>
__global__ __launch_bounds__(256) void k_fpfh(const double *__restrict__ xs, const double *__restrict__ ys,
                                              const double *__restrict__ zs, const int64_t *__restrict__ offset,
                                              const int32_t *__restrict__ idx, int64_t nbrs_begin,
                                              const int32_t *__restrict__ kp_pos, int64_t m, int nb3, int stride,
                                              const CT *__restrict__ counts, const int32_t *__restrict__ kk,
                                              double *__restrict__ out)
{
    const int lane = threadIdx.x & 63;
    const int64_t q = sf_uniform64(sf_xcd_block() * 4 + (threadIdx.x >> 6));
    if (q >= m) return;
    // keypoint's cell-sorted position and its slot in the neighbour lists
    const int64_t i = kp_pos ? (int64_t)kp_pos[q] : nbrs_begin + q;
    const int64_t slot = i - nbrs_begin;
    const int64_t s = offset[slot];
    const int k = (int)(offset[slot + 1] - s);
    const double px = xs[i], py = ys[i], pz = zs[i];
    double acc0[NB2], acc1[NB2];
#pragma unroll
    for (int u = 0; u < NB2; ++u) { acc0[u] = 0.0; acc1[u] = 0.0; }
    for (int t0 = 0; t0 < k; t0 += 64) {
        const int t = t0 + lane;
        int j = 0;
        double w = 0.0;
        if (t < k) {
            j = idx[s + t];
            const double cx = xs[j] - px, cy = ys[j] - py, cz = zs[j] - pz;
            const double d2 = (cx * cx + cy * cy) + cz * cz;
            // weight of neighbour j: spfh[j] / d_j with spfh[j] = count_j / k_j ; d == 0 is masked out
            if (d2 > 0.0) w = (1.0 / (double)kk[j]) / sqrt(d2);
        }
        const int cnt = min(64, k - t0);
        // tt is wave-uniform: v_readlane hands the neighbour index / weight over as scalars, the row base is
        // a scalar address, and rows are fetched in batches of 8 independent loads before any is consumed
        auto row_word = [&](int tt, int u) -> uint2 {
            // unconditional load (lanes past the row end re-read its first word and are masked by select):
            // a branch around the load would make hipcc wait for each row before issuing the next
            const unsigned jj = (unsigned)__builtin_amdgcn_readlane(j, tt);
            const char *rowp = reinterpret_cast<const char *>(counts) + (size_t)jj * ((size_t)stride * sizeof(CT));
            const int b = 2 * (lane + 64 * u);
            const bool ok = b < stride; // stride is even and >= nb3; padding counts are zero
            const int bb = ok ? b : 0;
            uint2 pk;
            if (sizeof(CT) == 2) {
                const unsigned int v = *reinterpret_cast<const unsigned int *>(rowp + (size_t)bb * 2);
                pk.x = v & 0xffffu;
                pk.y = v >> 16;
            } else {
                pk = *reinterpret_cast<const uint2 *>(rowp + (size_t)bb * 4);
            }
            pk.x = ok ? pk.x : 0u;
            pk.y = ok ? pk.y : 0u;
            return pk;
        };
        auto weight = [&](int tt) -> double {
            return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(w), tt),
                                    __builtin_amdgcn_readlane(__double2loint(w), tt));
        };
        int tt = 0;
        for (; tt + 8 <= cnt; tt += 8) {
            uint2 pk[8][NB2];
#pragma unroll
            for (int e = 0; e < 8; ++e)
#pragma unroll
                for (int u = 0; u < NB2; ++u) pk[e][u] = row_word(tt + e, u);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const double ww = weight(tt + e);
#pragma unroll
                for (int u = 0; u < NB2; ++u) {
                    acc0[u] = __builtin_fma((double)pk[e][u].x, ww, acc0[u]);
                    acc1[u] = __builtin_fma((double)pk[e][u].y, ww, acc1[u]);
                }
            }
        }
        for (; tt < cnt; ++tt) {
            const double ww = weight(tt);
#pragma unroll
            for (int u = 0; u < NB2; ++u) {
                const uint2 pk = row_word(tt, u);
                acc0[u] = __builtin_fma((double)pk.x, ww, acc0[u]);
                acc1[u] = __builtin_fma((double)pk.y, ww, acc1[u]);
            }
        }
    }
    const double kd = (double)k;
    const CT *own = counts + i * (int64_t)stride;
    double *o = out + q * (int64_t)nb3;
#pragma unroll
    for (int u = 0; u < NB2; ++u) {
        const int b = 2 * (lane + 64 * u);
        if (b < nb3) o[b] = (double)own[b] / kd + acc0[u] / kd;
        if (b + 1 < nb3) o[b + 1] = (double)own[b + 1] / kd + acc1[u] / kd;
    }
}

template <typename CT>
__global__ void k_spfh_export(const CT *__restrict__ counts, const int32_t *__restrict__ kk,
                              const int32_t *__restrict__ perm, int64_t n, int nb3, int stride, double *__restrict__ out)
{
    int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n * nb3) return;
    int64_t i = g / nb3;
    int b = (int)(g - i * nb3);
    out[(int64_t)perm[i] * nb3 + b] = (double)counts[i * stride + b] / (double)kk[i];
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

extern "C" sf_spfh *sf_spfh_create(sf_ctx *ctx, sf_cloud *c, int n_bins, int64_t max_count)
{
    if (!ctx || !c) { sf_set_error("sf_spfh_create: null argument"); return nullptr; }
    if (n_bins < 1 || n_bins > SF_MAX_FPFH_BINS) {
        sf_set_error("sf_spfh_create: n_bins=%d unsupported on device (1..%d)", n_bins, SF_MAX_FPFH_BINS);
        return nullptr;
    }
    if (hipSetDevice(ctx->device) != hipSuccess) { sf_set_error("hipSetDevice failed"); return nullptr; }
    sf_spfh *sp = new sf_spfh();
    sp->n = c->n;
    sp->n_bins = n_bins;
    sp->nb3 = n_bins * n_bins * n_bins;
    sp->elem_bytes = max_count > 65535 ? 4 : 2;
    int per128 = 128 / sp->elem_bytes; // rows padded to a multiple of 128 B
    sp->stride = (int)sf_div_up(sp->nb3, per128) * per128;
    // room for ceil(n / nranks) rows per rank so the table can be all-gathered in place
    const int64_t nr = ctx->nranks > 0 ? ctx->nranks : 1;
    sp->rows_alloc = std::max<int64_t>(sf_div_up(c->n, nr) * nr, 1);
    size_t nn = (size_t)sp->rows_alloc;
    if (hipMalloc(&sp->counts, nn * sp->stride * sp->elem_bytes) != hipSuccess ||
        hipMalloc(&sp->k, nn * sizeof(int32_t)) != hipSuccess) {
        sf_set_error("sf_spfh_create: out of device memory");
        sf_spfh_free(ctx, sp);
        return nullptr;
    }
    return sp;
}

extern "C" void sf_spfh_free(sf_ctx *ctx, sf_spfh *sp)
{
    if (!sp) return;
    if (ctx) (void)hipStreamSynchronize(ctx->stream);
    if (sp->counts) (void)hipFree(sp->counts);
    if (sp->k) (void)hipFree(sp->k);
    delete sp;
}

extern "C" int sf_spfh_compute(sf_ctx *ctx, sf_cloud *c, sf_nbrs *nb, sf_spfh *sp, const double *edges)
{
    if (!ctx || !c || !nb || !sp || !edges) { sf_set_error("sf_spfh_compute: null argument"); return SF_ERR_ARG; }
    if (!nb->self) { sf_set_error("sf_spfh_compute: needs a sf_radius_search_self result"); return SF_ERR_ARG; }
    if (sp->n != c->n || nb->self_begin + nb->m > sp->n) { sf_set_error("sf_spfh_compute: table/cloud size mismatch"); return SF_ERR_ARG; }
    if (sp->elem_bytes == 2 && nb->max_count > 65535) {
        sf_set_error("sf_spfh_compute: neighbourhood of %lld points needs a 32-bit table (pass max_count to sf_spfh_create)",
                     (long long)nb->max_count);
        return SF_ERR_ARG;
    }
    SF_HIP(hipSetDevice(ctx->device));
    SF_CHECK(sf_cloud_ensure_sorted_normals(ctx, c));
    fpfh_edges ed;
    const int nbn = sp->n_bins;
    for (int i = 0; i <= SF_MAX_FPFH_BINS; ++i) {
        ed.a[i] = edges[i <= nbn ? i : nbn];
        ed.p[i] = edges[(nbn + 1) + (i <= nbn ? i : nbn)];
        ed.t[i] = edges[2 * (nbn + 1) + (i <= nbn ? i : nbn)];
    }
    const int64_t m = nb->m;
    if (!m) return SF_OK;
    const dim3 grid(sf_xcd_grid(sf_div_up(m, 4))), block(256);
    if (sp->elem_bytes == 2) {
        SF_LAUNCH(ctx, "k6_spfh", k_spfh<uint16_t>, grid, block, c->xs, c->ys, c->zs, c->nxs, c->nys, c->nzs, nb->offset,
                  nb->idx, m, nb->self_begin, ed, nbn, sp->nb3, sp->stride, (uint16_t *)sp->counts, sp->k);
    } else {
        SF_LAUNCH(ctx, "k6_spfh", k_spfh<uint32_t>, grid, block, c->xs, c->ys, c->zs, c->nxs, c->nys, c->nzs, nb->offset,
                  nb->idx, m, nb->self_begin, ed, nbn, sp->nb3, sp->stride, (uint32_t *)sp->counts, sp->k);
    }
    return SF_OK;
}

extern "C" int sf_spfh_allgather(sf_ctx *ctx, sf_spfh *sp, int64_t rows_per_rank)
{
    if (!ctx || !sp) { sf_set_error("sf_spfh_allgather: null argument"); return SF_ERR_ARG; }
    if (ctx->nranks == 1) return SF_OK;
    if (rows_per_rank <= 0 || rows_per_rank * ctx->nranks > sp->rows_alloc || rows_per_rank * ctx->nranks < sp->n) {
        sf_set_error("sf_spfh_allgather: %lld rows/rank x %d ranks does not tile a table of %lld (+pad %lld) rows",
                     (long long)rows_per_rank, ctx->nranks, (long long)sp->n, (long long)sp->rows_alloc);
        return SF_ERR_ARG;
    }
    const size_t row_bytes = (size_t)sp->stride * sp->elem_bytes;
    char *base = (char *)sp->counts;
    SF_CHECK(sf_comm_allgather(ctx, base + (size_t)ctx->rank * rows_per_rank * row_bytes, base,
                               (size_t)rows_per_rank * row_bytes));
    char *kb = (char *)sp->k;
    SF_CHECK(sf_comm_allgather(ctx, kb + (size_t)ctx->rank * rows_per_rank * sizeof(int32_t), kb,
                               (size_t)rows_per_rank * sizeof(int32_t)));
    return SF_OK;
}

extern "C" int sf_spfh_export(sf_ctx *ctx, sf_cloud *c, sf_spfh *sp, double *out, int flags)
{
    if (!ctx || !c || !sp || !out) { sf_set_error("sf_spfh_export: null argument"); return SF_ERR_ARG; }
    SF_HIP(hipSetDevice(ctx->device));
    const int64_t n = sp->n, tot = n * sp->nb3;
    double *dout = out, *owned = nullptr;
    if (!(flags & SF_OUT_DEVICE)) {
        SF_HIP(hipMalloc(&owned, (size_t)(tot ? tot : 1) * sizeof(double)));
        dout = owned;
    }
    if (tot) {
        const dim3 grid((unsigned)sf_div_up(tot, 256)), block(256);
        if (sp->elem_bytes == 2) {
            SF_LAUNCH(ctx, "k6_spfh_export", k_spfh_export<uint16_t>, grid, block, (const uint16_t *)sp->counts, sp->k,
                      c->perm, n, sp->nb3, sp->stride, dout);
        } else {
            SF_LAUNCH(ctx, "k6_spfh_export", k_spfh_export<uint32_t>, grid, block, (const uint32_t *)sp->counts, sp->k,
                      c->perm, n, sp->nb3, sp->stride, dout);
        }
    }
    if (owned) {
        if (tot) SF_HIP(hipMemcpyAsync(out, owned, (size_t)tot * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        SF_HIP(hipStreamSynchronize(ctx->stream));
        SF_HIP(hipFree(owned));
    }
    return SF_OK;
}

template <typename CT>
static int launch_fpfh(sf_ctx *ctx, sf_cloud *c, sf_nbrs *nb, sf_spfh *sp, const int32_t *kp_pos, int64_t m,
                       double *dout)
{
    const dim3 grid(sf_xcd_grid(sf_div_up(m, 4))), block(256);
    const int pairs = sp->stride / 2;                 // bin pairs per row
    const int nb2 = (int)sf_div_up(pairs, 64);        // pairs per lane
#define SF_FPFH_CASE(NB2)                                                                                           \
    case NB2: {                                                                                                     \
        SF_LAUNCH(ctx, "k7_fpfh", (k_fpfh<CT, NB2>), grid, block, c->xs, c->ys, c->zs, nb->offset, nb->idx,          \
                  nb->self_begin, kp_pos, m, sp->nb3, sp->stride, (const CT *)sp->counts, sp->k, dout);             \
    } break;
    switch (nb2) {
        SF_FPFH_CASE(1)
        SF_FPFH_CASE(2)
        SF_FPFH_CASE(3)
        SF_FPFH_CASE(4)
    default:
        sf_set_error("sf_fpfh: %d histogram cells per point unsupported", sp->nb3);
        return SF_ERR_UNSUPPORTED;
    }
#undef SF_FPFH_CASE
    return SF_OK;
}

extern "C" int sf_fpfh(sf_ctx *ctx, sf_cloud *c, sf_nbrs *nb, sf_spfh *sp, const int64_t *kp_idx, int64_t m, double *out,
                       int flags)
{
    if (!ctx || !c || !nb || !sp || !out || m < 0) { sf_set_error("sf_fpfh: bad argument"); return SF_ERR_ARG; }
    if (!nb->self) { sf_set_error("sf_fpfh: needs a sf_radius_search_self result"); return SF_ERR_ARG; }
    if (!kp_idx && m != nb->m) { sf_set_error("sf_fpfh: m must equal the query count when kp_idx is NULL"); return SF_ERR_ARG; }
    if (kp_idx && !(nb->self_begin == 0 && nb->m == c->n)) {
        sf_set_error("sf_fpfh: keypoints by index need neighbour lists of the whole cloud");
        return SF_ERR_ARG;
    }
    SF_HIP(hipSetDevice(ctx->device));
    int32_t *pos = nullptr;
    int64_t *dkp = nullptr;
    int *dbad = nullptr;
    if (kp_idx && m) {
        const int64_t *src = kp_idx;
        if (!(flags & SF_IN_DEVICE)) {
            SF_HIP(hipMalloc(&dkp, (size_t)m * sizeof(int64_t)));
            SF_HIP(hipMemcpyAsync(dkp, kp_idx, (size_t)m * sizeof(int64_t), hipMemcpyHostToDevice, ctx->stream));
            src = dkp;
        }
        SF_HIP(hipMalloc(&pos, (size_t)m * sizeof(int32_t)));
        SF_HIP(hipMalloc(&dbad, sizeof(int)));
        SF_HIP(hipMemsetAsync(dbad, 0, sizeof(int), ctx->stream));
        SF_LAUNCH(ctx, "k7_map_positions", k_map_positions, dim3((unsigned)sf_div_up(m, 256)), dim3(256), src,
                  c->inv_perm, m, c->n, pos, dbad);
        int bad = 0;
        SF_HIP(hipMemcpyAsync(&bad, dbad, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
        SF_HIP(hipStreamSynchronize(ctx->stream));
        SF_HIP(hipFree(dbad));
        if (dkp) SF_HIP(hipFree(dkp));
        if (bad) {
            SF_HIP(hipFree(pos));
            sf_set_error("sf_fpfh: keypoint index out of range for a cloud of %lld points", (long long)c->n);
            return SF_ERR_ARG;
        }
    }
    const int64_t tot = m * sp->nb3;
    double *dout = out, *owned = nullptr;
    if (!(flags & SF_OUT_DEVICE)) {
        SF_HIP(hipMalloc(&owned, (size_t)(tot ? tot : 1) * sizeof(double)));
        dout = owned;
    }
    int rc = SF_OK;
    if (m) rc = sp->elem_bytes == 2 ? launch_fpfh<uint16_t>(ctx, c, nb, sp, pos, m, dout)
                                    : launch_fpfh<uint32_t>(ctx, c, nb, sp, pos, m, dout);
    if (rc == SF_OK && owned) {
        if (tot) SF_HIP(hipMemcpyAsync(out, owned, (size_t)tot * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    }
    if (owned || pos) SF_HIP(hipStreamSynchronize(ctx->stream));
    if (owned) SF_HIP(hipFree(owned));
    if (pos) SF_HIP(hipFree(pos));
    return rc;
}
