// common.h -- shared host-side plumbing of libshotfpfh.so (context, errors, launch + timing).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <atomic>
#include <tuple>
#include <string>
#include <vector>

#include "../../include/shotfpfh.h"

// Every host wait on a stream goes through this counter (sf_sync_count): a step that is to be captured into a HIP graph must not
// wait for the device, and whether it does is checked on an eager run of the same step first (Engine.capture's callers).
extern std::atomic<unsigned long long> g_sf_sync_count; // context.hip
static inline hipError_t sf_counted_stream_sync(hipStream_t s)
{
    g_sf_sync_count.fetch_add(1, std::memory_order_relaxed);
    return hipStreamSynchronize(s);
}
#define hipStreamSynchronize(s) sf_counted_stream_sync(s)

#define SF_WAVE 64

void sf_set_error(const char *fmt, ...);

#define SF_HIP(call)                                                                                  \
    do {                                                                                              \
        hipError_t e_ = (call);                                                                       \
        if (e_ != hipSuccess) {                                                                       \
            sf_set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
            return SF_ERR_HIP;                                                                        \
        }                                                                                             \
    } while (0)

#define SF_HIP_NULL(call)                                                                             \
    do {                                                                                              \
        hipError_t e_ = (call);                                                                       \
        if (e_ != hipSuccess) {                                                                       \
            sf_set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
            return nullptr;                                                                           \
        }                                                                                             \
    } while (0)

#define SF_CHECK(expr)      \
    do {                    \
        int rc_ = (expr);   \
        if (rc_ != SF_OK) return rc_; \
    } while (0)

struct sf_prof_entry {
    int64_t launches = 0;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
    double total_ms = 0.0;
};

// A captured step (sf_graph_begin / sf_graph_end): the HIP graph of everything issued between the two calls, and the pool blocks
// released meanwhile -- a replay writes into them again, so they stay out of the pool for as long as the graph lives.
struct sf_graph {
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    std::vector<void *> held;
};

struct sf_ctx {
    int device = 0;
    hipStream_t stream = nullptr;   // the stream calls are issued on (one of streams[])
    hipStream_t streams[2] = {nullptr, nullptr};
    hipEvent_t join_event = nullptr;
    hipEvent_t mark_event = nullptr; // sf_mark / sf_wait_mark
    bool profiling = false;
    std::string prof_only; // sf_profile_only: time launches of this name only ("" = all)
    std::map<std::string, sf_prof_entry> prof;
    std::vector<hipEvent_t> event_pool;
    void *comm = nullptr; // ncclComm_t
    int nranks = 1, rank = 0;
    bool collective_stats = false; // sf_comm_collective_stats: radius searches all-reduce their list statistics
    // small reusable device scratch (bbox partials etc.)
    void *scratch = nullptr;
    size_t scratch_bytes = 0;
    // page-locked host words for the few scalars a step reads back (bounding box, list statistics): a copy into
    // pageable memory is staged and costs tens of microseconds of idle GPU each time
    void *pinned = nullptr; // SF_PINNED_BYTES
    double *shot_coef = nullptr; // K5's polynomial coefficients in device memory (descriptors.hip::shot_coef_table)
    // one page-locked, device-visible word that kernels set when an index array handed in by the caller (a row selection,
    // a visiting order) holds a value out of range -- such an element is skipped, never dereferenced.  The host looks at
    // the word after every synchronisation it makes anyway (sf_ctx_check_flag) and turns it into SF_ERR_ARG.
    volatile int *dev_flag = nullptr;
    // stream-ordered caching allocator: freed blocks are reused by later launches on the SAME stream,
    // so neither hipMalloc nor the implicit device sync of hipFree sits inside a step of the path
    std::multimap<size_t, void *> pool_free;
    std::map<void *, size_t> pool_size;
    size_t pool_cached = 0;
    sf_graph *capture = nullptr; // non-null between sf_graph_begin and sf_graph_end
};

int sf_pool_alloc(sf_ctx *ctx, size_t bytes, void **out);
void sf_pool_release(sf_ctx *ctx, void *p);
void sf_pool_trim(sf_ctx *ctx);
template <typename T>
static inline int sf_palloc(sf_ctx *ctx, T **out, size_t count)
{
    void *p = nullptr;
    int rc = sf_pool_alloc(ctx, (count ? count : 1) * sizeof(T), &p);
    *out = (T *)p;
    return rc;
}

// Temporaries of one entry point: every block goes back to the pool when the guard leaves scope, on the error
// returns of SF_HIP / SF_CHECK / SF_LAUNCH as well as on success.  Releasing is stream-ordered (the block is only
// handed to later launches on the same stream), so it needs no synchronisation.
struct sf_pool_guard {
    sf_ctx *ctx;
    std::vector<void *> held;
    explicit sf_pool_guard(sf_ctx *c) : ctx(c) {}
    sf_pool_guard(const sf_pool_guard &) = delete;
    sf_pool_guard &operator=(const sf_pool_guard &) = delete;
    template <typename T>
    int alloc(T **out, size_t count)
    {
        const int rc = sf_palloc(ctx, out, count);
        if (rc == SF_OK) held.push_back((void *)*out);
        return rc;
    }
    void release(void *p) // early release of one block
    {
        for (auto &h : held)
            if (h == p) { sf_pool_release(ctx, p); h = nullptr; }
    }
    ~sf_pool_guard()
    {
        for (void *p : held)
            if (p) sf_pool_release(ctx, p);
    }
};

int sf_comm_allreduce_max_i32(sf_ctx *ctx, const int *send, int *recv, size_t n); // comm.hip
int sf_ctx_scratch(sf_ctx *ctx, size_t bytes, void **out);
#define SF_FLAG_ROWS_GATHER 1
#define SF_FLAG_VOXEL_ORDER 2
int sf_ctx_check_flag(sf_ctx *ctx); // after a stream synchronisation: SF_OK, or SF_ERR_ARG with the message of the raised bit(s)
#define SF_PINNED_BYTES 32768
int sf_ctx_pinned(sf_ctx *ctx, void **out);
hipEvent_t sf_ctx_event(sf_ctx *ctx);

// RAII-ish timing scope around one kernel launch (active only when profiling is on).
struct sf_launch_timer {
    sf_ctx *ctx;
    const char *name;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    bool active;
    sf_launch_timer(sf_ctx *c, const char *n)
        : ctx(c), name(n), active(c->profiling && (c->prof_only.empty() || c->prof_only == n))
    {
        if (active) {
            e0 = sf_ctx_event(ctx);
            e1 = sf_ctx_event(ctx);
            (void)hipEventRecord(e0, ctx->stream);
        }
    }
    ~sf_launch_timer()
    {
        if (active) {
            (void)hipEventRecord(e1, ctx->stream);
            sf_prof_entry &p = ctx->prof[name];
            p.launches++;
            p.pending.emplace_back(e0, e1);
        } else {
            ctx->prof[name].launches++;
        }
    }
};

// Launch `kernel<<<grid, block, 0, ctx->stream>>>(args...)` under a named timer.
#define SF_LAUNCH(ctx, name, kernel, grid, block, ...)                                   \
    do {                                                                                 \
        sf_launch_timer t_((ctx), (name));                                               \
        hipLaunchKernelGGL(kernel, (grid), (block), 0, (ctx)->stream, __VA_ARGS__);      \
    } while (0);                                                                         \
    SF_HIP(hipGetLastError())

struct sf_cloud {
    int64_t n = 0;
    // The uploaded points, kept SORTED BY z (stable: ties in ascending caller index) -- done once per upload, whatever radius
    // is searched later: every z-layer of any grid is then a contiguous run of this "internal" order, so a rank of a sharded
    // job reads only its own slab when it builds its block of the grid (until round 4 it looked at the whole replicated
    // cloud twice per build), and the gather of a grid build reads a layer's points from one stretch of the array.  Sorting
    // cell ids is stable on THIS order for every build, whole or block, so cell-sorted positions -- hence every result bit --
    // do not depend on how many ranks share the cloud.
    double *xyz_orig = nullptr;     // n x 3 AoS, internal (z-sorted) order
    double *nrm_orig = nullptr;     // n x 3 AoS or null, internal order
    int32_t *zperm = nullptr;       // internal index -> the caller's point index
    int32_t *perm_int = nullptr;    // cell-sorted position -> internal index (what gathers from xyz_orig / nrm_orig use)
    // bounding box of the uploaded points: a property of the (immutable) cloud, computed by the first grid build and
    // kept -- every later build (another radius, a k-NN retry, the next pass over a resident cloud) skips the
    // reduction kernels and the device-to-host read-back of their six numbers
    bool bbox_known = false;
    double bbox_lo[3] = {0, 0, 0}, bbox_hi[3] = {0, 0, 0};
    // grid
    uint64_t grid_gen = 0;          // counts the builds of the grid (every build re-allocates the sorted arrays): sf_nbrs::grid_gen
    double cell = 0.0;              // actual cell edge used
    double inv_cell = 0.0;
    double lo[3] = {0, 0, 0};
    int dim[3] = {1, 1, 1};         // cells per axis; dim[0] counts the FINE cells along x (xsub per grid edge)
    int xsub = 1;                   // x is subdivided xsub times finer than y / z (see grid.hip)
    int64_t ncell = 0;
    int32_t *cell_start = nullptr;  // ncell + 1
    int32_t *perm = nullptr;        // sorted position -> original index
    int32_t *inv_perm = nullptr;    // original index -> sorted position (filled on demand: sf_cloud_ensure_inv_perm)
    bool inv_perm_valid = false;
    // cell-sorted SoA
    double *xs = nullptr, *ys = nullptr, *zs = nullptr;
    // cell-sorted AoS records {x, y, z, nx, ny, nz} (48 B, 16-byte aligned): what the list-driven kernels
    // gather -- three 16-byte loads bring in a neighbour's position AND normal (the vector-memory pipe
    // costs 16 cycles per wave instruction whatever the width, so 3 wide loads beat 6 narrow ones)
    double *rec = nullptr;
    bool normals_sorted = false;
    double nrm_max2 = -1.0;         // largest squared norm among the normals (< 0: not computed yet; sf_cloud_normals_max2)
    // cell-sorted positions that are actually populated: [0, n) after sf_cloud_build_grid, the slab a block needs
    // after sf_cloud_build_grid_block (positions keep their GLOBAL numbering either way)
    int64_t pop_begin = 0, pop_end = 0;
    // the z coordinates alone, ascending (the sort keys of the upload): a block build finds every z-layer's first point by a
    // binary search in it
    double *z_orig = nullptr;
    // ... and a host copy of them, made by the first block build (up to SF_Z_HOST_MAX points): the layer bounds of a block build
    // are then a few binary searches on the host -- no launch, no read-back, and the build no longer waits for whatever the
    // stream still holds (a job's next pass is queued while the last kernels of the previous one run)
    std::vector<double> z_host;
    // the counting build's per-cell counters (grid.hip::k_cell_count): all zero between builds -- the scan that reads them
    // puts the zeros back, so no build starts with a fill (count_dirty: a build failed in between; zeroed again before use)
    int32_t *cell_count = nullptr;
    int64_t cell_count_cap = 0;
    bool count_dirty = false;
    // first cell-sorted position of every z-layer of cells (dim[2] + 1 entries, host): written by the block build
    // (which needs it anyway), fetched from cell_start on demand after a whole-cloud build (sf_cloud_layer_table)
    std::vector<int64_t> layer_first;
    // mean and maximum list length of the last radius search on this cloud, per radius: what the NEXT search with that
    // radius sizes its slots from instead of counting a sample first (search.hip::run_search).  A capacity hint, nothing
    // more: a list that outgrows its slot is re-done exactly whatever the slot size was.
    std::map<std::pair<double, bool>, std::pair<double, int64_t>> list_stats; // (radius, self search?) -> (mean, longest list)
    // The lists of a SELF search are a function of (cloud, radius, GRID, query range) alone -- the points never change after the
    // upload and every build of one grid (cell edge, x subdivision) gives the same cell-sorted order -- so the host-side numbers a
    // search ends with (total, longest list, histogram of the lengths, how many lists overflowed their slot at which slot size)
    // are remembered per (radius, cell edge, xsub, first position, count): the next search of that range ON THAT GRID launches
    // its sweep and plans every later launch from the record, without the step's one read-back (round 5; SF_K2_NO_HINT=1 turns
    // it off, SF_K2_CHECK_RECORD=1 re-counts and compares).  The grid is part of the key (round 6, advisor): ensure_grid serves
    // a radius from any grid with cell in [r, 2r], and positions [begin, begin + m) of a sub-range name other points on another
    // grid -- a record of the 0.05-grid must not size the index array of the same range on the 0.03-grid.
    struct search_record {
        int64_t total = 0, n_overflow = 0, ovf_total = 0, cap = 0, hist[5] = {0, 0, 0, 0, 0};
        int32_t max_count = 0, max_count_all = 0;
        bool folded = false; // max_count_all is the maximum over every rank (sf_comm_collective_stats was on)
    };
    typedef std::tuple<double, double, int, int64_t, int64_t> search_key; // radius, cell edge, xsub, first position, +-count
    std::map<search_key, search_record> search_records;
};

struct sf_nbrs {
    int64_t m = 0;
    double radius = 0.0;
    int64_t total = 0;
    int64_t max_count = 0;
    int64_t max_count_all = 0; // the same over every rank's lists when the context folds its statistics (else = max_count)
    bool view = false;       // non-owning slice of another sf_nbrs
    bool self = false;       // queries are cloud points
    int64_t self_begin = 0;  // first sorted position when self
    double *qx = nullptr, *qy = nullptr, *qz = nullptr; // query coords in PROCESSING order (owned unless self)
    int32_t *qrow = nullptr; // processing slot -> caller's query row (null = identity)
    int32_t *count = nullptr; // per processing slot
    int64_t *offset = nullptr; // m + 1, per processing slot
    int32_t *idx = nullptr;    // total, sorted positions
    int32_t *idx_ovf = nullptr; // lists of the queries that overflowed their slot (offset[q] points into it RELATIVE TO idx)
    int64_t cap = 0;            // slot capacity of the single-sweep search (0: exact CSR)
    int64_t n_overflow = 0;     // queries whose list did not fit its slot (re-done exactly, those alone)
    // ---- per-keypoint dispatch of the list-driven kernels (K3, K5, K6, K7) -------------------------------------------
    // The register-cached / matrix-core kernels hold lists of at most 255 points (four 64-neighbour chunks).  The queries
    // whose OWN list is longer are left out by the main launch -- which takes the instantiation `main_chunks` that covers
    // the longest list it does serve -- and are served by a second launch over `tail_sel` (their processing slots,
    // ascending) in the streaming forms.  One long list never moves the other keypoints to a slower form, and which form
    // serves a keypoint depends on its own list alone (bit-identical rows for any sharding).
    int64_t hist[5] = {0, 0, 0, 0, 0}; // queries with count <= 64, <= 128, <= 192, <= 255, > 255
    bool planned = false;          // main_chunks / tail_* below were set from the statistics of a radius search
    int main_chunks = 4;
    int tail_limit = 0x7fffffff;   // lists longer than this belong to the tail launch
    int32_t *tail_sel = nullptr;   // n_tail processing slots (of the OWNING list set), ascending; shared with views
    int64_t n_tail = 0;
    // A small share of lists that need MORE chunks than the bulk (but fit the register-cached forms: <= 255 points) gets a
    // launch of its own in the 4-chunk instantiation of the SAME form -- the instantiations of one form agree bit for bit,
    // so which launch serves such a keypoint changes nothing but the bulk's speed: main_chunks covers all but <= 2 % of the
    // lists of at most 255 points, `mid_sel` names the rest (lists of 64 main_chunks + 1 .. 255 points).
    int32_t *mid_sel = nullptr;
    int64_t n_mid = 0;
    int64_t view_first = 0;        // a view's first slot in the owner's numbering (tail_sel entries are owner slots)
    // `idx` holds cell-sorted positions of ONE build of the cloud's grid, and a self search's qx / qy / qz point INTO that build's
    // arrays: a later search with another radius, a k-NN search or an explicit sf_cloud_build_grid rebuilds the grid -- other
    // cells, or the same cells in freshly allocated arrays -- and the lists then name other points or read released memory.
    // Every list set is stamped with the build it was made on (sf_cloud::grid_gen) and every consumer checks the stamp
    // (sf_nbrs_on_grid): SF_ERR_STATE instead of rows computed from the wrong points (round 6).
    uint64_t grid_gen = 0;
};

static inline void sf_nbrs_stamp(sf_nbrs *nb, const sf_cloud *c) { nb->grid_gen = c->grid_gen; }
static inline int sf_nbrs_on_grid(const sf_nbrs *nb, const sf_cloud *c, const char *who)
{
    if (nb->grid_gen == c->grid_gen) return SF_OK;
    sf_set_error("%s: the neighbour lists were made on another grid of this cloud (build %llu, now %llu): a search with another "
                 "radius, a k-NN search or sf_cloud_build_grid rebuilt it -- search again, or use the lists before the next rebuild",
                 who, (unsigned long long)nb->grid_gen, (unsigned long long)c->grid_gen);
    return SF_ERR_STATE;
}

struct sf_spfh {
    int64_t n = 0;
    int64_t rows_alloc = 0; // >= n, padded so that an all-gather of equal blocks fits
    int n_bins = 0;
    int nb3 = 0;
    int stride = 0;      // elements per row (padded)
    int elem_bytes = 2;  // 1 = biased uint8 counts (matrix-core K7), 2 = uint16 counts, 4 = uint32 counts
    int bias = 0;        // stored value = count ^ bias (128 for the uint8 table: the byte read as int8 is count - 128)
    int win_lo = 0;      // uint8 table: column c of a row holds bin win_lo + c, for c < win_len (the other bins of the nb3 are
    int win_len = 0;     // structurally empty: alpha pinned to its one or two central bins -- sf_spfh_create_for_radius);
                         // win_lo = 0, win_len = nb3 for every table of at most 128 bins
    void *counts = nullptr; // n x stride, by sorted position
    int32_t *k = nullptr;   // n, neighbourhood size (self included), by sorted position
    double *p4 = nullptr;   // uint8 table only: n x {x, y, z, (double)k} -- all the matrix-core K7 gathers per neighbour
                            // besides the table row, in ONE 32-byte record (one cache line per lane instead of three)
    unsigned *live = nullptr; // uint8 table only, four words ([2]: raised by a K7 launched in the wrong form, [3]: unused).  [0]: bit b set <=> some row computed so far has a non-zero
                              // count among bins 16 b .. 16 b + 15 (OR-accumulated by K6, never cleared: a superset is always
                              // safe); the matrix-core K7 streams and multiplies only the live 16-bin blocks of the rows.
                              // [1]: the mask under which EVERY row of `packed` was last written (~0: not valid).
    uint8_t *packed = nullptr; // uint8 table only: n x 32 bytes, the (at most) two live blocks of each row side by side --
                               // four rows per cache line instead of one for K7's gather
    uint8_t *hi = nullptr;     // uint8 table whose longest list exceeds 255 points: n x 128 bytes, count >> 8 of the rows of the
                               // points with MORE than 255 neighbours (the other rows are never written nor read).  `counts`
                               // holds count & 255 (^ 128): a long point's bins are lo + 256 hi, everybody else's rows -- and the
                               // matrix-core K7 of every keypoint whose own list fits it -- stay what they are without long points
    unsigned host_live[2] = {0xffu, ~0u}; // the two words of `live` as last read back ...
    bool host_live_valid = false;         // ... valid until the next K6 whose blocks the data decides (sf_fpfh reads them back once)
    bool mask_known = false;              // every mask that ever went into live[0] was known on the host: host_live is exact
                                          // and a function of the calls' parameters alone (the same on every rank)
};

static inline int64_t sf_div_up(int64_t a, int64_t b) { return (a + b - 1) / b; }

// How a list-driven kernel is launched over these lists (see sf_nbrs): the register-cached instantiation of the main launch
// (chunks = 1 .. 4; 0: only the streaming form fits), the list length above which the main launch leaves a query out, and
// the selection a second launch serves.  Lists that did not come out of a radius search (k-NN lists: all of one length) are
// dispatched by their longest list, as a whole.
struct sf_dispatch {
    int chunks = 4;
    int limit = 0x7fffffff;       // the main launch leaves out lists longer than this
    int tail_limit = 0x7fffffff;  // lists longer than this belong to the streaming / vector forms (255 when there are any)
    const int32_t *tail_sel = nullptr;
    int64_t n_tail = 0, view_first = 0;
    const int32_t *mid_sel = nullptr; // lists of limit + 1 .. 255 points: same form, 4 chunks, a launch of their own
    int64_t n_mid = 0;
};
static inline sf_dispatch sf_nbrs_dispatch(const sf_nbrs *nb)
{
    sf_dispatch d;
    if (nb->planned) {
        d.chunks = nb->main_chunks;
        d.limit = nb->n_mid ? 64 * nb->main_chunks : nb->tail_limit;
        d.tail_limit = nb->tail_limit;
        d.tail_sel = nb->tail_sel;
        d.n_tail = nb->n_tail;
        d.mid_sel = nb->mid_sel;
        d.n_mid = nb->n_mid;
        d.view_first = nb->view_first;
    } else {
        const int64_t mx = nb->max_count > 0 ? nb->max_count : 1;
        d.chunks = mx <= 256 ? (int)sf_div_up(mx, 64) : 0;
    }
    return d;
}
