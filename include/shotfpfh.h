/*
 * shotfpfh.h -- C ABI of libshotfpfh.so, the MI355X (gfx950) SHOT / FPFH descriptor engine.
 *
 * This is the drop-in boundary for the hot path of aubin-tchoi/shot-fpfh.  The reference is pure
 * Python with no FFI of its own; each entry point below replaces the NumPy / scikit-learn /
 * SciPy call named next to it (file:line into the reference checkout) and is what a ctypes stub
 * in the reference would bind.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes.  No torch / numpy types.
 *   - Every int-returning call returns SF_OK (0) or a negative SF_ERR_* code; the message is
 *     available from sf_last_error() (thread-local).  Handle-returning calls return NULL on error.
 *   - All floating-point data is float64, row-major.  Indices into the cloud are int32/int64 in
 *     the caller's ORIGINAL point numbering unless a parameter says otherwise.
 *   - `flags` tells where the data pointers of a call live:
 *       SF_HOST (0)      every data pointer is host memory (the library copies in/out)
 *       SF_OUT_DEVICE    output pointers are device memory of ctx's GPU (no D2H copy)
 *       SF_IN_DEVICE     input data pointers are device memory (no H2D copy)
 *     Handles (sf_cloud, sf_nbrs, sf_spfh) always live on the device.
 *   - The library never keeps a caller pointer after a call returns.
 *   - One sf_ctx per GPU and per host thread; calls on one ctx are stream-ordered on the ctx's
 *     own HIP stream (sf_stream).  Do not fork() after sf_create().
 *   - There is NO CPU fallback: without a GPU sf_create() fails.
 */
#ifndef SHOTFPFH_H
#define SHOTFPFH_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SF_OK 0
#define SF_ERR_ARG (-1)
#define SF_ERR_HIP (-2)
#define SF_ERR_NOMEM (-3)
#define SF_ERR_COMM (-4)
#define SF_ERR_STATE (-5)
#define SF_ERR_UNSUPPORTED (-6)

#define SF_HOST 0
#define SF_OUT_DEVICE 1
#define SF_IN_DEVICE 2

#define SF_SHOT_LEN 352 /* 11 cosine x 8 azimuth x 2 elevation x 2 radial bins (shot.py:195) */
#define SF_FAST_FPFH_BINS 8 /* n_bins up to here: LDS-histogram K6 and the matrix-core / streaming K7 */
#define SF_MAX_FPFH_BINS 1290 /* n_bins^3 must fit an int; the reference takes any n_bins (fpfh.py:16).  What bounds n_bins in
                                practice is the table of n x n_bins^3 32-bit counts in HBM (the reference holds twice that in host RAM) */

typedef struct sf_ctx sf_ctx;     /* one GPU: device id, stream, scratch, optional RCCL comm  */
typedef struct sf_cloud sf_cloud; /* resident point cloud + uniform grid (replaces KDTree(X))  */
typedef struct sf_nbrs sf_nbrs;   /* resident CSR radius-neighbour lists of a query set        */
typedef struct sf_spfh sf_spfh;   /* resident SPFH table (integer bin counts + list lengths)   */
typedef struct sf_voxels sf_voxels; /* voxel partition of a point set (grid_subsampling)       */

/* ---- library / context ---------------------------------------------------------------- */
const char *sf_last_error(void);
const char *sf_version(void);
int sf_device_count(void);
sf_ctx *sf_create(int device);
void sf_destroy(sf_ctx *ctx);
int sf_sync(sf_ctx *ctx);
void *sf_stream(sf_ctx *ctx); /* hipStream_t the kernels are launched on */
/* A context owns two HIP streams.  sf_fork: following calls go to the side stream, ordered after the work issued so
 * far; sf_switch(0|1) picks the stream of the following calls while forked; sf_join: back on the main stream, ordered
 * after the side stream.  While forked only calls that neither allocate nor free device memory may be issued
 * (resident K4-K7 calls).  Used to run the FPFH chain and the SHOT chain of one pass side by side. */
int sf_fork(sf_ctx *ctx);
int sf_switch(sf_ctx *ctx, int side);
int sf_join(sf_ctx *ctx);
/* sf_mark: remember the point reached on the current stream; sf_wait_mark: the current stream waits for that point -- and
 * only for it, not for work issued on the marked stream later (sf_join waits for everything). */
int sf_mark(sf_ctx *ctx);
int sf_wait_mark(sf_ctx *ctx);

/* ---- raw device memory (for callers that keep results resident) ------------------------- */
void *sf_dev_alloc(sf_ctx *ctx, size_t bytes);
int sf_dev_free(sf_ctx *ctx, void *dev_ptr);
/* Page-locked host memory: host output pointers of any call may point into it, and the device-to-host copy is then one
 * DMA at PCIe speed instead of a staged copy into pageable memory.  Pinning is slow: allocate once, reuse.
 * sf_host_free may be called after sf_destroy (ctx is ignored). */
void *sf_host_alloc(sf_ctx *ctx, size_t bytes);
int sf_host_free(sf_ctx *ctx, void *host_ptr);
int sf_h2d(sf_ctx *ctx, void *dst_dev, const void *src_host, size_t bytes);
int sf_d2h(sf_ctx *ctx, void *dst_host, const void *src_dev, size_t bytes);
int sf_d2d(sf_ctx *ctx, void *dst_dev, const void *src_dev, size_t bytes); /* stream-ordered, asynchronous */

/* ---- cloud + grid: replaces sklearn KDTree(X) -------------------------------------------
 * call sites: fpfh.py:26, shot_parallelization.py:167,220,229,283, pca_based_descriptors.py:45-49
 * sf_cloud_upload copies xyz (n x 3) and normals (n x 3, nullable) to the GPU.
 * sf_cloud_build_grid (kernel K1) bins the points into cells of edge >= `cell`, sorts them by cell
 * and keeps cell-sorted SoA copies of xyz / normals; it must be called with cell >= the largest
 * radius later searched (re-callable; a search with a larger radius -- or one below half the cell, for speed -- and every
 * k-NN search rebuild automatically).  Every build re-sorts into fresh arrays: list sets (sf_nbrs) made on an earlier build
 * are refused by their consumers with SF_ERR_STATE from then on -- use a search's lists before the next rebuild. */
sf_cloud *sf_cloud_upload(sf_ctx *ctx, const double *xyz, const double *normals, int64_t n, int flags);
int sf_cloud_set_normals(sf_ctx *ctx, sf_cloud *cloud, const double *normals, int flags);
int sf_cloud_build_grid(sf_ctx *ctx, sf_cloud *cloud, double cell);
/* Multi-GPU variant of K1: build only what the queries at cell-sorted positions [begin, end) -- a rank's block --
 * can reach within `reach` grid cells (1 for SHOT / normals / SPFH of the block, 2 when the SPFH rows of the
 * block's one-cell halo are recomputed locally for FPFH).  Whole z-layers of cells are kept; positions, perm and
 * the cell table keep their GLOBAL numbering, so block / halo ranges and every result are exactly those of a
 * whole-cloud build, but only ~1/N of the cloud is sorted and gathered.  [*pop_begin, *pop_end) (nullable) returns
 * the populated position range; self-searches must stay inside it (anything else rebuilds the whole grid). */
int sf_cloud_build_grid_block(sf_ctx *ctx, sf_cloud *cloud, double cell, int64_t begin, int64_t end, int reach,
                              int64_t *pop_begin, int64_t *pop_end);
int64_t sf_cloud_size(const sf_cloud *cloud);
void sf_cloud_free(sf_ctx *ctx, sf_cloud *cloud);
/* cell-sorted position -> original index (n entries, host) */
int sf_cloud_perm(sf_ctx *ctx, sf_cloud *cloud, int32_t *perm);
/* Multi-GPU sharding helper: the smallest range [halo_begin, halo_end) of cell-sorted positions that
 * contains every point within one grid cell (>= the search radius) of the block [begin, end).  The grid
 * orders cells z-slowest, so a block is a z-slab and its halo is one contiguous run on each side. */
int sf_cloud_halo_range(sf_ctx *ctx, sf_cloud *cloud, int64_t begin, int64_t end, int64_t *halo_begin,
                        int64_t *halo_end);
/* first[z] = first cell-sorted position of z-layer z of the grid's cells, z = 0 .. *n_layers (first[*n_layers] = n);
 * `first` (host, nullable) has room for `cap` entries.  A block build keeps the table for the WHOLE replicated cloud
 * (it comes out of its per-layer histogram), so every rank can work out every rank's block, halo and the rows two
 * ranks exchange (shot_fpfh_amd/sharding.py) without talking to anybody. */
int sf_cloud_layer_table(sf_ctx *ctx, sf_cloud *cloud, int64_t *first, int64_t cap, int64_t *n_layers);

/* ---- radius search: replaces KDTree.query_radius (kernels K2 count / fill) --------------
 * Inclusion rule of sklearn's euclidean KDTree: ((dx*dx + dy*dy) + dz*dz) <= r*r in float64,
 * evaluated without FMA, self-match included.
 * sf_radius_search:        m arbitrary query points (shot_parallelization.py:167-169,
 *                          pca_based_descriptors.py:48).
 * sf_radius_search_self:   the cloud's own points as queries (fpfh.py:28-30); [begin, end) is a
 *                          range of CELL-SORTED positions (0, n = everything), which is how a
 *                          multi-GPU shard selects its block of queries.
 * sf_nbrs_export returns the lists in the caller's numbering, ascending inside each list, with
 * distances sqrt(d2) when `dist` is non-NULL (KDTree return_distance=True). Host pointers only. */
sf_nbrs *sf_radius_search(sf_ctx *ctx, sf_cloud *cloud, const double *queries, int64_t m, double radius,
                          int flags);
sf_nbrs *sf_radius_search_self(sf_ctx *ctx, sf_cloud *cloud, double radius, int64_t begin, int64_t end);
/* k nearest neighbours: replaces KDTree.query(Q, k=k, return_distance=False) (pca_based_descriptors.py:46).
 * Every list has exactly k entries, nearest first (ties: lower cell-sorted position).  1 <= k <= n (k <= 1984: the k best
 * are kept in LDS during one sweep; beyond: count / fill / segmented sort through global memory). */
sf_nbrs *sf_knn_search(sf_ctx *ctx, sf_cloud *cloud, const double *queries, int64_t m, int k, int flags);
/* Caller-supplied neighbourhoods: ShotMultiprocessor.compute_local_rf / compute_descriptor work on the lists they are HANDED --
 * support[neighborhoods[i]], shot_parallelization.py:46-84, 86-133 -- whatever made them (KDTree.query_radius of another radius,
 * KDTree.query, a selection of the caller's).  sf_nbrs_import makes a list set of them that sf_shot_lrf / sf_shot / sf_normals /
 * sf_pca consume like a search result.  queries: m x 3 (row i = keypoint i); offsets: m + 1 ascending from 0; idx:
 * offsets[m] point indices in the numbering of sf_cloud_upload (each in 0 .. n-1, else SF_ERR_ARG through NULL + sf_last_error);
 * radius: the value get_local_rf / compute_single_shot_descriptor are called with (shot.py:16-48, 175-306) -- it enters their
 * formulas and filters nothing.  flags: SF_HOST, or SF_IN_DEVICE with all three arrays on the device.  Like every list set, the
 * result is bound to the cloud's current grid: use it before the next search with another radius (consumers return
 * SF_ERR_STATE otherwise). */
sf_nbrs *sf_nbrs_import(sf_ctx *ctx, sf_cloud *cloud, const double *queries, int64_t m, const int64_t *offsets,
                        const int64_t *idx, double radius, int flags);
/* non-owning view of queries [first, first+count) of `nbrs` (free it before the parent) */
sf_nbrs *sf_nbrs_slice(sf_ctx *ctx, sf_nbrs *nbrs, int64_t first, int64_t count);
int64_t sf_nbrs_num_queries(const sf_nbrs *nbrs);
int64_t sf_nbrs_total(const sf_nbrs *nbrs);
int64_t sf_nbrs_max_count(const sf_nbrs *nbrs);
/* the longest list over EVERY rank's searches when sf_comm_collective_stats is on (else = sf_nbrs_max_count) */
int64_t sf_nbrs_max_count_all(const sf_nbrs *nbrs);
int sf_nbrs_export(sf_ctx *ctx, sf_cloud *cloud, sf_nbrs *nbrs, int64_t *offsets /* m+1 */,
                   int32_t *idx /* total */, double *dist /* nullable, total */);
void sf_nbrs_free(sf_ctx *ctx, sf_nbrs *nbrs);

/* ---- PCA normals: compute_normals radius branch (pca_based_descriptors.py:15-59), K3 ----
 * `nbrs` = lists of the query points. pre (m x 3, nullable) = pre_computed_normals. Without it the
 * sign is the one LAPACK dsyevd returns for the lower-triangle covariance (emulated on device). */
/* compute_normals(query_points, cloud_points, radius=...) -- pca_based_descriptors.py:29-59, the radius branch (:48) -- in ONE
 * sweep: the neighbour lists are never materialised (the hits of the candidate sweep are reduced to the covariance in LDS).
 * queries == NULL: the cloud's own points at cell-sorted positions [begin, end) (row i of `out` = position begin + i; m is
 * ignored); else m coordinate queries, rows in the caller's order.  `pre` (nullable): pre_computed_normals, m x 3.
 * Bit-identical to sf_radius_search(_self) followed by sf_normals. */
int sf_normals_radius(sf_ctx *ctx, sf_cloud *cloud, const double *queries, int64_t m, int64_t begin, int64_t end, double radius,
                      const double *pre, double *out /* m x 3 */, int flags);
int sf_normals(sf_ctx *ctx, sf_cloud *cloud, sf_nbrs *nbrs, const double *pre, double *out /* m x 3 */,
               int flags);
/* Local PCA of every query's neighbourhood, same kernel (K3) with the full decomposition as output: replaces
 * the per-point loops of compute_sphericity / compute_pca_based_basic_features (pca_based_descriptors.py:60-73,
 * 150-187, via pca() :15-26) and compute_local_pca_with_moments (:75-146).  eigenvalues: ascending, m x 3.
 * eigenvectors: m x 3 x 3 row-major exactly as numpy.linalg.eigh returns them (column k = eigenvector k, LAPACK
 * dsyevd signs).  moments (nullable, m x 8): |mean(c V^T)| (3), mean((c V^T)^2) (3), mean(c_z), mean(c_z^2) with c
 * the neighbours centred on their barycentre (:121-144).  The feature formulas on top stay on the host. */
int sf_pca(sf_ctx *ctx, sf_cloud *cloud, sf_nbrs *nbrs, double *eigenvalues /* m x 3 */,
           double *eigenvectors /* m x 9 */, double *moments /* nullable, m x 8 */, int flags);

/* ---- SHOT --------------------------------------------------------------------------------
 * sf_shot_lrf (K4): get_local_rf (shot.py:16-48) for every query of `nbrs`; radius = the search
 *   radius of `nbrs`.  lrf is m x 9, row-major 3x3 with COLUMNS x, y, z (shot.py:48).
 * sf_shot (K5): compute_single_shot_descriptor (shot.py:175-306) incl. the last-writer-wins
 *   semantics of the ten fancy-index "+=" statements (shot.py:244-298).  out is m x 352. */
int sf_shot_lrf(sf_ctx *ctx, sf_cloud *cloud, sf_nbrs *nbrs, double *lrf /* m x 9 */, int flags);
int sf_shot(sf_ctx *ctx, sf_cloud *cloud, sf_nbrs *nbrs, const double *lrf /* m x 9 */, int normalize,
            int64_t min_neighborhood_size, double *out /* m x 352 */, int flags);
/* compute_descriptor_single_scale in one call (shot_parallelization.py:135-183): frames and descriptors from the
 * same lists; the frame's sign votes are fused into K5.  lrf (m x 9, nullable) receives the frames. */
int sf_shot_single_scale(sf_ctx *ctx, sf_cloud *cloud, sf_nbrs *nbrs, int normalize, int64_t min_neighborhood_size,
                         double *lrf /* m x 9, nullable */, double *out /* m x 352 */, int flags);

/* get_azimuth_idx (shot.py:51-70), elementwise: octant 0..7 of (x, y), a boundary ray belonging to the LOWER octant
 * (+x -> 3, +y -> 5, -x -> 7, -y -> 1, diagonals 4 / 6 / 0 / 2, origin -> 0).  The device function K5 bins with. */
int sf_azimuth_idx(sf_ctx *ctx, const double *x, const double *y, int64_t n, int64_t *idx, int flags);
/* compute_shot_descriptor, the reference's serial variant (shot.py:310-499; "debug only", not in descriptors.__all__):
 * per keypoint the gate `#(rho > 0) > min_neighborhood_size` (:360), the frame from the neighbours at NON-ZERO distance
 * only (:361-363 -- the keypoint and its duplicates neither weigh in the covariance normaliser nor vote), the ten
 * statements of sf_shot, and a row that is always L2-normalised (:496-497). */
int sf_shot_serial(sf_ctx *ctx, sf_cloud *cloud, sf_nbrs *nbrs, int64_t min_neighborhood_size, double *out /* m x 352 */,
                   int flags);

/* ---- a repeated step as ONE launch (HIP graph) -------------------------------------------------------------------------
 * sf_graph_begin .. sf_graph_end capture every device operation the calls in between issue on the context's streams (nothing
 * runs yet) into an executable graph; sf_graph_launch replays it on the main stream.  What is captured must not synchronise with
 * the device: a REPEATED step of the path qualifies (same cloud, radius and buffers: every launch is planned from the previous
 * search's record), a first one does not -- run the step once eagerly and capture only if sf_sync_count() did not move (a
 * capture that hits a wait is invalidated by the runtime: sf_graph_end then returns NULL with its reason).  The arguments of the
 * captured calls (device buffers, sizes) are part of the graph: replay only while they are alive and unchanged.  Launch timers
 * (sf_profile) must be off.  Not part of the reference's API: a host-overhead lever for small, repeated steps (a rank's share
 * of a strong-scaled job). */
typedef struct sf_graph sf_graph;
unsigned long long sf_sync_count(void); /* host waits on a stream made by the library so far, in this process */
int sf_graph_begin(sf_ctx *ctx);
sf_graph *sf_graph_end(sf_ctx *ctx);
int sf_graph_launch(sf_ctx *ctx, sf_graph *graph);
void sf_graph_free(sf_ctx *ctx, sf_graph *graph);

/* ---- FPFH: compute_fpfh_descriptor (fpfh.py:16-117, decorrelated=False) -------------------
 * sf_spfh_create allocates the table for all n cloud points; sf_spfh_compute (K6) fills the rows of
 * the queries of `self_nbrs` (a sf_radius_search_self result); edges = 3 x (n_bins+1) histogram
 * edges exactly as np.histogramdd builds them (np.linspace; fpfh.py:82-87).
 * sf_fpfh (K7) reduces  spfh[kp] + sum_{j, d_j>0} spfh[j]/d_j / k  (fpfh.py:101-116) for keypoints
 * given either as original indices kp_idx (m; needs lists of the whole cloud) or, when kp_idx == NULL,
 * for every query of `self_nbrs` in cell-sorted order (m must equal its query count; with a
 * sf_nbrs_slice view this is how a shard reduces only its own block).
 * sf_spfh_export writes the float64 SPFH table (n x n_bins^3, original numbering).
 * max_count (the largest neighbourhood the table will see, sf_nbrs_max_count; over every rank of a sharded job:
 * sf_nbrs_max_count_all) picks the storage of the integer bin counts: bytes for n_bins <= 5 and lists of at most 65535
 * points -- with a second table for count >> 8 of the points that have more than 255 neighbours, allocated when max_count
 * exceeds 255 --, else 16 bits up to 65535, 32 bits beyond.  Every list-driven entry point dispatches per keypoint by list
 * length: a list of more than 255 points is served by a second launch, it never changes the kernels the others run.
 * On the byte table sf_spfh_compute also records which 16-bin blocks of the rows hold any count; sf_fpfh reads that
 * mask back (8 bytes, once per sf_spfh_compute: it waits for K6 on the context's stream) and, when at most two of the
 * eight blocks do, multiplies only those -- same results, bit for bit. */
sf_spfh *sf_spfh_create(sf_ctx *ctx, sf_cloud *cloud, int n_bins, int64_t max_count);
/* The table for a KNOWN search radius (the `radius` argument of compute_fpfh_descriptor, fpfh.py:19).  alpha = (c x u) . n_j
 * with v not normalised (fpfh.py:60) never exceeds radius * max|n|^2: when that stays inside the one or two central bins of
 * the alpha histogram (an even n_bins has an edge at 0), only those bins' n_bins^2 (2 n_bins^2) slots of a row can ever hold a
 * count.  For n_bins = 6, 7, 8, 9, 11 -- 72 of 216, 49 of 343, 128 of 512, 81 of 729, 121 of 1331 bins: every count whose
 * window fits the 128 columns of a byte row -- the table then keeps exactly that window of bins,
 * one byte each, and the whole byte-table path (packed rows, high bytes, K7 on the matrix cores, the wire image of the
 * exchange) serves these bin counts too; sf_fpfh writes zeros for the bins outside the window.  Without a usable window
 * (radius <= 0, no normals, a larger reach, other bin counts) this IS sf_spfh_create.  sf_spfh_compute fails with
 * SF_ERR_STATE when it is run with a radius that reaches beyond the table's window.  sf_spfh_elem_bytes: 1, 2 or 4. */
sf_spfh *sf_spfh_create_for_radius(sf_ctx *ctx, sf_cloud *cloud, int n_bins, int64_t max_count, double radius);
int sf_spfh_elem_bytes(const sf_spfh *spfh);
int sf_spfh_compute(sf_ctx *ctx, sf_cloud *cloud, sf_nbrs *self_nbrs, sf_spfh *spfh, const double *edges);
/* Extension for callers that want FPFH and SHOT from the SAME self-search lists (both descriptors of config 3 / 5):
 * sf_spfh_compute_moments is sf_spfh_compute that also leaves, per query of `self_nbrs`, the weighted covariance of
 * the SHOT local frame (get_local_rf, shot.py:27-35; 6 doubles c11 c21 c31 c22 c32 c33, device memory) -- the
 * neighbours are gathered once instead of twice.  sf_shot_from_moments is sf_shot_single_scale starting from those
 * moments (eigen-solves, then the fused K5); `nbrs` may be a sf_nbrs_slice view with cov advanced accordingly. */
int sf_spfh_compute_moments(sf_ctx *ctx, sf_cloud *cloud, sf_nbrs *self_nbrs, sf_spfh *spfh, const double *edges,
                            double *cov_dev /* m x 6 */);
int sf_shot_from_moments(sf_ctx *ctx, sf_cloud *cloud, sf_nbrs *nbrs, const double *cov_dev /* m x 6 */, int normalize,
                         int64_t min_neighborhood_size, double *lrf /* m x 9, nullable */, double *out /* m x 352 */,
                         int flags);
/* The two halves of sf_shot_from_moments (device pointers only): the eigen-solves, which leave the raw axes in
 * lrf_dev, and the fused K5, which completes the frames in place and writes the descriptors. */
int sf_lrf_raw_from_moments(sf_ctx *ctx, sf_cloud *cloud, sf_nbrs *nbrs, const double *cov_dev, double *lrf_dev);
int sf_shot_from_raw_lrf(sf_ctx *ctx, sf_cloud *cloud, sf_nbrs *nbrs, double *lrf_dev, int normalize,
                         int64_t min_neighborhood_size, double *out_dev);
int sf_spfh_allgather(sf_ctx *ctx, sf_spfh *spfh, int64_t rows_per_rank); /* RCCL, in place */
/* Neighbour-to-neighbour exchange of SPFH rows in one RCCL send/recv group: operation i sends this rank's rows
 * [send_begin[i], send_end[i]) to rank peer[i] and receives that rank's rows into [recv_begin[i], recv_end[i]) (cell-sorted
 * positions; the two sides of an operation must name equally many rows).  Per row only what K7 reads per neighbour
 * travels (64 B on the byte table with at most two live blocks).  See fpfh.hip. */
int sf_spfh_exchange_rows(sf_ctx *ctx, sf_spfh *spfh, int n_ops, const int *peer, const int64_t *send_begin,
                          const int64_t *send_end, const int64_t *recv_begin, const int64_t *recv_end);
/* The same wire image of rows [begin, end) through HOST memory, for transports other than RCCL: write_back = 0 copies it
 * out of the table into `host` (cap bytes of room), 1 copies it from `host` into the table's rows; *bytes (nullable)
 * receives the image size, and host == NULL only queries it. */
int sf_spfh_rows_image(sf_ctx *ctx, sf_spfh *spfh, int64_t begin, int64_t end, void *host, size_t cap, int write_back,
                       size_t *bytes);
int sf_spfh_export(sf_ctx *ctx, sf_cloud *cloud, sf_spfh *spfh, double *out /* n x nb^3 */, int flags);
void sf_spfh_free(sf_ctx *ctx, sf_spfh *spfh);
int sf_fpfh(sf_ctx *ctx, sf_cloud *cloud, sf_nbrs *self_nbrs, sf_spfh *spfh, const int64_t *kp_idx, int64_t m,
            double *out /* m x nb^3 */, int flags);

/* ---- matching: cdist + argmin (matching.py:47-52, 63-65, 164-168), K8 ---------------------
 * a: m1 x d, b: m2 x d.  idx[i] = first j minimising sqrt(sum_t (a[i,t]-b[j,t])^2), summed left to
 * right in float64 without FMA (scipy's cdist loop); dist (nullable) = that distance; col_idx
 * (nullable, m2) = argmin over axis 0 (for the reciprocity test). */
int sf_match_argmin(sf_ctx *ctx, const double *a, int64_t m1, const double *b, int64_t m2, int64_t d,
                    int64_t *idx, double *dist, int64_t *col_idx, int flags);

/* Multi-scale ("minimum over scales") form of match_descriptors (matching.py:77-136).  a: n_scales x m1 x d,
 * b: n_scales x m2 x d; a_ok / b_ok: n_scales x m bytes, 1 where the row has a non-zero entry at that scale.
 * dist(i,j) = min over scales of (a_ok && b_ok ? euclidean distance : max_val); idx = first arg-min over j. */
/* mask_dev[i] = 1 when row i of the device matrix rows_dev (m x d) has a non-zero entry (np.any(desc, axis=1)). */
int sf_rows_nonzero(sf_ctx *ctx, const double *rows_dev, int64_t m, int64_t d, unsigned char *mask_dev);
/* out_dev row i = row sel_dev[i] of rows_dev (n_rows x d; all device memory), or a zero row where sel_dev[i] < 0: selects a
 * keypoint subset of a resident descriptor matrix, padded to the equal per-rank block an all-gather needs (zero rows are
 * skipped by the matching, matching.py:162-163).  A selection >= n_rows is never dereferenced: its row comes out zero and
 * the next synchronising call on the context (sf_sync, sf_d2h) returns SF_ERR_ARG. */
int sf_rows_gather(sf_ctx *ctx, const double *rows_dev, int64_t n_rows, const int64_t *sel_dev, int64_t m, int64_t d,
                   double *out_dev);
/* flags: SF_HOST, or SF_IN_DEVICE | SF_OUT_DEVICE with every pointer (masks included) on the device -- the form
 * the sharded matching uses with n_scales = 1 and max_val = +inf to skip all-zero descriptors in place. */
int sf_match_argmin_multiscale(sf_ctx *ctx, const double *a, const double *b, int n_scales, int64_t m1, int64_t m2,
                               int64_t d, const unsigned char *a_ok, const unsigned char *b_ok, double max_val,
                               int64_t *idx, double *dist, int flags);

/* Reciprocity test of match_descriptors (matching.py:62-74) when the scan rows are sharded over ranks: with each rank's
 * column arg-min over its own scan block (sf_match_argmin_multiscale, operands swapped) and the all-reduced column minimum
 * (sf_comm_allreduce_min_u64 on the distances' bit patterns), cand[j] = row_offset + local_idx[j] where this rank attains
 * the global minimum of column j, ~0 elsewhere; a second all-reduce(min) of `cand` leaves distance_matrix.argmin(axis=0),
 * first minimum included.  All pointers device memory. */
int sf_match_col_candidates(sf_ctx *ctx, const double *local_dist_dev, const double *global_dist_dev,
                            const int64_t *local_idx_dev, int64_t row_offset, int64_t m, void *cand_dev);

/* K8 while the reference rows are still ARRIVING (a sharded matching: every scan row must see every reference row, and the rows
 * of the other ranks cross xGMI in chunks -- sf_comm_exchange on the context's side stream).  sf_match_stream_begin takes the
 * resident scan rows a (m1 x d, mask a_ok) and the buffer b (m2 x d, mask b_ok) the reference rows will land in, in their final
 * order; b_entry_max: the largest |entry| of ANY reference row (sf_rows_abs_max of the local block, max over the ranks -- any
 * positive value is valid, the quantisation error is measured, a poor one only costs speed); max_ranges: how many
 * sf_match_stream_feed calls will follow.  sf_match_stream_feed(begin, end): rows [begin, end) of b AND of b_ok are now in place
 * (begin a multiple of 64, end a multiple of 64 or m2; every row exactly once, any order): their int8 image is made and the
 * integer pass (match_i8.hip) takes its minima over them for every scan row, on the context's current stream, while the next
 * chunk travels.  sf_match_stream_end: the decision steps over all chunks' minima -- idx / dist as sf_match_argmin_multiscale
 * (n_scales = 1) leaves them, bit for bit; the handle is released.  When the integer pass does not suit the problem (d > 352,
 * most rows without a clear nearest descriptor, SF_MATCH_I8=0) the feeds cost nothing and _end runs the one-shot paths.
 * All pointers device memory. */
typedef struct sf_match_stream sf_match_stream;
sf_match_stream *sf_match_stream_begin(sf_ctx *ctx, const double *a_dev, const unsigned char *a_ok_dev, int64_t m1,
                                       const double *b_dev, const unsigned char *b_ok_dev, int64_t m2, int64_t d,
                                       double b_entry_max, int64_t max_ranges);
int sf_match_stream_feed(sf_ctx *ctx, sf_match_stream *stream, int64_t row_begin, int64_t row_end);
int sf_match_stream_end(sf_ctx *ctx, sf_match_stream *stream, int64_t *idx_dev, double *dist_dev /* nullable */);
void sf_match_stream_abort(sf_ctx *ctx, sf_match_stream *stream);
/* largest |entry| of m x d resident rows (+inf when an entry is not finite) -> *out_host */
int sf_rows_abs_max(sf_ctx *ctx, const double *rows_dev, int64_t m, int64_t d, double *out_host);

/* ---- RANSAC scoring: inlier count of ransac.py:60-67, K9 ----------------------------------
 * a, b: m x 3 matched points; Rt: n_draws x 12 (row-major R, then t); counts ||a R^T + t - b|| <= thr. */
int sf_ransac_score(sf_ctx *ctx, const double *a, const double *b, int64_t m, const double *Rt, int64_t n_draws,
                    double thr, int64_t *inliers, int flags);

/* ---- voxel subsampling: grid_subsampling (core/subsampling.py:5-39) and the voxel loop of
 * select_keypoints_with_density_threshold (keypoint_selection.py:80-101) ---------------------------------
 * sf_voxels_build: keys ((p - min p) // voxel).astype(int) with NumPy's floor_divide, voxels ranked in np.unique's
 *   lexicographic key order (stable device sort).  xyz: n x 3, host or (SF_IN_DEVICE) device memory that must stay valid.
 * sf_voxels_count: number of occupied voxels.
 * sf_voxels_inverse: np.unique's `inverse` (voxel rank of every point, n, host).
 * sf_voxels_select: per voxel the point closest to the voxel's barycentre (first minimum) and, optionally, the voxel
 *   populations (np.unique's counts).  `order` (nullable, n, host) is the visiting order of the points, grouped by voxel
 *   in voxel order -- the reference uses np.argsort(inverse), an UNSTABLE sort whose order decides exact distance ties
 *   (every two-point voxel is one); NULL = ascending point index inside a voxel. */
sf_voxels *sf_voxels_build(sf_ctx *ctx, const double *xyz, int64_t n, double voxel_size, int flags);
int64_t sf_voxels_count(const sf_voxels *vox);
int sf_voxels_inverse(sf_ctx *ctx, sf_voxels *vox, int64_t *inverse /* n */);
int sf_voxels_select(sf_ctx *ctx, sf_voxels *vox, const int64_t *order /* nullable, n */, int64_t *selected /* count */,
                     int64_t *counts /* nullable, count */);
void sf_voxels_free(sf_ctx *ctx, sf_voxels *vox);

/* ---- ICP: one iteration's device work (icp.py:64-72, 108-124, 160-183; core/solvers.py:17-18, 38-46) --------
 * sf_icp_accumulate: rows sel_dev[0..m) (or 0..m when NULL) of the resident points pts_dev are moved by Rt (12 host
 *   doubles: R row-major then t; NULL = identity), their nearest reference point found (KDTree.query, k = 1), pairs
 *   with distance <= d_max kept, and the sums of the iteration's solver returned in sums[40] (host):
 *     [0] inlier count, [1..3] sum of inlier points p, [4..6] sum of their neighbours q,
 *     mode 0 (point to point): [8..16] sum (p - pbar)(q - qbar)^T row-major, [17] sum |p - q|^2
 *     mode 1 (point to plane; the cloud needs normals): [8..28] upper triangle of G^T G row by row, [29..34] G^T h,
 *       [35] sum |h|, with g = [p x n, n], h = (q - p) . n.
 * sf_transform_points: p <- p R^T + t in place on resident points (RigidTransform.__getitem__). */
int sf_icp_accumulate(sf_ctx *ctx, sf_cloud *ref, const double *pts_dev, const int64_t *sel_dev, int64_t m, const double *Rt,
                      double d_max, int mode, double *sums /* 40 */);
int sf_transform_points(sf_ctx *ctx, double *pts_dev, int64_t n, const double *Rt /* 12 */);

/* ---- multi-GPU: RCCL over xGMI (no counterpart in the reference) --------------------------
 * One process per GPU.  Rank 0 calls sf_comm_unique_id, ships the 128 bytes to the other ranks by
 * any host channel, then every rank calls sf_comm_init.  sf_comm_allgather gathers
 * bytes_per_rank from each rank into recv (rank-major); send may alias recv + rank*bytes_per_rank. */
int sf_comm_unique_id(char id[128]);
int sf_comm_init(sf_ctx *ctx, const char id[128], int nranks, int rank);
int sf_comm_allgather(sf_ctx *ctx, const void *send_dev, void *recv_dev, size_t bytes_per_rank);
/* Grouped point-to-point exchange (ncclSend / ncclRecv in one group): operation i sends send_bytes[i] bytes to rank
 * peer[i] and receives recv_bytes[i] bytes from it; a rank may name itself. */
int sf_comm_exchange(sf_ctx *ctx, int n_ops, const int *peer, const void *const *send_dev, const size_t *send_bytes,
                     void *const *recv_dev, const size_t *recv_bytes);
/* element-wise minimum over the ranks of n unsigned 64-bit words (ncclAllReduce; in place allowed): the column arg-min of
 * a sharded matching travels as packed (distance bits, index) keys */
int sf_comm_allreduce_min_u64(sf_ctx *ctx, const void *send_dev, void *recv_dev, size_t n);
/* on: every radius search of the context also folds its longest-list statistic over all ranks (collective call) */
int sf_comm_collective_stats(sf_ctx *ctx, int on);
int sf_comm_destroy(sf_ctx *ctx);

/* ---- per-kernel timing with HIP events on the ctx stream -------------------------------- */
int sf_profile_enable(sf_ctx *ctx, int on);
int sf_profile_reset(sf_ctx *ctx);
/* time launches of this kernel name only (NULL or "" = all): two event records per launch cost a few microseconds, which a
 * caller timing whole passes may not want on every kernel */
int sf_profile_only(sf_ctx *ctx, const char *name);
/* writes "name launches total_ms\n" lines; returns bytes needed (call with buf == NULL to size) */
int64_t sf_profile_report(sf_ctx *ctx, char *buf, int64_t cap);

#ifdef __cplusplus
}
#endif
#endif /* SHOTFPFH_H */
