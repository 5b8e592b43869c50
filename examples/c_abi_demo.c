/* A caller of the C ABI that is not Python: FPFH (5 bins) for a subset of a cloud's points and SHOT for the same points,
 * through include/shotfpfh.h alone -- the calls a cgo / JNI / ctypes binding of the reference's
 * compute_fpfh_descriptor (fpfh.py:16) and ShotMultiprocessor.compute_descriptor_single_scale
 * (shot_parallelization.py:135) would make.
 *
 *   gcc -std=c99 -O1 -I include examples/c_abi_demo.c -L shot_fpfh_amd -lshotfpfh -Wl,-rpath,$PWD/shot_fpfh_amd -o c_abi_demo
 *   ./c_abi_demo cloud.bin rows.bin 0.05
 *
 * cloud.bin: int64 n, int64 m, n x 3 float64 points, n x 3 float64 normals, m int64 keypoint indices
 * rows.bin : m x 125 float64 FPFH rows, then m x 352 float64 SHOT rows
 * (tests/test_abi.py compiles it on the CPU; tests/test_hip_round5.py runs it and compares the rows with the Python drop-ins',
 * bit for bit.) */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "shotfpfh.h"

#define CHECK(call)                                                                  \
    do {                                                                             \
        if ((call) != SF_OK) {                                                       \
            fprintf(stderr, "%s failed: %s\n", #call, sf_last_error());              \
            return 2;                                                                \
        }                                                                            \
    } while (0)

/* np.linspace(start, stop, num) as NumPy evaluates it: arange * step + start, last element = stop */
static void linspace(double start, double stop, int num, double *out)
{
    const double step = (stop - start) / (double)(num - 1);
    for (int i = 0; i < num; ++i) out[i] = (double)i * step + start;
    out[num - 1] = stop;
}

int main(int argc, char **argv)
{
    if (argc != 4) {
        fprintf(stderr, "usage: %s cloud.bin rows.bin radius\n", argv[0]);
        return 1;
    }
    const double radius = atof(argv[3]);
    const int n_bins = 5, nb3 = n_bins * n_bins * n_bins;
    FILE *f = fopen(argv[1], "rb");
    int64_t n = 0, m = 0;
    if (!f || fread(&n, 8, 1, f) != 1 || fread(&m, 8, 1, f) != 1 || n <= 0 || m <= 0) {
        fprintf(stderr, "cannot read %s\n", argv[1]);
        return 1;
    }
    double *xyz = malloc((size_t)n * 3 * sizeof(double)), *nrm = malloc((size_t)n * 3 * sizeof(double));
    int64_t *kp = malloc((size_t)m * sizeof(int64_t));
    double *queries = malloc((size_t)m * 3 * sizeof(double));
    double *fpfh = malloc((size_t)m * nb3 * sizeof(double)), *shot = malloc((size_t)m * SF_SHOT_LEN * sizeof(double));
    if (!xyz || !nrm || !kp || !queries || !fpfh || !shot) return 1;
    if (fread(xyz, sizeof(double), (size_t)n * 3, f) != (size_t)n * 3 || fread(nrm, sizeof(double), (size_t)n * 3, f) != (size_t)n * 3 ||
        fread(kp, sizeof(int64_t), (size_t)m, f) != (size_t)m) {
        fprintf(stderr, "short read\n");
        return 1;
    }
    fclose(f);
    for (int64_t i = 0; i < m; ++i)
        for (int a = 0; a < 3; ++a) queries[3 * i + a] = xyz[3 * kp[i] + a];

    sf_ctx *ctx = sf_create(0);
    if (!ctx) {
        fprintf(stderr, "sf_create: %s\n", sf_last_error());
        return 2;
    }
    printf("%s\n", sf_version());
    sf_cloud *cloud = sf_cloud_upload(ctx, xyz, nrm, n, SF_HOST);
    if (!cloud) { fprintf(stderr, "sf_cloud_upload: %s\n", sf_last_error()); return 2; }

    /* compute_fpfh_descriptor(kp, cloud, normals, radius, n_bins): lists of every point, SPFH table, weighted reduction */
    sf_nbrs *self = sf_radius_search_self(ctx, cloud, radius, 0, n);
    if (!self) { fprintf(stderr, "sf_radius_search_self: %s\n", sf_last_error()); return 2; }
    sf_spfh *table = sf_spfh_create_for_radius(ctx, cloud, n_bins, sf_nbrs_max_count(self), radius);
    if (!table) { fprintf(stderr, "sf_spfh_create_for_radius: %s\n", sf_last_error()); return 2; }
    double edges[3 * 6]; /* the edges np.histogramdd derives from bins = 5 and the ranges of fpfh.py:82-87 */
    linspace(-1.0, 1.0, n_bins + 1, edges);
    linspace(-1.0, 1.0, n_bins + 1, edges + 6);
    linspace(-acos(-1.0) / 2, acos(-1.0) / 2, n_bins + 1, edges + 12);
    CHECK(sf_spfh_compute(ctx, cloud, self, table, edges));
    CHECK(sf_fpfh(ctx, cloud, self, table, kp, m, fpfh, SF_HOST));
    sf_spfh_free(ctx, table);
    sf_nbrs_free(ctx, self);

    /* ShotMultiprocessor(normalize=True, min_neighborhood_size=10).compute_descriptor_single_scale(cloud, normals, keypoints, radius) */
    sf_nbrs *lists = sf_radius_search(ctx, cloud, queries, m, radius, SF_HOST);
    if (!lists) { fprintf(stderr, "sf_radius_search: %s\n", sf_last_error()); return 2; }
    CHECK(sf_shot_single_scale(ctx, cloud, lists, 1, 10, NULL, shot, SF_HOST));
    printf("%lld neighbours in %lld lists, longest %lld\n", (long long)sf_nbrs_total(lists), (long long)sf_nbrs_num_queries(lists),
           (long long)sf_nbrs_max_count(lists));
    sf_nbrs_free(ctx, lists);
    sf_cloud_free(ctx, cloud);
    sf_destroy(ctx);

    f = fopen(argv[2], "wb");
    if (!f || fwrite(fpfh, sizeof(double), (size_t)m * nb3, f) != (size_t)m * nb3 ||
        fwrite(shot, sizeof(double), (size_t)m * SF_SHOT_LEN, f) != (size_t)m * SF_SHOT_LEN) {
        fprintf(stderr, "cannot write %s\n", argv[2]);
        return 1;
    }
    fclose(f);
    free(xyz); free(nrm); free(kp); free(queries); free(fpfh); free(shot);
    return 0;
}
