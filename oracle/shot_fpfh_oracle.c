/*
 * oracle/shot_fpfh_oracle.c -- CPU restatement of the SHOT / FPFH hot path.
 *
 * THIS FILE IS TEST INFRASTRUCTURE.  It is the checker the HIP kernels are compared with.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it; the
 * product package (shot_fpfh_amd/) never imports, links or calls anything under oracle/.
 *
 * Parity status: PINNED.  Every function here is checked (tests/test_oracle_golden.py) against
 * golden vectors produced by importing the reference itself in the build container
 * (tools/gen_golden.py -> tests/golden/ *.npz), and orc_eigh3 is additionally checked against
 * numpy.linalg.eigh (LAPACK dsyevd) including eigenvector SIGNS on random matrices.
 *
 * Citations are file:line into the reference checkout (aubin-tchoi/shot-fpfh @ 2025-04-04).
 * Plain scalar C, double precision, no FMA contraction (compiled with -ffp-contract=off) so the
 * bin-deciding arithmetic is evaluated in the order NumPy evaluates it.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_PI 3.141592653589793 /* == numpy.pi */

/* ------------------------------------------------------------------------------------------
 * small helpers
 * ---------------------------------------------------------------------------------------- */
static inline double sq_dist3(const double *a, const double *b)
{
    /* sklearn KDTree euclidean rdist (sklearn/neighbors/_binary_tree.pxi.tp, query_radius
     * leaf loop) and numpy.linalg.norm(axis=1) (used at fpfh.py:48, shot.py:211): squares are
     * accumulated left to right, ((dx^2 + dy^2) + dz^2). */
    double dx = a[0] - b[0], dy = a[1] - b[1], dz = a[2] - b[2];
    return (dx * dx + dy * dy) + dz * dz;
}

static inline double sign_of(double x) { return (x > 0.0) - (x < 0.0); } /* numpy.sign */

static inline double clip(double x, double lo, double hi) { return x < lo ? lo : (x > hi ? hi : x); }

/* ------------------------------------------------------------------------------------------
 * (a1) radius search -- replaces sklearn.neighbors.KDTree(X).query_radius(Q, r)
 * call sites: fpfh.py:26-30, shot_parallelization.py:167-169, pca_based_descriptors.py:45-49
 * Inclusion rule: d2 <= r*r in float64, self-match included.  Lists are returned with
 * ASCENDING point index (the KDTree's own order is tree-traversal order; callers only depend
 * on the set).
 * A uniform grid (cell edge = r) makes the oracle usable as the bench's cpu_baseline.
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    double lo[3];
    double inv_cell;
    int64_t dim[3];
    int64_t *cell_start; /* dim0*dim1*dim2 + 1 */
    int32_t *order;      /* point indices grouped by cell, ascending inside a cell */
    int64_t n;
    const double *xyz;
} orc_grid;

static int64_t grid_coord(const orc_grid *g, double v, int axis)
{
    double t = floor((v - g->lo[axis]) * g->inv_cell);
    if (!(t >= 0.0)) t = 0.0; /* also catches NaN */
    if (t > (double)(g->dim[axis] - 1)) t = (double)(g->dim[axis] - 1);
    return (int64_t)t;
}

static int grid_build(orc_grid *g, const double *xyz, int64_t n, double cell)
{
    memset(g, 0, sizeof(*g));
    g->n = n;
    g->xyz = xyz;
    double hi[3] = {0, 0, 0};
    for (int a = 0; a < 3; ++a) {
        g->lo[a] = n ? xyz[a] : 0.0;
        hi[a] = g->lo[a];
    }
    for (int64_t i = 0; i < n; ++i)
        for (int a = 0; a < 3; ++a) {
            double v = xyz[3 * i + a];
            if (v < g->lo[a]) g->lo[a] = v;
            if (v > hi[a]) hi[a] = v;
        }
    /* keep the table bounded: never more than ~2^24 cells */
    double c = cell;
    for (;;) {
        double tot = 1.0;
        for (int a = 0; a < 3; ++a) {
            double d = floor((hi[a] - g->lo[a]) / c) + 1.0;
            if (!(d >= 1.0)) d = 1.0;
            g->dim[a] = (int64_t)d;
            tot *= d;
        }
        if (tot <= 16777216.0) break;
        c *= 2.0;
    }
    g->inv_cell = 1.0 / c;
    int64_t ncell = g->dim[0] * g->dim[1] * g->dim[2];
    g->cell_start = (int64_t *)calloc((size_t)ncell + 1, sizeof(int64_t));
    g->order = (int32_t *)malloc((size_t)(n ? n : 1) * sizeof(int32_t));
    if (!g->cell_start || !g->order) return -1;
    int64_t *cid = (int64_t *)malloc((size_t)(n ? n : 1) * sizeof(int64_t));
    if (!cid) return -1;
    for (int64_t i = 0; i < n; ++i) {
        int64_t cx = grid_coord(g, xyz[3 * i], 0), cy = grid_coord(g, xyz[3 * i + 1], 1),
                cz = grid_coord(g, xyz[3 * i + 2], 2);
        cid[i] = (cz * g->dim[1] + cy) * g->dim[0] + cx;
        g->cell_start[cid[i] + 1]++;
    }
    for (int64_t c2 = 0; c2 < ncell; ++c2) g->cell_start[c2 + 1] += g->cell_start[c2];
    int64_t *cur = (int64_t *)malloc((size_t)ncell * sizeof(int64_t));
    if (!cur) return -1;
    memcpy(cur, g->cell_start, (size_t)ncell * sizeof(int64_t));
    for (int64_t i = 0; i < n; ++i) g->order[cur[cid[i]]++] = (int32_t)i;
    free(cur);
    free(cid);
    return 0;
}

static void grid_free(orc_grid *g)
{
    free(g->cell_start);
    free(g->order);
}

static int cmp_i32(const void *a, const void *b)
{
    int32_t x = *(const int32_t *)a, y = *(const int32_t *)b;
    return (x > y) - (x < y);
}

/* Collect the neighbours of q into buf (capacity cap, grown by the caller when -needed is
 * returned).  Returns the count. */
static int64_t grid_query(const orc_grid *g, const double *q, double r, int32_t *buf, int64_t cap)
{
    double r2 = r * r;
    int64_t cnt = 0;
    double cell = 1.0 / g->inv_cell;
    int64_t lo[3], hi[3];
    for (int a = 0; a < 3; ++a) {
        double tl = floor((q[a] - r - g->lo[a]) * g->inv_cell) - 1.0; /* -1/+1: rounding slack */
        double th = floor((q[a] + r - g->lo[a]) * g->inv_cell) + 1.0;
        if (!(tl >= 0.0)) tl = 0.0;
        if (!(th <= (double)(g->dim[a] - 1))) th = (double)(g->dim[a] - 1);
        if (tl > (double)(g->dim[a] - 1)) tl = (double)(g->dim[a] - 1);
        if (th < 0.0) th = 0.0;
        lo[a] = (int64_t)tl;
        hi[a] = (int64_t)th;
    }
    (void)cell;
    for (int64_t cz = lo[2]; cz <= hi[2]; ++cz)
        for (int64_t cy = lo[1]; cy <= hi[1]; ++cy) {
            int64_t base = (cz * g->dim[1] + cy) * g->dim[0];
            int64_t s = g->cell_start[base + lo[0]], e = g->cell_start[base + hi[0] + 1];
            for (int64_t t = s; t < e; ++t) {
                int32_t j = g->order[t];
                if (sq_dist3(g->xyz + 3 * (int64_t)j, q) <= r2) {
                    if (cnt < cap) buf[cnt] = j;
                    ++cnt;
                }
            }
        }
    if (cnt <= cap) qsort(buf, (size_t)cnt, sizeof(int32_t), cmp_i32);
    return cnt;
}

/* CSR, two calls: first with idx == NULL fills offsets[m+1] and returns the total; second
 * fills idx (and dist when non-NULL; dist = sqrt(d2) as KDTree return_distance=True). */
int64_t orc_radius_search(const double *xyz, int64_t n, const double *q, int64_t m, double r,
                          int64_t *offsets, int32_t *idx, double *dist)
{
    orc_grid g;
    if (grid_build(&g, xyz, n, r > 0 ? r : 1.0)) return -1;
    int64_t cap = 1024;
    int32_t *buf = (int32_t *)malloc((size_t)cap * sizeof(int32_t));
    int64_t total = 0;
    if (!idx) offsets[0] = 0;
    for (int64_t i = 0; i < m; ++i) {
        int64_t c = grid_query(&g, q + 3 * i, r, buf, cap);
        if (c > cap) {
            cap = c;
            buf = (int32_t *)realloc(buf, (size_t)cap * sizeof(int32_t));
            c = grid_query(&g, q + 3 * i, r, buf, cap);
        }
        if (!idx) {
            offsets[i + 1] = offsets[i] + c;
        } else {
            memcpy(idx + offsets[i], buf, (size_t)c * sizeof(int32_t));
            if (dist)
                for (int64_t t = 0; t < c; ++t)
                    dist[offsets[i] + t] = sqrt(sq_dist3(xyz + 3 * (int64_t)buf[t], q + 3 * i));
        }
        total += c;
    }
    free(buf);
    grid_free(&g);
    return total;
}

/* Brute-force variant: the independent check of the grid above (tests only). */
int64_t orc_radius_search_brute(const double *xyz, int64_t n, const double *q, int64_t m, double r,
                                int64_t *offsets, int32_t *idx)
{
    double r2 = r * r;
    int64_t total = 0;
    if (!idx) offsets[0] = 0;
    for (int64_t i = 0; i < m; ++i) {
        int64_t c = 0;
        for (int64_t j = 0; j < n; ++j)
            if (sq_dist3(xyz + 3 * j, q + 3 * i) <= r2) {
                if (idx) idx[offsets[i] + c] = (int32_t)j;
                ++c;
            }
        if (!idx) offsets[i + 1] = offsets[i] + c;
        total += c;
    }
    return total;
}

/* ------------------------------------------------------------------------------------------
 * numpy.linalg.eigh for a 3x3 symmetric matrix = LAPACK dsyevd(jobz='V', uplo='L'):
 *   dsytrd/dsytd2 (Householder tridiagonalisation, lower) -> dstedc('I') which for n <= 25 is
 *   dsteqr('I') (implicit QL/QR) -> dormtr (apply Q) -> ascending eigenvalues.
 * Third-party dependency, not in /root/reference: numpy pinned 1.26.4 (poetry.lock:579-580),
 * bundled OpenBLAS/LAPACK >= 3.10 (new dlartg).  The published algorithm is restated here for
 * n = 3 because the reference's results depend on the SIGN LAPACK happens to return:
 * get_local_rf (shot.py:36-48) leaves an axis as LAPACK returned it whenever its vote is
 * within +-1 of a tie, and compute_normals (pca_based_descriptors.py:51) returns the raw
 * eigenvector when no pre_computed_normals are given.
 * Call sites: shot.py:36, pca_based_descriptors.py:24.
 * ---------------------------------------------------------------------------------------- */
#define LP_EPS 1.1102230246251565e-16   /* dlamch('E') */
#define LP_SAFMIN 2.2250738585072014e-308 /* dlamch('S') */

static double lp_sign(double a, double b) { return signbit(b) ? -fabs(a) : fabs(a); } /* Fortran SIGN */

static double lp_dlapy2(double x, double y)
{
    double xa = fabs(x), ya = fabs(y);
    double w = xa > ya ? xa : ya, z = xa > ya ? ya : xa;
    if (z == 0.0) return w;
    double t = z / w;
    return w * sqrt(1.0 + t * t);
}

static void lp_dlartg(double f, double g, double *c, double *s, double *r)
{
    /* LAPACK >= 3.10 la_lartg.f90 */
    const double safmin = LP_SAFMIN, safmax = 1.0 / LP_SAFMIN;
    const double rtmin = sqrt(safmin), rtmax = sqrt(safmax / 2.0);
    double f1 = fabs(f), g1 = fabs(g);
    if (g == 0.0) {
        *c = 1.0; *s = 0.0; *r = f;
    } else if (f == 0.0) {
        *c = 0.0; *s = lp_sign(1.0, g); *r = g1;
    } else if (f1 > rtmin && f1 < rtmax && g1 > rtmin && g1 < rtmax) {
        double d = sqrt(f * f + g * g);
        *c = f1 / d;
        *r = lp_sign(d, f);
        *s = g / *r;
    } else {
        double u = f1 > g1 ? f1 : g1;
        if (u < safmin) u = safmin;
        if (u > safmax) u = safmax;
        double fs = f / u, gs = g / u;
        double d = sqrt(fs * fs + gs * gs);
        *c = fabs(fs) / d;
        *r = lp_sign(d, f);
        *s = gs / *r;
        *r = *r * u;
    }
}

static void lp_dlaev2(double a, double b, double c, double *rt1, double *rt2, double *cs1, double *sn1)
{
    double sm = a + c, df = a - c, adf = fabs(df), tb = b + b, ab = fabs(tb);
    double acmx, acmn, rt;
    int sgn1, sgn2;
    if (fabs(a) > fabs(c)) { acmx = a; acmn = c; } else { acmx = c; acmn = a; }
    if (adf > ab) { double t = ab / adf; rt = adf * sqrt(1.0 + t * t); }
    else if (adf < ab) { double t = adf / ab; rt = ab * sqrt(1.0 + t * t); }
    else rt = ab * sqrt(2.0);
    if (sm < 0.0) {
        *rt1 = 0.5 * (sm - rt); sgn1 = -1;
        *rt2 = (acmx / *rt1) * acmn - (b / *rt1) * b;
    } else if (sm > 0.0) {
        *rt1 = 0.5 * (sm + rt); sgn1 = 1;
        *rt2 = (acmx / *rt1) * acmn - (b / *rt1) * b;
    } else {
        *rt1 = 0.5 * rt; *rt2 = -0.5 * rt; sgn1 = 1;
    }
    double cs;
    if (df >= 0.0) { cs = df + rt; sgn2 = 1; } else { cs = df - rt; sgn2 = -1; }
    if (fabs(cs) > ab) {
        double ct = -tb / cs;
        *sn1 = 1.0 / sqrt(1.0 + ct * ct);
        *cs1 = ct * *sn1;
    } else if (ab == 0.0) {
        *cs1 = 1.0; *sn1 = 0.0;
    } else {
        double tn = -cs / tb;
        *cs1 = 1.0 / sqrt(1.0 + tn * tn);
        *sn1 = tn * *cs1;
    }
    if (sgn1 == sgn2) {
        double tn = *cs1;
        *cs1 = -*sn1;
        *sn1 = tn;
    }
}

/* dlasr(side='R', pivot='V', direct) on the 3 x ncol block of z (column-major, ld 3)
 * starting at column col0; c/s hold ncol-1 rotations. */
static void lp_dlasr_rv(int forward, int ncol, const double *c, const double *s, double *z, int col0)
{
    for (int t = 0; t < ncol - 1; ++t) {
        int j = forward ? t : ncol - 2 - t;
        double ct = c[j], st = s[j];
        if (ct != 1.0 || st != 0.0)
            for (int i = 0; i < 3; ++i) {
                double *zj = z + 3 * (col0 + j) + i, *zj1 = z + 3 * (col0 + j + 1) + i;
                double temp = *zj1;
                *zj1 = ct * temp - st * *zj;
                *zj = st * temp + ct * *zj;
            }
    }
}

/* dsteqr(compz='I') for n = 3.  d[3], e[2] in/out; z column-major 3x3 out. 1-based indexing
 * is kept through macros so the control flow reads like the Fortran. */
static void lp_dsteqr3(double *d_, double *e_, double *z)
{
#define D(i) d_[(i) - 1]
#define E(i) e_[(i) - 1]
    const int n = 3, maxit = 30;
    const double eps = LP_EPS, eps2 = eps * eps, safmin = LP_SAFMIN;
    const double safmax = 1.0 / safmin, ssfmax = sqrt(safmax) / 3.0, ssfmin = sqrt(safmin) / eps2;
    double wc[2], ws[2];
    for (int i = 0; i < 9; ++i) z[i] = 0.0;
    z[0] = z[4] = z[8] = 1.0;
    int nmaxit = n * maxit, jtot = 0, l1 = 1, nm1 = n - 1;
    int l, m, lsv, lend, lendsv, iscale;
    double anorm, p, g, r, c, s, f, b, rt1, rt2;
    for (;;) { /* label 10 */
        if (l1 > n) break;
        if (l1 > 1) E(l1 - 1) = 0.0;
        m = n;
        if (l1 <= nm1)
            for (int mm = l1; mm <= nm1; ++mm) {
                double tst = fabs(E(mm));
                if (tst == 0.0) { m = mm; break; }
                if (tst <= (sqrt(fabs(D(mm))) * sqrt(fabs(D(mm + 1)))) * eps) { E(mm) = 0.0; m = mm; break; }
            }
        l = l1; lsv = l; lend = m; lendsv = lend; l1 = m + 1;
        if (lend == l) continue;
        /* scale submatrix */
        anorm = 0.0;
        for (int i = l; i <= lend; ++i) if (fabs(D(i)) > anorm) anorm = fabs(D(i));
        for (int i = l; i <= lend - 1; ++i) if (fabs(E(i)) > anorm) anorm = fabs(E(i));
        iscale = 0;
        if (anorm == 0.0) continue;
        if (anorm > ssfmax) {
            iscale = 1;
            for (int i = l; i <= lend; ++i) D(i) = D(i) / anorm * ssfmax;
            for (int i = l; i <= lend - 1; ++i) E(i) = E(i) / anorm * ssfmax;
        } else if (anorm < ssfmin) {
            iscale = 2;
            for (int i = l; i <= lend; ++i) D(i) = D(i) / anorm * ssfmin;
            for (int i = l; i <= lend - 1; ++i) E(i) = E(i) / anorm * ssfmin;
        }
        if (fabs(D(lend)) < fabs(D(l))) { lend = lsv; l = lendsv; }
        if (lend > l) {
            /* QL iteration */
            for (;;) { /* label 40 */
                m = lend;
                if (l != lend)
                    for (int mm = l; mm <= lend - 1; ++mm) {
                        double tst = fabs(E(mm)) * fabs(E(mm));
                        if (tst <= (eps2 * fabs(D(mm))) * fabs(D(mm + 1)) + safmin) { m = mm; break; }
                    }
                if (m < lend) E(m) = 0.0;
                p = D(l);
                if (m == l) { /* label 80: eigenvalue found */
                    D(l) = p;
                    l = l + 1;
                    if (l <= lend) continue;
                    break;
                }
                if (m == l + 1) {
                    lp_dlaev2(D(l), E(l), D(l + 1), &rt1, &rt2, &c, &s);
                    wc[0] = c; ws[0] = s;
                    lp_dlasr_rv(0, 2, wc, ws, z, l - 1);
                    D(l) = rt1; D(l + 1) = rt2; E(l) = 0.0;
                    l = l + 2;
                    if (l <= lend) continue;
                    break;
                }
                if (jtot == nmaxit) break;
                jtot++;
                g = (D(l + 1) - p) / (2.0 * E(l));
                r = lp_dlapy2(g, 1.0);
                g = D(m) - p + (E(l) / (g + lp_sign(r, g)));
                s = 1.0; c = 1.0; p = 0.0;
                for (int i = m - 1; i >= l; --i) {
                    f = s * E(i);
                    b = c * E(i);
                    lp_dlartg(g, f, &c, &s, &r);
                    if (i != m - 1) E(i + 1) = r;
                    g = D(i + 1) - p;
                    r = (D(i) - g) * s + 2.0 * c * b;
                    p = s * r;
                    D(i + 1) = g + p;
                    g = c * r - b;
                    wc[i - l] = c; ws[i - l] = -s;
                }
                lp_dlasr_rv(0, m - l + 1, wc, ws, z, l - 1);
                D(l) = D(l) - p;
                E(l) = g;
            }
        } else {
            /* QR iteration */
            for (;;) { /* label 90 */
                m = lend;
                if (l != lend)
                    for (int mm = l; mm >= lend + 1; --mm) {
                        double tst = fabs(E(mm - 1)) * fabs(E(mm - 1));
                        if (tst <= (eps2 * fabs(D(mm))) * fabs(D(mm - 1)) + safmin) { m = mm; break; }
                    }
                if (m > lend) E(m - 1) = 0.0;
                p = D(l);
                if (m == l) { /* label 130 */
                    D(l) = p;
                    l = l - 1;
                    if (l >= lend) continue;
                    break;
                }
                if (m == l - 1) {
                    lp_dlaev2(D(l - 1), E(l - 1), D(l), &rt1, &rt2, &c, &s);
                    wc[0] = c; ws[0] = s;
                    lp_dlasr_rv(1, 2, wc, ws, z, l - 2);
                    D(l - 1) = rt1; D(l) = rt2; E(l - 1) = 0.0;
                    l = l - 2;
                    if (l >= lend) continue;
                    break;
                }
                if (jtot == nmaxit) break;
                jtot++;
                g = (D(l - 1) - p) / (2.0 * E(l - 1));
                r = lp_dlapy2(g, 1.0);
                g = D(m) - p + (E(l - 1) / (g + lp_sign(r, g)));
                s = 1.0; c = 1.0; p = 0.0;
                for (int i = m; i <= l - 1; ++i) {
                    f = s * E(i);
                    b = c * E(i);
                    lp_dlartg(g, f, &c, &s, &r);
                    if (i != m) E(i - 1) = r;
                    g = D(i) - p;
                    r = (D(i + 1) - g) * s + 2.0 * c * b;
                    p = s * r;
                    D(i) = g + p;
                    g = c * r - b;
                    wc[i - m] = c; ws[i - m] = s;
                }
                lp_dlasr_rv(1, l - m + 1, wc, ws, z, m - 1);
                D(l) = D(l) - p;
                E(l - 1) = g;
            }
        }
        /* label 140: undo scaling */
        if (iscale == 1) {
            for (int i = lsv; i <= lendsv; ++i) D(i) = D(i) / ssfmax * anorm;
            for (int i = lsv; i <= lendsv - 1; ++i) E(i) = E(i) / ssfmax * anorm;
        } else if (iscale == 2) {
            for (int i = lsv; i <= lendsv; ++i) D(i) = D(i) / ssfmin * anorm;
            for (int i = lsv; i <= lendsv - 1; ++i) E(i) = E(i) / ssfmin * anorm;
        }
        if (jtot >= nmaxit) break;
    }
    /* selection sort, ascending, swapping eigenvector columns */
    for (int ii = 2; ii <= n; ++ii) {
        int i = ii - 1, k = i;
        p = D(i);
        for (int j = ii; j <= n; ++j)
            if (D(j) < p) { k = j; p = D(j); }
        if (k != i) {
            D(k) = D(i);
            D(i) = p;
            for (int t = 0; t < 3; ++t) {
                double tmp = z[3 * (i - 1) + t];
                z[3 * (i - 1) + t] = z[3 * (k - 1) + t];
                z[3 * (k - 1) + t] = tmp;
            }
        }
    }
#undef D
#undef E
}

/* a: row-major 3x3, only the LOWER triangle is read (numpy UPLO='L').  w[3] ascending,
 * v row-major with v[3*i + k] = component i of eigenvector k (== numpy's v[i, k]). */
void orc_eigh3(const double *a, double *w, double *v)
{
    double a11 = a[0], a21 = a[3], a31 = a[6], a22 = a[4], a32 = a[7], a33 = a[8];
    double d[3], e[2], tau = 0.0, v2 = 0.0;
    /* dsytd2, uplo = 'L', i = 1: dlarfg(2, a21, a31) */
    double xnorm = fabs(a31);
    if (xnorm == 0.0) {
        tau = 0.0;
        e[0] = a21;
    } else {
        double beta = -lp_sign(lp_dlapy2(a21, xnorm), a21);
        tau = (beta - a21) / beta;
        v2 = a31 * (1.0 / (a21 - beta));
        e[0] = beta;
        /* x := tau * A22 * v, A22 = [[a22, a32],[a32, a33]], v = (1, v2).  Reference dsymv
         * (lower) column sweep: j=1: temp1 = tau*1; y1 += temp1*a22; y2 += temp1*a32;
         * temp2 = a32*v2; y1 += tau*temp2.  j=2: temp1 = tau*v2; y2 += temp1*a33. */
        double x1 = (tau * 1.0) * a22 + tau * (a32 * v2);
        double x2 = (tau * 1.0) * a32 + (tau * v2) * a33;
        double alpha = -0.5 * tau * (x1 * 1.0 + x2 * v2);
        double w1 = x1 + alpha * 1.0, w2 = x2 + alpha * v2;
        /* dsyr2 lower, alpha=-1, x=v, y=w */
        a22 = a22 + (1.0 * (-w1) + w1 * (-1.0));
        a32 = a32 + (v2 * (-w1) + w2 * (-1.0));
        a33 = a33 + (v2 * (-w2) + w2 * (-v2));
    }
    d[0] = a11;
    /* i = 2: dlarfg(1, ...) -> tau2 = 0 */
    e[1] = a32;
    d[1] = a22;
    d[2] = a33;
    double z[9]; /* column-major */
    lp_dsteqr3(d, e, z);
    /* dormtr('L','L','N'): rows 2..3 of Z := (I - tau v v^T) rows 2..3 */
    if (tau != 0.0)
        for (int j = 0; j < 3; ++j) {
            double wj = z[3 * j + 1] * 1.0 + z[3 * j + 2] * v2;
            double t = -tau * wj;
            z[3 * j + 1] = z[3 * j + 1] + 1.0 * t;
            z[3 * j + 2] = z[3 * j + 2] + v2 * t;
        }
    for (int k = 0; k < 3; ++k) {
        w[k] = d[k];
        for (int i = 0; i < 3; ++i) v[3 * i + k] = z[3 * k + i];
    }
}

/* ------------------------------------------------------------------------------------------
 * (a2) compute_normals, radius branch and k-NN branch  (pca_based_descriptors.py:15-59)
 * nbr lists are given as CSR (offsets/idx) so the same code serves both branches.
 * ---------------------------------------------------------------------------------------- */
void orc_normals_from_lists(const double *xyz, const int64_t *offsets, const int32_t *idx, int64_t m,
                            const double *pre /* nullable m x 3 */, double *out /* m x 3 */)
{
    for (int64_t i = 0; i < m; ++i) {
        int64_t s = offsets[i], k = offsets[i + 1] - s;
        double mean[3] = {0, 0, 0}, cov[9] = {0};
        /* pca(): barycenter = points.mean(axis=0) (15-21) */
        for (int64_t t = 0; t < k; ++t)
            for (int a = 0; a < 3; ++a) mean[a] += xyz[3 * (int64_t)idx[s + t] + a];
        for (int a = 0; a < 3; ++a) mean[a] /= (double)k;
        /* cov = centered.T @ centered / k (22-23); eigh reads the lower triangle */
        for (int64_t t = 0; t < k; ++t) {
            double c[3];
            for (int a = 0; a < 3; ++a) c[a] = xyz[3 * (int64_t)idx[s + t] + a] - mean[a];
            for (int a = 0; a < 3; ++a)
                for (int b = 0; b <= a; ++b) cov[3 * a + b] += c[a] * c[b];
        }
        for (int a = 0; a < 3; ++a)
            for (int b = 0; b <= a; ++b) cov[3 * a + b] /= (double)k;
        double w[3], v[9];
        orc_eigh3(cov, w, v);
        double nrm[3] = {v[0], v[3], v[6]}; /* eigenvectors[:, 0] (51) */
        if (pre) {
            double dot = (nrm[0] * pre[3 * i] + nrm[1] * pre[3 * i + 1]) + nrm[2] * pre[3 * i + 2];
            if (dot < 0.0) /* 53-57 */
                for (int a = 0; a < 3; ++a) nrm[a] = -nrm[a];
        }
        for (int a = 0; a < 3; ++a) out[3 * i + a] = nrm[a];
    }
}

/* pca() for every list with the whole decomposition kept, plus the moments of
 * compute_local_pca_with_moments (pca_based_descriptors.py:15-26, 115-144).  evals m x 3 ascending,
 * evecs m x 9 row-major as numpy.linalg.eigh returns them, moments (nullable) m x 8. */
void orc_pca_from_lists(const double *xyz, const int64_t *offsets, const int32_t *idx, int64_t m,
                        double *evals, double *evecs, double *moments)
{
    for (int64_t i = 0; i < m; ++i) {
        int64_t s = offsets[i], k = offsets[i + 1] - s;
        double mean[3] = {0, 0, 0}, cov[9] = {0};
        for (int64_t t = 0; t < k; ++t)
            for (int a = 0; a < 3; ++a) mean[a] += xyz[3 * (int64_t)idx[s + t] + a];
        for (int a = 0; a < 3; ++a) mean[a] /= (double)k;
        for (int64_t t = 0; t < k; ++t) {
            double c[3];
            for (int a = 0; a < 3; ++a) c[a] = xyz[3 * (int64_t)idx[s + t] + a] - mean[a];
            for (int a = 0; a < 3; ++a)
                for (int b = 0; b <= a; ++b) cov[3 * a + b] += c[a] * c[b];
        }
        for (int a = 0; a < 3; ++a)
            for (int b = 0; b <= a; ++b) cov[3 * a + b] /= (double)k;
        double *w = evals + 3 * i, *v = evecs + 9 * i;
        orc_eigh3(cov, w, v);
        if (!moments) continue;
        /* moment = centered @ eigenvectors.T (124): component a uses ROW a of v; vert_moment = centered[:, 2] */
        double sm[3] = {0, 0, 0}, sq[3] = {0, 0, 0}, vz = 0, vz2 = 0;
        for (int64_t t = 0; t < k; ++t) {
            double c[3];
            for (int a = 0; a < 3; ++a) c[a] = xyz[3 * (int64_t)idx[s + t] + a] - mean[a];
            for (int a = 0; a < 3; ++a) {
                double u = (c[0] * v[3 * a] + c[1] * v[3 * a + 1]) + c[2] * v[3 * a + 2];
                sm[a] += u;
                sq[a] += u * u;
            }
            vz += c[2];
            vz2 += c[2] * c[2];
        }
        double *o = moments + 8 * i;
        for (int a = 0; a < 3; ++a) { o[a] = fabs(sm[a] / (double)k); o[3 + a] = sq[a] / (double)k; }
        o[6] = vz / (double)k;
        o[7] = vz2 / (double)k;
    }
}

int orc_normals_radius(const double *xyz, int64_t n, const double *q, int64_t m, double radius,
                       const double *pre, double *out)
{
    int64_t *off = (int64_t *)malloc((size_t)(m + 1) * sizeof(int64_t));
    int64_t total = orc_radius_search(xyz, n, q, m, radius, off, NULL, NULL);
    if (total < 0) return -1;
    int32_t *idx = (int32_t *)malloc((size_t)(total ? total : 1) * sizeof(int32_t));
    orc_radius_search(xyz, n, q, m, radius, off, idx, NULL);
    orc_normals_from_lists(xyz, off, idx, m, pre, out);
    free(idx);
    free(off);
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * (a3) get_local_rf  (shot.py:16-48), query INCLUDED in the support as ShotMultiprocessor
 * passes it (shot_parallelization.py:70-75).
 * ---------------------------------------------------------------------------------------- */
void orc_lrf_single(const double *point, const double *xyz, const int32_t *idx, int64_t k, double radius,
                    double *lrf /* row-major 3x3, columns = x, y, z axes */)
{
    if (k == 0) { /* 24-25 */
        for (int i = 0; i < 9; ++i) lrf[i] = (i % 4 == 0) ? 1.0 : 0.0;
        return;
    }
    double cov[9] = {0}, wsum = 0.0;
    for (int64_t t = 0; t < k; ++t) {
        const double *p = xyz + 3 * (int64_t)idx[t];
        double c[3] = {p[0] - point[0], p[1] - point[1], p[2] - point[2]};
        double wgt = radius - sqrt((c[0] * c[0] + c[1] * c[1]) + c[2] * c[2]); /* 30 */
        wsum += wgt;
        for (int a = 0; a < 3; ++a)
            for (int b = 0; b <= a; ++b) cov[3 * a + b] += c[a] * (c[b] * wgt); /* 31-34 */
    }
    for (int a = 0; a < 3; ++a)
        for (int b = 0; b <= a; ++b) cov[3 * a + b] /= wsum; /* 34 */
    double w[3], v[9];
    orc_eigh3(cov, w, v); /* 36 */
    double x[3] = {v[2], v[5], v[8]}, z[3] = {v[0], v[3], v[6]};
    int64_t xneg = 0, xpos = 0, zneg = 0, zpos = 0;
    for (int64_t t = 0; t < k; ++t) { /* 40-45 */
        const double *p = xyz + 3 * (int64_t)idx[t];
        double c[3] = {p[0] - point[0], p[1] - point[1], p[2] - point[2]};
        double xo = (c[0] * x[0] + c[1] * x[1]) + c[2] * x[2];
        double zo = (c[0] * z[0] + c[1] * z[1]) + c[2] * z[2];
        if (xo < 0.0) ++xneg; else if (xo >= 0.0) ++xpos;
        if (zo < 0.0) ++zneg; else if (zo >= 0.0) ++zpos;
    }
    if (xneg > xpos) for (int a = 0; a < 3; ++a) x[a] = -x[a];
    if (zneg > zpos) for (int a = 0; a < 3; ++a) z[a] = -z[a];
    /* y = cross(z, x) (46) */
    double y[3] = {z[1] * x[2] - z[2] * x[1], z[2] * x[0] - z[0] * x[2], z[0] * x[1] - z[1] * x[0]};
    for (int a = 0; a < 3; ++a) { /* np.flip(axis=1): columns [x y z] (48) */
        lrf[3 * a + 0] = x[a];
        lrf[3 * a + 1] = y[a];
        lrf[3 * a + 2] = z[a];
    }
}

/* ------------------------------------------------------------------------------------------
 * (a4) get_azimuth_idx  (shot.py:51-70)
 * ---------------------------------------------------------------------------------------- */
static int azimuth_idx(double x, double y)
{
    int a = (y > 0.0) || ((y == 0.0) && (x < 0.0));
    int bsel = ((x > 0.0) || ((x == 0.0) && (y > 0.0))) != a;
    int csel = ((x * y > 0.0) || (x == 0.0)) ? (fabs(x) < fabs(y)) : (fabs(x) > fabs(y));
    return 4 * a + 2 * bsel + csel;
}

int orc_azimuth_idx(double x, double y) { return azimuth_idx(x, y); } /* exported for the boundary-table test */

typedef struct {
    double rho;
    int64_t pos;
} rho_key;

static int cmp_rho(const void *a, const void *b)
{
    const rho_key *x = (const rho_key *)a, *y = (const rho_key *)b;
    if (x->rho < y->rho) return -1;
    if (x->rho > y->rho) return 1;
    return (x->pos > y->pos) - (x->pos < y->pos);
}

/* One "descriptor[idx] += val" statement of shot.py:244-298 for every neighbour at once:
 * NumPy gathers descriptor[idx], adds val, and scatters back, so with duplicate idx only the
 * LAST neighbour (ascending rho) lands.  new[b] = old[b] + val[last writer of b]. */
static void fancy_add(double *desc, const int *bins, const double *vals, int64_t k, double *scratch)
{
    memcpy(scratch, desc, 352 * sizeof(double));
    for (int64_t t = 0; t < k; ++t) desc[bins[t]] = scratch[bins[t]] + vals[t];
}

/* (a7) compute_single_shot_descriptor  (shot.py:175-306) */
void orc_shot_single(const double *point, const double *xyz, const double *normals, const int32_t *idx,
                     int64_t k_all, double radius, const double *lrf, int normalize, int64_t min_nb,
                     double *out /* 352 */)
{
    for (int i = 0; i < 352; ++i) out[i] = 0.0;
    rho_key *keys = (rho_key *)malloc((size_t)(k_all ? k_all : 1) * sizeof(rho_key));
    int64_t k = 0;
    for (int64_t t = 0; t < k_all; ++t) {
        double rho = sqrt(sq_dist3(xyz + 3 * (int64_t)idx[t], point)); /* 211 */
        if (rho > 0.0) { keys[k].rho = rho; keys[k].pos = t; ++k; }
    }
    if (!(k > min_nb)) { free(keys); return; } /* 212, 306 */
    qsort(keys, (size_t)k, sizeof(rho_key), cmp_rho); /* 218-221 (ties: list order) */

    int *b_base = (int *)malloc((size_t)k * 7 * sizeof(int));
    int *b_cos = b_base + k, *b_r1 = b_base + 2 * k, *b_r0 = b_base + 3 * k, *b_p1 = b_base + 4 * k,
        *b_p0 = b_base + 5 * k, *b_th = b_base + 6 * k;
    double *vals = (double *)malloc((size_t)k * 10 * sizeof(double));
    double *v1 = vals, *v2 = vals + k, *v3 = vals + 2 * k, *v4 = vals + 3 * k, *v5 = vals + 4 * k,
           *v6 = vals + 5 * k, *v7 = vals + 6 * k, *v8 = vals + 7 * k, *v9 = vals + 8 * k, *v10 = vals + 9 * k;
    const double half_r = radius / 2, q1 = radius / 4, q3 = radius * 3 / 4;
    const double hpi = ORC_PI / 2, pi34 = ORC_PI * 3 / 4, pi4 = ORC_PI / 4;
    const double tsz = 2 * ORC_PI / 8;
    for (int64_t t = 0; t < k; ++t) {
        const double *p = xyz + 3 * (int64_t)idx[keys[t].pos];
        const double *nn = normals + 3 * (int64_t)idx[keys[t].pos];
        double rho = keys[t].rho;
        double c[3] = {p[0] - point[0], p[1] - point[1], p[2] - point[2]};
        /* local = (neighbors - point) @ eigenvectors (214) */
        double lx = (c[0] * lrf[0] + c[1] * lrf[3]) + c[2] * lrf[6];
        double ly = (c[0] * lrf[1] + c[1] * lrf[4]) + c[2] * lrf[7];
        double lz = (c[0] * lrf[2] + c[1] * lrf[5]) + c[2] * lrf[8];
        double cosine = clip((nn[0] * lrf[2] + nn[1] * lrf[5]) + nn[2] * lrf[8], -1.0, 1.0); /* 215 */
        double theta = atan2(ly, lx);                 /* 224 */
        double phi = acos(clip(lz / rho, -1.0, 1.0)); /* 225 */
        double cpos = (cosine + 1.0) * 11 / 2.0 - 0.5; /* 228 */
        double cidx_f = nearbyint(cpos);               /* np.rint: half to even (229) */
        int ci = (int)cidx_f;
        int ti = azimuth_idx(lx, ly); /* 230-232 */
        int pi_ = lz > 0.0;           /* 234 */
        int ri = rho > half_r;        /* 235 */
        double dc = cpos - cidx_f, sc = sign_of(dc), adc = sc * dc; /* 238-242 */
        int ci_n = (int)(cidx_f + sc) % 11;
        if (ci_n < 0) ci_n += 11; /* python % (245) */
        int base = ((ci * 8 + ti) * 2 + pi_) * 2 + ri;
        b_base[t] = base;
        b_cos[t] = ((ci_n * 8 + ti) * 2 + pi_) * 2 + ri;
        v1[t] = adc * (double)((cidx_f > -0.5) && (cidx_f < 11 - 0.5)); /* 249-251 */
        v2[t] = 1 - adc;                                                 /* 252-254 */
        /* interpolate_on_adjacent_husks (73-118) */
        double inner = (double)((rho > half_r) && (rho < q3)) * (q3 - rho) / half_r;
        double outer = (double)((rho < half_r) && (rho > q1)) * (rho - q1) / half_r;
        double cur = (double)(rho < half_r) * (1 - fabs(rho - q1) / half_r) +
                     (double)(rho > half_r) * (1 - fabs(rho - q3) / half_r);
        b_r1[t] = ((ci * 8 + ti) * 2 + pi_) * 2 + 1;
        b_r0[t] = ((ci * 8 + ti) * 2 + pi_) * 2 + 0;
        v3[t] = outer * (double)(ri == 0); /* 258-260 */
        v4[t] = inner * (double)(ri == 1); /* 261-263 */
        v5[t] = cur;                       /* 264 */
        /* interpolate_vertical_volumes (121-171) */
        double upper = (double)(((phi > hpi) || ((fabs(phi - hpi) < 1e-10) && (lz <= 0.0))) && (phi <= pi34)) *
                       (pi34 - phi) / hpi;
        double lower = (double)(((phi < hpi) && ((fabs(phi - hpi) >= 1e-10) || (lz > 0.0))) && (phi >= pi4)) *
                       (phi - pi4) / hpi;
        double curv = (double)(phi < hpi) * (1 - fabs(phi - pi4) / hpi) +
                      (double)(phi >= hpi) * (1 - fabs(phi - pi34) / hpi);
        b_p1[t] = ((ci * 8 + ti) * 2 + 1) * 2 + ri;
        b_p0[t] = ((ci * 8 + ti) * 2 + 0) * 2 + ri;
        v6[t] = upper * (double)(pi_ == 0); /* 270-272 */
        v7[t] = lower * (double)(pi_ == 1); /* 273-275 */
        v8[t] = curv;                       /* 276-278 */
        /* azimuth interpolation (282-298) */
        double dth = clip((theta - (-ORC_PI + ti * tsz)) / tsz - 0.5, -0.5, 0.5);
        double sth = sign_of(dth), adth = sth * dth;
        int ti_n = (int)((double)ti + sth) % 8;
        if (ti_n < 0) ti_n += 8;
        b_th[t] = ((ci * 8 + ti_n) * 2 + pi_) * 2 + ri;
        v9[t] = adth;
        v10[t] = 1 - adth;
    }
    double desc[352] = {0}, scratch[352];
    fancy_add(desc, b_cos, v1, k, scratch);
    fancy_add(desc, b_base, v2, k, scratch);
    fancy_add(desc, b_r1, v3, k, scratch);
    fancy_add(desc, b_r0, v4, k, scratch);
    fancy_add(desc, b_base, v5, k, scratch);
    fancy_add(desc, b_p1, v6, k, scratch);
    fancy_add(desc, b_p0, v7, k, scratch);
    fancy_add(desc, b_base, v8, k, scratch);
    fancy_add(desc, b_th, v9, k, scratch);
    fancy_add(desc, b_base, v10, k, scratch);
    double nrm = 0.0;
    for (int i = 0; i < 352; ++i) nrm += desc[i] * desc[i];
    nrm = sqrt(nrm);
    if (nrm > 0.0) /* 301-305 */
        for (int i = 0; i < 352; ++i) out[i] = normalize ? desc[i] / nrm : desc[i];
    free(vals);
    free(b_base);
    free(keys);
}

/* Drivers over many keypoints: ShotMultiprocessor.compute_local_rf / compute_descriptor
 * (shot_parallelization.py:46-133) -- the Pool is irrelevant to the values. */
int orc_shot_lrf(const double *xyz, int64_t n, const double *q, int64_t m, double radius, double *lrf)
{
    int64_t *off = (int64_t *)malloc((size_t)(m + 1) * sizeof(int64_t));
    int64_t total = orc_radius_search(xyz, n, q, m, radius, off, NULL, NULL);
    if (total < 0) return -1;
    int32_t *idx = (int32_t *)malloc((size_t)(total ? total : 1) * sizeof(int32_t));
    orc_radius_search(xyz, n, q, m, radius, off, idx, NULL);
    for (int64_t i = 0; i < m; ++i)
        orc_lrf_single(q + 3 * i, xyz, idx + off[i], off[i + 1] - off[i], radius, lrf + 9 * i);
    free(idx);
    free(off);
    return 0;
}

int orc_shot(const double *xyz, const double *normals, int64_t n, const double *q, int64_t m, double radius,
             const double *lrf, int normalize, int64_t min_nb, double *out)
{
    int64_t *off = (int64_t *)malloc((size_t)(m + 1) * sizeof(int64_t));
    int64_t total = orc_radius_search(xyz, n, q, m, radius, off, NULL, NULL);
    if (total < 0) return -1;
    int32_t *idx = (int32_t *)malloc((size_t)(total ? total : 1) * sizeof(int32_t));
    orc_radius_search(xyz, n, q, m, radius, off, idx, NULL);
    for (int64_t i = 0; i < m; ++i)
        orc_shot_single(q + 3 * i, xyz, normals, idx + off[i], off[i + 1] - off[i], radius, lrf + 9 * i,
                        normalize, min_nb, out + 352 * i);
    free(idx);
    free(off);
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * (a8) compute_shot_descriptor, the serial / debug variant (shot.py:310-499).
 * Differences from the ShotMultiprocessor path: the gate comes first (:360), the neighbours at
 * distance zero are dropped BEFORE the frame is computed (:361-363: neither their weight
 * `radius` in the covariance normaliser nor their ">= 0" sign vote exists), and the row is
 * always divided by its norm (:496-497).  The ten statements are those of
 * compute_single_shot_descriptor (same text at :376-494), so orc_shot_single serves both.
 * ---------------------------------------------------------------------------------------- */
int orc_shot_serial(const double *xyz, const double *normals, int64_t n, const double *q, int64_t m, double radius,
                    int64_t min_nb, double *out)
{
    int64_t *off = (int64_t *)malloc((size_t)(m + 1) * sizeof(int64_t));
    int64_t total = orc_radius_search(xyz, n, q, m, radius, off, NULL, NULL); /* 340-341 */
    if (total < 0) return -1;
    int32_t *idx = (int32_t *)malloc((size_t)(total ? total : 1) * sizeof(int32_t));
    int32_t *pos = (int32_t *)malloc((size_t)(total ? total : 1) * sizeof(int32_t));
    orc_radius_search(xyz, n, q, m, radius, off, idx, NULL);
    for (int64_t i = 0; i < m; ++i) {
        const double *point = q + 3 * i;
        double *row = out + 352 * i;
        int64_t k = off[i + 1] - off[i], kp = 0;
        for (int64_t t = 0; t < k; ++t) { /* distances > 0 (:359-361) */
            int32_t j = idx[off[i] + t];
            if (sqrt(sq_dist3(xyz + 3 * (int64_t)j, point)) > 0.0) pos[kp++] = j;
        }
        for (int b = 0; b < 352; ++b) row[b] = 0.0; /* all_descriptors = np.zeros (:343-348) */
        if (!(kp > min_nb)) continue;              /* :360 */
        double lrf[9];
        orc_lrf_single(point, xyz, pos, kp, radius, lrf); /* :362 */
        orc_shot_single(point, xyz, normals, pos, kp, radius, lrf, 1, min_nb, row);
    }
    free(pos);
    free(idx);
    free(off);
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * (a10) compute_fpfh_descriptor  (fpfh.py:16-117), decorrelated=False.
 * edges: 3 x (n_bins+1) doubles = np.linspace(lo, hi, n_bins+1) for (-1,1), (-1,1),
 * (-pi/2, pi/2), exactly what np.histogramdd builds (fpfh.py:82-87).
 * ---------------------------------------------------------------------------------------- */
static int hist_bin(const double *edges, int n_bins, double x)
{
    /* np.histogramdd: searchsorted(edges, x, 'right') - 1, x == last edge -> last bin,
     * anything outside (or NaN) is dropped. */
    if (!(x >= edges[0]) || x > edges[n_bins]) return -1;
    if (x == edges[n_bins]) return n_bins - 1;
    int lo = 0, hi = n_bins + 1; /* first index with edges[i] > x */
    while (lo < hi) {
        int mid = (lo + hi) / 2;
        if (edges[mid] <= x) lo = mid + 1; else hi = mid;
    }
    return lo - 1;
}

/* SPFH row of cloud point i from its neighbour list (fpfh.py:44-90): pairs at distance zero skipped,
 * out-of-range samples dropped by the histogram, counts divided by the FULL list length. */
static void spfh_row(const double *xyz, const double *normals, int64_t i, const int32_t *idx, int64_t k, int n_bins,
                     const double *edges, double *row /* n_bins^3, zeroed here */)
{
    const int64_t nb3 = (int64_t)n_bins * n_bins * n_bins;
    const double *ea = edges, *ep = edges + (n_bins + 1), *et = edges + 2 * (n_bins + 1);
    for (int64_t b = 0; b < nb3; ++b) row[b] = 0.0;
    if (k == 0) return;
    const double *pi_ = xyz + 3 * i, *u = normals + 3 * i;
    for (int64_t t = 0; t < k; ++t) {
        int64_t j = idx[t];
        const double *pj = xyz + 3 * j, *nj = normals + 3 * j;
        double c[3] = {pj[0] - pi_[0], pj[1] - pi_[1], pj[2] - pi_[2]};
        double dist = sqrt((c[0] * c[0] + c[1] * c[1]) + c[2] * c[2]); /* 48 */
        if (!(dist > 0.0)) continue;
        /* v = cross(c, u) (50); w = cross(u, v) (51) */
        double v[3] = {c[1] * u[2] - c[2] * u[1], c[2] * u[0] - c[0] * u[2], c[0] * u[1] - c[1] * u[0]};
        double w[3] = {u[1] * v[2] - u[2] * v[1], u[2] * v[0] - u[0] * v[2], u[0] * v[1] - u[1] * v[0]};
        double alpha = (v[0] * nj[0] + v[1] * nj[1]) + v[2] * nj[2];          /* 52 */
        double phi = ((c[0] * u[0] + c[1] * u[1]) + c[2] * u[2]) / dist;      /* 53 */
        double theta = atan2((nj[0] * w[0] + nj[1] * w[1]) + nj[2] * w[2],    /* 54-57 */
                             (nj[0] * u[0] + nj[1] * u[1]) + nj[2] * u[2]);
        int ba = hist_bin(ea, n_bins, alpha), bp = hist_bin(ep, n_bins, phi), bt = hist_bin(et, n_bins, theta);
        if (ba < 0 || bp < 0 || bt < 0) continue;
        row[((int64_t)ba * n_bins + bp) * n_bins + bt] += 1.0;
    }
    for (int64_t b = 0; b < nb3; ++b) row[b] = row[b] / (double)k; /* 88 */
}

/* FPFH row of keypoint i (fpfh.py:101-116) given a function-like access to SPFH rows: spfh_of[j] is the row of
 * cloud point j. */
static void fpfh_row(const double *xyz, int64_t i, const int32_t *idx, int64_t k, int64_t nb3, const double *own,
                     const double *const *rows /* rows[t] = SPFH row of idx[t] */, double *o)
{
    for (int64_t b = 0; b < nb3; ++b) o[b] = 0.0;
    for (int64_t t = 0; t < k; ++t) {
        int64_t j = idx[t];
        double dist = sqrt(sq_dist3(xyz + 3 * j, xyz + 3 * i));
        if (!(dist > 0.0)) continue;
        const double *rj = rows[t];
        for (int64_t b = 0; b < nb3; ++b) o[b] += rj[b] / dist;
    }
    for (int64_t b = 0; b < nb3; ++b) o[b] = own[b] + o[b] / (double)k;
}

int orc_fpfh(const double *xyz, const double *normals, int64_t n, const int64_t *kp_idx, int64_t m,
             double radius, int n_bins, const double *edges, double *out /* m x n_bins^3 */,
             double *spfh_out /* nullable n x n_bins^3 */)
{
    int64_t nb3 = (int64_t)n_bins * n_bins * n_bins;
    int64_t *off = (int64_t *)malloc((size_t)(n + 1) * sizeof(int64_t));
    int64_t total = orc_radius_search(xyz, n, xyz, n, radius, off, NULL, NULL); /* 26-30 */
    if (total < 0) return -1;
    int32_t *idx = (int32_t *)malloc((size_t)(total ? total : 1) * sizeof(int32_t));
    orc_radius_search(xyz, n, xyz, n, radius, off, idx, NULL);
    double *spfh = spfh_out ? spfh_out : (double *)malloc((size_t)(n * nb3 ? n * nb3 : 1) * sizeof(double));
    for (int64_t i = 0; i < n; ++i) /* 38-90 */
        spfh_row(xyz, normals, i, idx + off[i], off[i + 1] - off[i], n_bins, edges, spfh + i * nb3);
    int64_t kmax = 1;
    for (int64_t i = 0; i < n; ++i)
        if (off[i + 1] - off[i] > kmax) kmax = off[i + 1] - off[i];
    const double **rows = (const double **)malloc((size_t)kmax * sizeof(double *));
    for (int64_t q = 0; q < m; ++q) { /* 101-116 */
        int64_t i = kp_idx[q], s = off[i], k = off[i + 1] - s;
        for (int64_t t = 0; t < k; ++t) rows[t] = spfh + (int64_t)idx[s + t] * nb3;
        fpfh_row(xyz, i, idx + s, k, nb3, spfh + i * nb3, rows, out + q * nb3);
    }
    free(rows);
    if (!spfh_out) free(spfh);
    free(idx);
    free(off);
    return 0;
}

/* The same function for a SAMPLE of keypoints of a large cloud: identical arithmetic, but the SPFH rows are
 * evaluated only for the points the sample needs (the keypoints and their neighbours) instead of for all n -- what
 * makes a full-size check of a 1M / 8M-point cloud affordable.  Results are bit-identical to orc_fpfh's. */
int orc_fpfh_sample(const double *xyz, const double *normals, int64_t n, const int64_t *kp_idx, int64_t m,
                    double radius, int n_bins, const double *edges, double *out /* m x n_bins^3 */)
{
    const int64_t nb3 = (int64_t)n_bins * n_bins * n_bins;
    orc_grid g;
    if (grid_build(&g, xyz, n, radius > 0 ? radius : 1.0)) return -1;
    int64_t *slot = (int64_t *)malloc((size_t)(n ? n : 1) * sizeof(int64_t)); /* point -> cached row, -1 = none */
    for (int64_t i = 0; i < n; ++i) slot[i] = -1;
    int64_t cap = 1024, cap2 = 1024, nrows = 0, rows_cap = 4096;
    int32_t *buf = (int32_t *)malloc((size_t)cap * sizeof(int32_t));
    int32_t *buf2 = (int32_t *)malloc((size_t)cap2 * sizeof(int32_t));
    double *cache = (double *)malloc((size_t)rows_cap * nb3 * sizeof(double));
    const double **rows = (const double **)malloc((size_t)cap * sizeof(double *));
    int64_t *need = (int64_t *)malloc((size_t)(cap + 1) * sizeof(int64_t));
    for (int64_t q = 0; q < m; ++q) {
        const int64_t i = kp_idx[q];
        int64_t k = grid_query(&g, xyz + 3 * i, radius, buf, cap);
        if (k > cap) {
            cap = k;
            buf = (int32_t *)realloc(buf, (size_t)cap * sizeof(int32_t));
            rows = (const double **)realloc(rows, (size_t)cap * sizeof(double *));
            need = (int64_t *)realloc(need, (size_t)(cap + 1) * sizeof(int64_t));
            k = grid_query(&g, xyz + 3 * i, radius, buf, cap);
        }
        /* SPFH rows of the keypoint and of every neighbour, computed once each */
        for (int64_t t = 0; t <= k; ++t) {
            const int64_t j = t < k ? (int64_t)buf[t] : i;
            if (slot[j] < 0) {
                int64_t kj = grid_query(&g, xyz + 3 * j, radius, buf2, cap2);
                if (kj > cap2) {
                    cap2 = kj;
                    buf2 = (int32_t *)realloc(buf2, (size_t)cap2 * sizeof(int32_t));
                    kj = grid_query(&g, xyz + 3 * j, radius, buf2, cap2);
                }
                if (nrows == rows_cap) {
                    rows_cap *= 2;
                    cache = (double *)realloc(cache, (size_t)rows_cap * nb3 * sizeof(double));
                }
                spfh_row(xyz, normals, j, buf2, kj, n_bins, edges, cache + nrows * nb3);
                slot[j] = nrows++;
            }
            need[t] = slot[j];
        }
        for (int64_t t = 0; t < k; ++t) rows[t] = cache + need[t] * nb3; /* (after any realloc of the cache) */
        fpfh_row(xyz, i, buf, k, nb3, cache + need[k] * nb3, rows, out + q * nb3);
    }
    free(need);
    free(rows);
    free(cache);
    free(buf2);
    free(buf);
    free(slot);
    grid_free(&g);
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * (a11) cdist + argmin  (matching.py:47-52, 63-65, 164-168)
 * scipy cdist 'euclidean' (pinned 1.14.0, un-vendored): sqrt of the left-to-right sum of
 * squared differences; argmin returns the FIRST minimum.
 * ---------------------------------------------------------------------------------------- */
void orc_match_argmin(const double *a, int64_t m1, const double *b, int64_t m2, int64_t d, int64_t *idx,
                      double *dist, int64_t *col_idx /* nullable m2 */)
{
    double *colmin = col_idx ? (double *)malloc((size_t)(m2 ? m2 : 1) * sizeof(double)) : NULL;
    if (col_idx)
        for (int64_t j = 0; j < m2; ++j) { colmin[j] = INFINITY; col_idx[j] = 0; }
    for (int64_t i = 0; i < m1; ++i) {
        double best = INFINITY;
        int64_t bj = 0;
        for (int64_t j = 0; j < m2; ++j) {
            double s = 0.0;
            for (int64_t t = 0; t < d; ++t) {
                double df = a[i * d + t] - b[j * d + t];
                s += df * df;
            }
            s = sqrt(s);
            if (s < best) { best = s; bj = j; }
            if (col_idx && s < colmin[j]) { colmin[j] = s; col_idx[j] = i; }
        }
        idx[i] = bj;
        if (dist) dist[i] = best;
    }
    free(colmin);
}

/* ------------------------------------------------------------------------------------------
 * (a12) inlier count of ransac_on_matches (ransac.py:60-67) with RigidTransform.__getitem__
 * (core/rigid_transform.py:81-88): ||a R^T + t - b|| <= thr over all matches.
 * Rt: n_draws x 12 = row-major R (9) then t (3).
 * ---------------------------------------------------------------------------------------- */
void orc_ransac_score(const double *a, const double *b, int64_t m, const double *Rt, int64_t n_draws,
                      double thr, int64_t *inliers)
{
    for (int64_t dr = 0; dr < n_draws; ++dr) {
        const double *R = Rt + 12 * dr, *t = R + 9;
        int64_t cnt = 0;
        for (int64_t i = 0; i < m; ++i) {
            const double *p = a + 3 * i, *qv = b + 3 * i;
            double e0 = ((p[0] * R[0] + p[1] * R[1]) + p[2] * R[2]) + t[0] - qv[0];
            double e1 = ((p[0] * R[3] + p[1] * R[4]) + p[2] * R[5]) + t[1] - qv[1];
            double e2 = ((p[0] * R[6] + p[1] * R[7]) + p[2] * R[8]) + t[2] - qv[2];
            if (sqrt((e0 * e0 + e1 * e1) + e2 * e2) <= thr) ++cnt;
        }
        inliers[dr] = cnt;
    }
}
