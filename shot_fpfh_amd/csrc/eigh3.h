// eigh3.h -- device-side 3x3 symmetric eigensolver that reproduces numpy.linalg.eigh, i.e. LAPACK
// dsyevd(jobz='V', uplo='L'), INCLUDING the sign and ordering of the returned eigenvectors.
//
// Why not a closed-form / Jacobi solver: the reference keeps whatever sign LAPACK returns whenever
// the SHOT disambiguation vote is within +-1 of a tie (shot.py:40-45) and always for
// compute_normals without pre_computed_normals (pca_based_descriptors.py:51), so the sign is
// observable.  LAPACK's path for n = 3 is: dsytd2 (one Householder reflector, lower storage) ->
// dsteqr('I') (implicit QL/QR with Wilkinson shift, dlaev2 for 2x2 blocks, new-style dlartg,
// ascending selection sort) -> dormtr (apply the reflector).  The same sequence of operations is
// carried out here in float64 registers, so signs agree except for numerically repeated
// eigenvalues, where the basis is rounding noise in LAPACK too.
//
// All state is scalar (no arrays indexed at run time) so it stays in VGPRs.
#pragma once
#include <hip/hip_runtime.h>

namespace sf_eig {

__device__ inline double fsign(double a, double b) { return copysign(fabs(a), b); } // Fortran SIGN

__device__ inline double pythag(double x, double y) // dlapy2
{
    double xa = fabs(x), ya = fabs(y);
    double w = fmax(xa, ya), z = fmin(xa, ya);
    if (z == 0.0) return w;
    double t = z / w;
    return w * sqrt(1.0 + t * t);
}

// Givens rotation, LAPACK >= 3.10 convention (c >= 0, r carries the sign of f).
__device__ inline void givens(double f, double g, double &c, double &s, double &r)
{
    const double safmin = 2.2250738585072014e-308, safmax = 1.0 / safmin;
    const double rtmin = 1.4916681462400413e-154, rtmax = 9.480751908109176e+153; // sqrt(safmin), sqrt(safmax/2)
    double f1 = fabs(f), g1 = fabs(g);
    if (g == 0.0) {
        c = 1.0; s = 0.0; r = f;
    } else if (f == 0.0) {
        c = 0.0; s = fsign(1.0, g); r = g1;
    } else if (f1 > rtmin && f1 < rtmax && g1 > rtmin && g1 < rtmax) {
        double d = sqrt(f * f + g * g);
        c = f1 / d;
        r = fsign(d, f);
        s = g / r;
    } else {
        double u = fmin(safmax, fmax(safmin, fmax(f1, g1)));
        double fs = f / u, gs = g / u;
        double d = sqrt(fs * fs + gs * gs);
        c = fabs(fs) / d;
        r = fsign(d, f);
        s = gs / r;
        r = r * u;
    }
}

// Eigen-decomposition of [[a, b], [b, c]] (dlaev2): rt1 has the larger |.|, (cs1, sn1) its vector.
__device__ inline void sym2x2(double a, double b, double c, double &rt1, double &rt2, double &cs1, double &sn1)
{
    double sm = a + c, df = a - c, adf = fabs(df), tb = b + b, ab = fabs(tb);
    double acmx = fabs(a) > fabs(c) ? a : c, acmn = fabs(a) > fabs(c) ? c : a;
    double rt;
    if (adf > ab) { double t = ab / adf; rt = adf * sqrt(1.0 + t * t); }
    else if (adf < ab) { double t = adf / ab; rt = ab * sqrt(1.0 + t * t); }
    else rt = ab * sqrt(2.0);
    int sgn1;
    if (sm < 0.0) { rt1 = 0.5 * (sm - rt); sgn1 = -1; rt2 = (acmx / rt1) * acmn - (b / rt1) * b; }
    else if (sm > 0.0) { rt1 = 0.5 * (sm + rt); sgn1 = 1; rt2 = (acmx / rt1) * acmn - (b / rt1) * b; }
    else { rt1 = 0.5 * rt; rt2 = -0.5 * rt; sgn1 = 1; }
    double cs;
    int sgn2;
    if (df >= 0.0) { cs = df + rt; sgn2 = 1; } else { cs = df - rt; sgn2 = -1; }
    if (fabs(cs) > ab) {
        double ct = -tb / cs;
        sn1 = 1.0 / sqrt(1.0 + ct * ct);
        cs1 = ct * sn1;
    } else if (ab == 0.0) {
        cs1 = 1.0; sn1 = 0.0;
    } else {
        double tn = -cs / tb;
        cs1 = 1.0 / sqrt(1.0 + tn * tn);
        sn1 = tn * cs1;
    }
    if (sgn1 == sgn2) { double tn = cs1; cs1 = -sn1; sn1 = tn; }
}

// Tridiagonal 3x3 state: diagonal d1..d3, off-diagonal e1 (between 1,2) and e2 (between 2,3),
// eigenvector accumulator Z (columns z1, z2, z3; each a 3-vector of scalars).
struct tri3 {
    double d1, d2, d3, e1, e2;
    double z11, z21, z31, z12, z22, z32, z13, z23, z33; // z<row><col>

    // rotate columns (j, j+1), j in {1,2}: dlasr side='R', pivot='V' body
    __device__ inline void rot(int j, double ct, double st)
    {
        if (ct == 1.0 && st == 0.0) return;
        if (j == 1) {
            double t;
            t = z12; z12 = ct * t - st * z11; z11 = st * t + ct * z11;
            t = z22; z22 = ct * t - st * z21; z21 = st * t + ct * z21;
            t = z32; z32 = ct * t - st * z31; z31 = st * t + ct * z31;
        } else {
            double t;
            t = z13; z13 = ct * t - st * z12; z12 = st * t + ct * z12;
            t = z23; z23 = ct * t - st * z22; z22 = st * t + ct * z22;
            t = z33; z33 = ct * t - st * z32; z32 = st * t + ct * z32;
        }
    }
    // select-based accessors: a run-time index into a register array would be spilled to scratch
    __device__ inline double D(int i) const { return i == 1 ? d1 : (i == 2 ? d2 : d3); }
    __device__ inline double E(int i) const { return i == 1 ? e1 : e2; }
    __device__ inline void setD(int i, double v)
    {
        d1 = i == 1 ? v : d1;
        d2 = i == 2 ? v : d2;
        d3 = i == 3 ? v : d3;
    }
    __device__ inline void setE(int i, double v)
    {
        e1 = i == 1 ? v : e1;
        e2 = i == 2 ? v : e2;
    }
    __device__ inline void swap_cols(int i, int k)
    {
        // only (1,2), (1,3), (2,3) occur
        double t;
        if (i == 1 && k == 2) { t = z11; z11 = z12; z12 = t; t = z21; z21 = z22; z22 = t; t = z31; z31 = z32; z32 = t; }
        else if (i == 1 && k == 3) { t = z11; z11 = z13; z13 = t; t = z21; z21 = z23; z23 = t; t = z31; z31 = z33; z33 = t; }
        else { t = z12; z12 = z13; z13 = t; t = z22; z22 = z23; z23 = t; t = z32; z32 = z33; z33 = t; }
    }
};

// dsteqr(compz='I') specialised to n = 3.  The sequence of floating-point operations per matrix is LAPACK's (labels in
// comments) because the rotation sequence determines the eigenvector signs; its QL and QR loops share one body (below).
__device__ inline void steqr3(tri3 &T)
{
    const double eps = 1.1102230246251565e-16, eps2 = eps * eps, safmin = 2.2250738585072014e-308;
    const double ssfmax = 4.4692311799612e+153 /* sqrt(1/safmin)/3 */, ssfmin = 1.2100683175775647e-122 /* sqrt(safmin)/eps^2 */;
    T.z11 = T.z22 = T.z33 = 1.0;
    T.z21 = T.z31 = T.z12 = T.z32 = T.z13 = T.z23 = 0.0;
    const int n = 3, nmaxit = 90;
    int jtot = 0, l1 = 1;
    while (l1 <= n) { // label 10
        // (written out for n = 3: compile-time indices instead of the run-time-indexed accessors)
        if (l1 == 2) T.e1 = 0.0;
        if (l1 == 3) T.e2 = 0.0;
        int m = n;
        if (l1 == 1) { // mm = 1
            const double tst = fabs(T.e1);
            if (tst == 0.0) m = 1;
            else if (tst <= (sqrt(fabs(T.d1)) * sqrt(fabs(T.d2))) * eps) { T.e1 = 0.0; m = 1; }
        }
        if (m == n && l1 <= 2) { // mm = 2
            const double tst = fabs(T.e2);
            if (tst == 0.0) m = 2;
            else if (tst <= (sqrt(fabs(T.d2)) * sqrt(fabs(T.d3))) * eps) { T.e2 = 0.0; m = 2; }
        }
        int l = l1, lsv = l, lend = m, lendsv = lend;
        l1 = m + 1;
        if (lend == l) continue;
        double anorm = 0.0; // max |d(l..lend)|, |e(l..lend-1)| in dsteqr's order
        if (l <= 1 && lend >= 1) anorm = fmax(anorm, fabs(T.d1));
        if (l <= 2 && lend >= 2) anorm = fmax(anorm, fabs(T.d2));
        if (lend >= 3) anorm = fmax(anorm, fabs(T.d3));
        if (l <= 1 && lend >= 2) anorm = fmax(anorm, fabs(T.e1));
        if (l <= 2 && lend >= 3) anorm = fmax(anorm, fabs(T.e2));
        int iscale = 0;
        if (anorm == 0.0) continue;
        if (anorm > ssfmax) {
            iscale = 1;
            for (int i = l; i <= lend; ++i) T.setD(i, T.D(i) / anorm * ssfmax);
            for (int i = l; i <= lend - 1; ++i) T.setE(i, T.E(i) / anorm * ssfmax);
        } else if (anorm < ssfmin) {
            iscale = 2;
            for (int i = l; i <= lend; ++i) T.setD(i, T.D(i) / anorm * ssfmin);
            for (int i = l; i <= lend - 1; ++i) T.setE(i, T.E(i) / anorm * ssfmin);
        }
        if (fabs(T.D(lend)) < fabs(T.D(l))) { lend = lsv; l = lendsv; }
        // ---- QL (lend > l) and QR (lend < l) iteration in ONE loop body, on a mirrored copy ----
        // LAPACK writes the two as separate loops that are mirror images of each other -- the same expressions with the
        // index walking the other way and the saved rotation's sine negated.  A lane runs one or the other, so with two
        // copies of the code a wave pays for both.  Here the tridiagonal is copied into (A1, A2, A3 | B1, B2) read from
        // l's side -- node i' = i for QL, 4 - i for QR; pair p' (between nodes p' and p' + 1) = pair p resp. 3 - p -- and
        // every lane walks upwards through the same instructions with compile-time indices (the run-time-indexed
        // accessors, two or three selects per access, were two thirds of the old loop).  What is NOT mirror-symmetric in
        // dsteqr is kept as it is there: dlaev2 always takes (lower index, off-diagonal, upper index) of the ORIGINAL
        // numbering, and the rotations go to the original column pairs.
        {
            const bool fw = lend > l;
            int lm = fw ? l : 4 - l;            // l in the mirrored numbering: moves up
            const int lendm = fw ? lend : 4 - lend;
            double A1 = fw ? T.d1 : T.d3, A2 = T.d2, A3 = fw ? T.d3 : T.d1;
            double B1 = fw ? T.e1 : T.e2, B2 = fw ? T.e2 : T.e1;
            for (;;) {
                // look for a small off-diagonal between l and lend, starting at l (labels 40 / 90)
                int mm_ = lendm;
                if (lm == 1 && lendm >= 2 && fabs(B1) * fabs(B1) <= (eps2 * fabs(A1)) * fabs(A2) + safmin) mm_ = 1;
                else if (lm <= 2 && lendm == 3 && fabs(B2) * fabs(B2) <= (eps2 * fabs(A2)) * fabs(A3) + safmin) mm_ = 2;
                if (mm_ < lendm) { // E(m) := 0
                    B1 = mm_ == 1 ? 0.0 : B1;
                    B2 = mm_ == 2 ? 0.0 : B2;
                }
                if (mm_ == lm) { // an eigenvalue has converged (labels 80 / 130)
                    lm += 1;
                    if (lm <= lendm) continue;
                    break;
                }
                if (mm_ == lm + 1) { // a 2 x 2 block between mirrored nodes lm, lm + 1 (lm is 1 or 2)
                    const bool low = lm == 1;
                    const double an = low ? A1 : A2, af = low ? A2 : A3, bb = low ? B1 : B2; // node lm, node lm + 1, their pair
                    double rt1, rt2, c, s;
                    // dlaev2(d(lo), e(lo), d(lo + 1)) with lo the lower ORIGINAL index: node lm for QL, node lm + 1 for QR
                    sym2x2(fw ? an : af, bb, fw ? af : an, rt1, rt2, c, s);
                    T.rot(fw ? lm : 3 - lm, c, s); // original pair index
                    const double nn = fw ? rt1 : rt2, nf = fw ? rt2 : rt1; // d(lo) = rt1, d(lo + 1) = rt2
                    A1 = low ? nn : A1;
                    A2 = low ? nf : nn;
                    A3 = low ? A3 : nf;
                    B1 = low ? 0.0 : B1;
                    B2 = low ? B2 : 0.0;
                    lm += 2;
                    if (lm <= lendm) continue;
                    break;
                }
                if (jtot == nmaxit) break;
                ++jtot;
                // The step proper: here lm = 1 and the small-off-diagonal search came back with m = 3, the whole matrix.
                // Same operations in the same order as dsteqr's loops at labels 70 / 120.
                double p = A1;
                double g = (A2 - p) / (2.0 * B1); // Wilkinson shift from the pair next to l
                double r = pythag(g, 1.0);
                g = A3 - p + (B1 / (g + fsign(r, g)));
                double s = 1.0, c = 1.0;
                p = 0.0;
                // first rotation: the pair next to m
                double f = s * B2, b = c * B2;
                givens(g, f, c, s, r);
                g = A3 - p;
                r = (A2 - g) * s + 2.0 * c * b;
                p = s * r;
                A3 = g + p;
                g = c * r - b;
                const double c_k0 = c, s_k0 = fw ? -s : s;
                // second rotation: the pair next to l
                f = s * B1; b = c * B1;
                givens(g, f, c, s, r);
                B2 = r;
                g = A2 - p;
                r = (A1 - g) * s + 2.0 * c * b;
                p = s * r;
                A2 = g + p;
                g = c * r - b;
                const double c_k1 = c, s_k1 = fw ? -s : s;
                // the eigenvector columns, in the order the rotations were made (dlasr 'B' for QL, 'F' for QR)
                T.rot(fw ? 2 : 1, c_k0, s_k0);
                T.rot(fw ? 1 : 2, c_k1, s_k1);
                A1 = A1 - p;
                B1 = g;
            }
            T.d1 = fw ? A1 : A3; T.d2 = A2; T.d3 = fw ? A3 : A1;
            T.e1 = fw ? B1 : B2; T.e2 = fw ? B2 : B1;
        }
        if (iscale == 1) {
            for (int i = lsv; i <= lendsv; ++i) T.setD(i, T.D(i) / ssfmax * anorm);
            for (int i = lsv; i <= lendsv - 1; ++i) T.setE(i, T.E(i) / ssfmax * anorm);
        } else if (iscale == 2) {
            for (int i = lsv; i <= lendsv; ++i) T.setD(i, T.D(i) / ssfmin * anorm);
            for (int i = lsv; i <= lendsv - 1; ++i) T.setE(i, T.E(i) / ssfmin * anorm);
        }
        if (jtot >= nmaxit) break;
    }
    // ascending selection sort with column swaps (label 160), written out for n = 3: position 1 takes the first strict
    // minimum of (d1, d2, d3), then position 2 the smaller of what is left (ties keep their order, as in dsteqr)
    {
        int k = 1;
        double p = T.d1;
        if (T.d2 < p) { k = 2; p = T.d2; }
        if (T.d3 < p) { k = 3; p = T.d3; }
        if (k != 1) {
            T.d2 = k == 2 ? T.d1 : T.d2;
            T.d3 = k == 3 ? T.d1 : T.d3;
            T.d1 = p;
            T.swap_cols(1, k);
        }
        if (T.d3 < T.d2) {
            const double t = T.d2;
            T.d2 = T.d3;
            T.d3 = t;
            T.swap_cols(2, 3);
        }
    }
}

// Lower triangle in: a11, a21, a31, a22, a32, a33.  Out: eigenvalues w1 <= w2 <= w3 and eigenvectors
// as columns: v<row><col>, column k belongs to w<k> (numpy's v[:, k-1]).
struct eig3 {
    double w1, w2, w3;
    double v11, v21, v31, v12, v22, v32, v13, v23, v33;
};

__device__ inline eig3 eigh3_lower(double a11, double a21, double a31, double a22, double a32, double a33)
{
    tri3 T;
    double tau = 0.0, v2 = 0.0;
    // dsytd2, uplo='L', i = 1: dlarfg(2, a21, a31)
    double xnorm = fabs(a31);
    if (xnorm == 0.0) {
        T.e1 = a21;
    } else {
        double beta = -fsign(pythag(a21, xnorm), a21);
        tau = (beta - a21) / beta;
        v2 = a31 * (1.0 / (a21 - beta));
        T.e1 = beta;
        // x := tau * A22 * v (dsymv, lower), v = (1, v2)
        double x1 = tau * a22 + tau * (a32 * v2);
        double x2 = tau * a32 + (tau * v2) * a33;
        double alpha = -0.5 * tau * (x1 + x2 * v2);
        double w1 = x1 + alpha, w2 = x2 + alpha * v2;
        // A22 := A22 - v w^T - w v^T (dsyr2, lower)
        a22 = a22 + ((-w1) + w1 * (-1.0));
        a32 = a32 + (v2 * (-w1) + w2 * (-1.0));
        a33 = a33 + (v2 * (-w2) + w2 * (-v2));
    }
    T.d1 = a11;
    T.e2 = a32; // i = 2: dlarfg(1, ...) is the identity
    T.d2 = a22;
    T.d3 = a33;
    steqr3(T);
    // dormtr('L','L','N'): rows 2..3 of Z := (I - tau v v^T) rows 2..3
    if (tau != 0.0) {
        double t;
        t = -tau * (T.z21 + T.z31 * v2); T.z21 = T.z21 + t; T.z31 = T.z31 + v2 * t;
        t = -tau * (T.z22 + T.z32 * v2); T.z22 = T.z22 + t; T.z32 = T.z32 + v2 * t;
        t = -tau * (T.z23 + T.z33 * v2); T.z23 = T.z23 + t; T.z33 = T.z33 + v2 * t;
    }
    eig3 r;
    r.w1 = T.d1; r.w2 = T.d2; r.w3 = T.d3;
    r.v11 = T.z11; r.v21 = T.z21; r.v31 = T.z31;
    r.v12 = T.z12; r.v22 = T.z22; r.v32 = T.z32;
    r.v13 = T.z13; r.v23 = T.z23; r.v33 = T.z33;
    return r;
}

} // namespace sf_eig
