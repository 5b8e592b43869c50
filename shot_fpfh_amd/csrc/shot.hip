// shot.hip -- K5, the SHOT descriptor (register-cached, team and streaming forms) and its entry points.
//
// Replaces: compute_single_shot_descriptor  shot.py:175-306; ShotMultiprocessor's drivers shot_parallelization.py:135-312;
//           the serial compute_shot_descriptor shot.py:310-499; get_azimuth_idx shot.py:51-70.
// Mapping: one wave per keypoint (the wave holds the whole neighbourhood, 64 neighbours per register chunk); lists above 255
// points: a team of waves per keypoint, streamed beyond 3 072.  Neighbours are gathered from the cell-sorted AoS records.
// All bin-deciding arithmetic is float64 with FMA contraction off.  HBM roofline, algorithmic bytes: the 352 x 8 = 2816 B row
// + 24 B keypoint + 72 B frame per keypoint.  (Until round 6: the body of descriptors.hip; K3 / K4: normals_lrf.hip.)
#include "common.h"
#include "device_util.h"
#include "host_stage.h"
#include <hip/amd_detail/amd_hip_unsafe_atomics.h>

int sf_launch_shot_lrf(sf_ctx *ctx, sf_cloud *c, sf_nbrs *nb, int raw, int skip_zero, double *dlrf); // normals_lrf.hip
int sf_launch_lrf_from_cov(sf_ctx *ctx, const double *cov_dev, int64_t m, double *dlrf);             // normals_lrf.hip

namespace {


// --------------------------------------------------------------------------------------------------
// K5: SHOT descriptor.
//
// The reference accumulates with ten NumPy fancy-index statements "D[idx] += val" (shot.py:244-298).
// With duplicate indices NumPy keeps only the LAST write of each statement, and neighbours are
// ordered by ascending rho (shot.py:218), so per statement and bin the neighbour with the LARGEST rho
// wins and the ten per-statement winners are summed.  The ten statements use five distinct writer
// keys: A = (ci,ti,pi,ri) [S2,S5,S8,S10], B = (ci+-1,ti,pi,ri) [S1], G = (ci,ti+-1,pi,ri) [S9],
// CD = (ci,ti,pi) [S3 -> radial bin 1, S4 -> radial bin 0], EF = (ci,ti,ri) [S6 -> elevation bin 1,
// S7 -> elevation bin 0].  Sweep 1 elects winners with 64-bit LDS atomicMax on the bit pattern of rho
// (positive doubles order like unsigned integers); sweep 2 lets each winner overwrite its slot with its
// value, tagged by the sign bit (all values are >= 0) so that later lanes cannot mistake it for a key.
// One wave per keypoint: the register-cached form (k_shot_cached, 5.5 KB of LDS, two phases) takes every list of at most
// 255 points, the streamed form (k_shot_long) the longer ones.
// --------------------------------------------------------------------------------------------------
#define SHOT_PI 3.141592653589793

__device__ inline int azimuth_octant(double x, double y) // get_azimuth_idx, shot.py:51-70
{
    const bool a = (y > 0.0) || ((y == 0.0) && (x < 0.0));
    const bool b = ((x > 0.0) || ((x == 0.0) && (y > 0.0))) != a;
    const bool c = ((x * y > 0.0) || (x == 0.0)) ? (fabs(x) < fabs(y)) : (fabs(x) > fabs(y));
    return 4 * (int)a + 2 * (int)b + (int)c;
}

// The same function for a whole wave with the special cases (a zero coordinate, |x| = |y|, a product that underflows)
// moved behind a wave-uniform test: off them the three bits are two sign tests and one magnitude comparison.
__device__ inline int azimuth_octant_wave(double x, double y)
{
    const double ax = fabs(x), ay = fabs(y);
    const bool special = !(fmin(ax, ay) > 1e-150) || ax == ay; // (also catches NaN)
    if (__ballot(special)) return azimuth_octant(x, y);
    const bool a = y > 0.0, xp = x > 0.0, lt = ax < ay;
    return 4 * (int)a + 2 * (int)(xp != a) + (int)(xp == a ? lt : !lt);
}

// get_azimuth_idx as an elementwise function (shot.py:51-70): the very device function K5 bins with
__global__ void k_azimuth_idx(const double *__restrict__ x, const double *__restrict__ y, int64_t n, int64_t *__restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = azimuth_octant(x[i], y[i]);
}

// A resolved slot holds the NEGATED value (values are >= 0, so the sign bit marks it); an unresolved key is
// the bit pattern of rho > 0 and an empty slot is +0.  Decoding is therefore max(-x, 0): one instruction.
__device__ inline unsigned long long tag_value(double v)
{
    return (unsigned long long)__double_as_longlong(v) | 0x8000000000000000ull;
}

__device__ inline double sf_dot3(double a0, double a1, double a2, double b0, double b1, double b2)
{
    return __builtin_fma(a2, b2, __builtin_fma(a1, b1, a0 * b0));
}

// --------------------------------------------------------------------------------------------------
// K5, register-cached form for neighbourhoods of at most 64*NCH points (the common case; the streaming
// kernel above stays as the fallback for larger ones).  The neighbour coordinates are gathered ONCE,
// all NCH chunks in flight together; sweep 1 does the geometry (local coordinates, bins, rho) and keeps
// the five numbers sweep 2 needs per neighbour in VGPRs, so sweep 2 is only the transcendental /
// interpolation part.  The azimuth-neighbour decision in sweep 1 uses the sign of the cross product with
// the octant's centre direction (no atan2); when that is not clearly non-zero it falls back to the
// reference's own expression, so the decision equals shot.py:283-288 in every case.
// --------------------------------------------------------------------------------------------------
// sqrt(x) and 1/sqrt(x) together.  The root is ocml's own f64 sequence (v_rsq_f64 and three coupled Newton steps)
// minus its exponent pre/post-scaling, which only matters outside [1e-290, 1e290]: bit-identical to sqrt() there
// (tools/ubench/sqrt_check.hip: 0 differences in 1.6e7 inputs), 10 instructions instead of 22; the half-inverse
// the iteration carries along, refined once more, is 1/sqrt(x) to ~1 ulp for two more instructions.
__device__ inline void sf_sqrt_rsqrt(double x, double &root, double &inv)
{
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y, h = 0.5 * y;
    double r = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, r, g);
    h = __builtin_fma(h, r, h);
    double d = __builtin_fma(-g, g, x);
    g = __builtin_fma(d, h, g);
    d = __builtin_fma(-g, g, x);
    g = __builtin_fma(d, h, g);
    r = __builtin_fma(-h, g, 0.5);
    h = __builtin_fma(h, r, h);
    root = g;
    inv = h + h;
    // (wave-uniform, never taken for real clouds; the test is on the exponent -- 2^-964 <= x < 2^963, i.e. 4.6e-291 .. 7.8e289,
    // positive, finite, not NaN -- one integer subtraction and comparison with 32-bit literals instead of two comparisons
    // against 64-bit constants that each cost two scalar moves)
    if (__ballot(!((unsigned)__double2hiint(x) - 0x03b00000u < 0x7c200000u - 0x03b00000u))) {
        root = sqrt(x);
        inv = 1.0 / root;
    }
}

// The same pair for a WAVE-UNIFORM argument (the squared norm of a row: a sum of squares of weights, each 0 or >= 1e-19, so
// either 0 or far inside the fast range): the range test is two scalar instructions on the exponent instead of two vector
// comparisons against 64-bit literals (six vector instructions with their moves).
__device__ inline void sf_sqrt_rsqrt_uniform(double x, double &root, double &inv)
{
    const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane(__double2hiint(x));
    if (hi - 0x03b00000u < 0x7c200000u - 0x03b00000u) { // 2^-964 <= x < 2^963, positive, finite
        const double y = __builtin_amdgcn_rsq(x);
        double g = x * y, h = 0.5 * y;
        double r = __builtin_fma(-h, g, 0.5);
        g = __builtin_fma(g, r, g);
        h = __builtin_fma(h, r, h);
        double d = __builtin_fma(-g, g, x);
        g = __builtin_fma(d, h, g);
        d = __builtin_fma(-g, g, x);
        g = __builtin_fma(d, h, g);
        r = __builtin_fma(-h, g, 0.5);
        h = __builtin_fma(h, r, h);
        root = g;
        inv = h + h;
    } else {
        root = sqrt(x);
        inv = 1.0 / root;
    }
}

// ---- the frame of a fused SHOT kernel ---------------------------------------------------------------------------------
// K4's raw mode leaves, in the frame's final row-major layout [x y z] per component, the largest / smallest eigenvectors as
// returned (x, z) and y = cross(z, x) of THOSE.  The fused kernels count the sign votes (shot.py:40-45) from the neighbours they
// have gathered anyway and flip: x and z by their own vote, y when exactly one of the two flipped (every product of the cross
// product changes sign, so the rounded difference does too; a component that cancelled to zero stays +0, as -a + a does).  All of
// it is wave-uniform: the nine numbers arrive by scalar loads and a flip is a scalar xor (y: plus one vector "+ 0").
// (sign: 0 or the sign bit as a 64-bit mask, one scalar select per axis -- pinned, or the compiler distributes the select over
// the components; a flip is then ONE 64-bit scalar xor per component)
__device__ inline double shot_flip(double v, unsigned long long sign) { return __longlong_as_double(__double_as_longlong(v) ^ (long long)sign); }
// raw: the nine raw numbers; E: the finished frame.  Returns whether anything changed (the caller writes E back if so).
__device__ inline bool shot_finish_frame(const double (&raw)[9], int k, int xneg, int zneg, double (&E)[9])
{
    // coordinates are finite (checked at upload), so "not < 0" is ">= 0": flip when strictly more neighbours project negative
    const bool fx = xneg > k - xneg, fz = zneg > k - zneg;
    if (k == 0) { // shot.py:24-25
        E[0] = 1.0; E[1] = 0.0; E[2] = 0.0; E[3] = 0.0; E[4] = 1.0; E[5] = 0.0; E[6] = 0.0; E[7] = 0.0; E[8] = 1.0;
        return true;
    }
    unsigned long long sx = fx ? 0x8000000000000000ull : 0ull, sz = fz ? 0x8000000000000000ull : 0ull;
    asm volatile("" : "+s"(sx), "+s"(sz));
    const unsigned long long sy = sx ^ sz;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        E[3 * i + 0] = shot_flip(raw[3 * i + 0], sx);
        // (-a + a is +0: a component of y that cancelled stays +0 when y flips -- what "flipped + 0.0" would do, as integer
        // operations on the scalar unit: the vector add put y's three components into six vector registers for the whole
        // kernel, the difference between five and six waves per SIMD for the fused four-chunk form)
        unsigned long long yb = (unsigned long long)__double_as_longlong(raw[3 * i + 1]) ^ sy;
        yb = (yb << 1) ? yb : 0ull;
        E[3 * i + 1] = __longlong_as_double((long long)yb);
        E[3 * i + 2] = shot_flip(raw[3 * i + 2], sz);
    }
    return fx | fz;
}

// Phase markers for tools/k5_phases.py (an ANALYSIS build only, -DSF_K5_MARK_BUILD: scheduling barriers + an assembler
// comment; the shipped build compiles them to nothing): instruction counts per phase of the register-cached K5.
#ifdef SF_K5_MARK_BUILD
#define SF_K5_MARK(id, nch)                                                     \
    do {                                                                        \
        __builtin_amdgcn_sched_barrier(0);                                      \
        asm volatile("; K5MARK %0 %1" ::"n"(id), "n"(nch));                     \
        __builtin_amdgcn_sched_barrier(0);                                      \
    } while (0)
// (pure arithmetic sinks to its first use whatever barrier stands between: the values a phase produces are pinned in front
// of the next mark)
#define SF_K5_PIN(x) asm volatile("" : "+v"(x))
#else
#define SF_K5_MARK(id, nch) do { } while (0)
#define SF_K5_PIN(x) (void)(x)
#endif
// ids: 1 header+clear, 2 gather, 3 frame votes, 4 gate, 5 geometry (sub: 50 sqrt/rsqrt, 51 local coords + cosine, 52 cosine bin,
// 53 octant, 54 centre-ray cross/dot + neighbour octant, 55 lz/rho + packing), 6 election A (atomic max), 7 who-writes-what
// (key reads), 8 weights (sub: 80 atan fraction, 81 radial shells, 82 acos, 83 elevation + sum), 9 A store (CAS), 10 S3/S4 S6/S7
// adds, 11 S1 S9 elections + adds, 12 read-back + normalise + store

struct shot_kept {
    double rho, dc, tcross, tdot, lzr; // tcross / tdot: (lx, ly) against the octant's centre ray; lzr = lz / rho
    unsigned bins0, bins1;            // base | bcos << 9 | bth << 18 ; bins1: bit 31 = valid, bits 28-30 = election flags
};

// ---- short double-precision helpers for sweep 2 (coefficients: tools/fit_poly.py) ------------------------
// The interpolation weights are continuous in theta / phi, so these only have to be accurate, not
// correctly rounded: each is within ~2e-16 (absolute) of the libm value the reference uses, far inside the
// 1e-5 parity tolerance, at a quarter of the instruction count of ocml's atan2 / acos / sqrt / division.
__device__ inline double sf_rcp(double d) { return sf_rcp_fast(d); } // v_rcp_f64 + 2 Newton steps, ~1 ulp

__device__ inline double sf_sqrt_small(double x) // sqrt(x), 0 <= x <= 1/4, no denormal / huge handling
{
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y, h = 0.5 * y;
    const double r = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, r, g);
    h = __builtin_fma(h, r, h);
    const double d = __builtin_fma(-g, g, x);
    g = __builtin_fma(d, h, g);
    return x > 0.0 ? g : 0.0;
}


// The two polynomials' coefficients live in device memory (a 26-double table the context owns, reached through a kernel argument)
// and arrive by scalar loads -- sixteen dwords per instruction -- right where they are used.  As immediates every coefficient
// costs two s_mov_b32 in front of its FMA (the instruction takes one scalar operand, and keeping 24 of them live across the
// kernel is 48 scalar registers the kernel does not have): 96 scalar moves per keypoint, a quarter of its scalar instructions,
// and the scalar stream is what holds the vector pipe back at seven waves per SIMD (tools/pmc_k5.sh: vector instructions
// 783 -> 701 per keypoint bought nothing until the scalar ones followed).  (A __constant__ array with an initialiser is folded
// back into immediates by the compiler.)
#define SF_K_ATAN 1.2732395447351628  // 4 / pi
#define SF_K_ACOS 0.6366197723675814  // 2 / pi
#define SF_ATAN_TERMS 11
#define SF_ACOS_TERMS 13
#define SF_ACOS_AT 12 // (offset of the second table: both start on a 32-byte boundary)
static const double SF_SHOT_COEF[SF_ACOS_AT + SF_ACOS_TERMS + 1] = {
    // atan(t) / t / (pi/4) in s = t^2, highest power first
    0.021102961440831885 * SF_K_ATAN, -0.04345403041920663 * SF_K_ATAN, 0.05687431322104835 * SF_K_ATAN, -0.06640058350060206 * SF_K_ATAN,
    0.07689933264608774 * SF_K_ATAN,  -0.09090771637100807 * SF_K_ATAN, 0.11111106118508882 * SF_K_ATAN, -0.14285714179450393 * SF_K_ATAN,
    0.19999999998836118 * SF_K_ATAN,  -0.33333333333328347 * SF_K_ATAN, SF_K_ATAN, 0.0,
    // asin(r) / r / (pi/2) in s = r^2, highest power first
    0.028169218060881414 * SF_K_ACOS, -0.010749050339697808 * SF_K_ACOS, 0.01603551434914882 * SF_K_ACOS, 0.0078029494773533175 * SF_K_ACOS,
    0.011875494382636922 * SF_K_ACOS, 0.013929652902326633 * SF_K_ACOS,  0.017355259955786323 * SF_K_ACOS, 0.02237204763174451 * SF_K_ACOS,
    0.03038194736709848 * SF_K_ACOS,  0.044642857103423646 * SF_K_ACOS,  0.07500000000020764 * SF_K_ACOS,  0.1666666666666665 * SF_K_ACOS,
    SF_K_ACOS, 0.0};

// atan(t) / (pi/4) for 0 <= t <= tan(pi/8) (1 + 1e-3): the minimax polynomial of tools/fit_poly.py with 4/pi folded into its
// coefficients at compile time -- the azimuth weight is |dth| = angle / (pi/4), so the angle itself is never needed
// (passing a loaded coefficient through an empty asm with a scalar-register constraint keeps it the FMA's scalar operand; left
// alone the compiler selects the accumulate form v_fmac_f64, whose addend is the destination: two v_mov_b32 per coefficient)
__device__ inline double sf_scalar_operand(double c)
{
    asm("" : "+s"(c));
    return c;
}
typedef const __attribute__((address_space(4))) double *sf_const_doubles; // (read through the scalar cache: never written by a kernel)
__device__ inline double sf_atan_octant_fraction(double t, const double *__restrict__ coef_)
{
    sf_const_doubles coef = (sf_const_doubles)coef_;
    const double s = t * t;
    double p = coef[0];
#pragma unroll
    for (int i = 1; i < SF_ATAN_TERMS; ++i) p = __builtin_fma(p, s, sf_scalar_operand(coef[i]));
    return t * p;
}

// acos(|z|) / (pi/2) for |z| <= 1 (result in [0, 1]; |z| = 0 gives exactly 1, |z| = 1 exactly 0): the asin-form minimax
// polynomial with 2/pi folded into its coefficients.  The elevation weights are linear in phi / (pi/2) and symmetric about
// the equator -- acos(-z) = pi - acos(z) -- so the angle of |z| is all they need (shot_weights).
__device__ inline double sf_acos_abs_quadrants(double az, const double *__restrict__ coef_)
{
    sf_const_doubles coef = (sf_const_doubles)coef_;
    const bool big = az > 0.5;
    const double xb = __builtin_fma(-0.5, az, 0.5), xs = az * az; // (1 - |z|) / 2 is exact
    const double rb = sf_sqrt_small(fmin(xb, 0.25));
    const double x = big ? xb : xs;
    const double r = big ? rb : az;
    double p = coef[SF_ACOS_AT]; // asin(r) = r + r s R(s), s = r^2 <= 1/4
#pragma unroll
    for (int i = 1; i < SF_ACOS_TERMS; ++i) p = __builtin_fma(p, x, sf_scalar_operand(coef[SF_ACOS_AT + i]));
    const double as = r * p; // asin(r) / (pi/2)
    // |z| <= 1/2: 1 - as ;  |z| > 1/2: 2 as
    return big ? as + as : 1.0 - as;
}

__device__ inline void shot_geometry(double cx, double cy, double cz, double d2, double nx, double ny, double nz,
                                     const double *E, double half_r, shot_kept &o)
{
    double rho, inv_rho;
    SF_K5_MARK(50, 0);
    sf_sqrt_rsqrt(d2, rho, inv_rho);
    SF_K5_PIN(rho); SF_K5_PIN(inv_rho);
    SF_K5_MARK(51, 0);
    // (neighbors - point) @ eigenvectors and normals @ eigenvectors[:, 2] (shot.py:214-215) as the multiply-add chain an
    // FMA BLAS kernel runs over the inner index -- what the reference's NumPy does on any current x86 / OpenBLAS
    const double lx = sf_dot3(cx, cy, cz, E[0], E[3], E[6]);
    const double ly = sf_dot3(cx, cy, cz, E[1], E[4], E[7]);
    const double lz = sf_dot3(cx, cy, cz, E[2], E[5], E[8]);
    double cosine = sf_dot3(nx, ny, nz, E[2], E[5], E[8]);
    cosine = fmin(fmax(cosine, -1.0), 1.0);
    double lxp = lx, lyp = ly, lzp = lz;
    SF_K5_PIN(lxp); SF_K5_PIN(lyp); SF_K5_PIN(lzp); SF_K5_PIN(cosine);
    SF_K5_MARK(52, 0);
    const double cpos = (cosine + 1.0) * 11.0 / 2.0 - 0.5;
    const double cf = rint(cpos);
    int ci = (int)cf;
    double dcp = cpos - cf;
    SF_K5_PIN(ci); SF_K5_PIN(dcp);
    SF_K5_MARK(53, 0);
    int ti = azimuth_octant_wave(lx, ly);
    SF_K5_PIN(ti);
    SF_K5_MARK(54, 0);
    const int pi_ = lz > 0.0 ? 1 : 0;
    const int ri = rho > half_r ? 1 : 0; // (radius / 2 is exact: the host passes that very double)
    const double dc = cpos - cf;
    const int sc = (dc > 0.0) - (dc < 0.0);
    int cin = ci + sc; // -1 .. 11, wrapped as the reference's % 11 does
    cin = cin < 0 ? 10 : (cin > 10 ? 0 : cin);
    // Offset from the octant's centre ray, in the octant's own frame: with a >= b the larger / smaller of |lx|, |ly|, the
    // point sits atan(b / a) in [0, pi/4] off the nearest axis and the centre ray pi/8 off it, so rotating (a, b) by
    // -pi/8 gives cross' / dot' = tan(angle off the centre).  Whether theta grows or shrinks with that angle alternates
    // from octant to octant (even octants: grows).
    const double C8 = 0.9238795325112867, S8 = 0.3826834323650898; // cos, sin of pi/8
    const double am = fmax(fabs(lx), fabs(ly)), bm = fmin(fabs(lx), fabs(ly));
    const double crs = __builtin_fma(C8, bm, -(S8 * am));
    const double dot = __builtin_fma(C8, am, S8 * bm);
    const double cross = (ti & 1) ? -crs : crs;
    int sth;
    if (fabs(cross) > 1e-9 * (fabs(lx) + fabs(ly))) {
        sth = cross > 0.0 ? 1 : -1;
    } else { // on (or within rounding of) the centre ray, or lx = ly = 0: the reference's expression decides
        const double tsz = 2 * SHOT_PI / 8;
        double dth = (atan2(ly, lx) - (-SHOT_PI + ti * tsz)) / tsz - 0.5;
        dth = fmin(fmax(dth, -0.5), 0.5);
        sth = (dth > 0.0) - (dth < 0.0);
    }
    const int tin = (ti + sth) & 7;
    const unsigned base = (unsigned)(((ci * 8 + ti) * 2 + pi_) * 2 + ri);
    const unsigned bcos = (unsigned)(((cin * 8 + ti) * 2 + pi_) * 2 + ri);
    const unsigned bth = (unsigned)(((ci * 8 + tin) * 2 + pi_) * 2 + ri);
    {
        double crp = cross, dtp = dot;
        unsigned b0p = base, b1p = bcos, b2p = bth;
        SF_K5_PIN(crp); SF_K5_PIN(dtp); SF_K5_PIN(b0p); SF_K5_PIN(b1p); SF_K5_PIN(b2p);
    }
    SF_K5_MARK(55, 0);
    // lz / rho through the reciprocal, then one residual correction: acos has an unbounded derivative at +-1, so for a
    // neighbour on the frame's z axis a one-ulp error of the quotient would be a 1.5e-8 error of phi (the reference's
    // correctly rounded division gives exactly +-1 there)
    double lzr = lz * inv_rho;
    lzr = __builtin_fma(__builtin_fma(-lzr, rho, lz), inv_rho, lzr);
    o.rho = rho; o.dc = dc; o.tcross = cross; o.tdot = dot; o.lzr = lzr;
    o.bins0 = base | (bcos << 9) | (bth << 18);
    o.bins1 = 0x80000000u | (base & 3u); // (bits 0-1: shell and half-space, for the form that re-maps its slot numbers)
}

// Radius-derived constants of the interpolation, computed once on the host (as kernel arguments they live in SGPRs;
// computed in the kernel the wave-uniform division 1 / (r/2) was a 14-instruction vector sequence per chunk)
struct shot_consts {
    double radius, half_r, q1, q3, inv_hr;
    const double *coef; // SF_SHOT_COEF in device memory
};

// The interpolation weights of one neighbour (shot.py:73-171, 244-298), reduced to what the elections consume:
//   vA   = S2 + S5 + S8 + S10 = (1 - |dc|) + current radial + current elevation + (1 - |dth|)
//   v_cd = S3's `outer` if the neighbour is in the inner shell, S4's `inner` if in the outer one (the other is 0)
//   v_ef = S6's `upper` if it is in the lower half space, S7's `lower` if in the upper one (the other is 0)
//   adth = |dth| (S9's value; S1's is |dc|, already in g)
// Written so that only the shell / half-space the neighbour is actually in gets evaluated: the centre of ITS bin is
// selected first, the distance to that centre computed once.  theta and phi enter as fractions of their bin size
// (sf_atan_octant_fraction, sf_acos_quadrants).  All weights are continuous in rho / phi / theta except at
// rho = r/2 (decided on rho itself, as the reference does) and phi = pi/2 (decided by the sign of z inside the
// reference's 1e-10 band), so last-bit differences of the short polynomial forms cannot flip a term.
__device__ inline void shot_weights(const shot_kept &g, const shot_consts &k, double &vA, double &v_cd, double &v_ef, double &adth)
{
    const bool z_pos = g.bins1 & 2u; // lz > 0, decided in shot_geometry
    const double rho = g.rho;
    const double adc = fabs(g.dc);
    // |dth|: angle off the octant's centre ray as a fraction of the octant, clipped to 1/2.  lx = ly = 0 has dot = 0:
    // the reference's atan2(0, 0) = 0 sits 3.5 octants from octant 0's start -> 1/2.
    SF_K5_MARK(80, 0);
    const bool fwd = g.tdot > 0.0;
    const double tq = fmin(fabs(g.tcross) * sf_rcp(fwd ? g.tdot : 1.0), 0.4146);
    const double at = fmin(sf_atan_octant_fraction(tq, k.coef), 0.5);
    adth = fwd ? at : 0.5;
    SF_K5_PIN(adth);
    // radial shells (interpolate_on_adjacent_husks): rho == r/2 belongs to neither and gets all three terms zero
    // The two shells mirror each other about rho = r/2: with s = |rho - r/2| the distance to the current shell's centre
    // is |s - r/4| and the distance "towards the other shell" (3r/4 - rho outside, rho - r/4 inside) is r/4 - s, in both.
    SF_K5_MARK(81, 0);
    const bool off_half = rho != k.half_r;
    const double ds = fabs(rho - k.half_r) - k.q1;
    const double cur = off_half ? 1.0 - fabs(ds) * k.inv_hr : 0.0;
    v_cd = off_half ? fmax(-ds, 0.0) * k.inv_hr : 0.0;
    { double curp = cur; SF_K5_PIN(curp); SF_K5_PIN(v_cd); }
    // elevation (interpolate_vertical_volumes).  With u = phi / (pi/2) the reference's terms are
    //   current = 1 - |u - 1/2| for phi < pi/2, 1 - |u - 3/2| for phi >= pi/2;
    //   lower  = [phi < pi/2 and (not near or z > 0) and phi >= pi/4] (u - 1/2), counted for z > 0 writers;
    //   upper  = [(phi > pi/2 or (near and z <= 0)) and phi <= 3pi/4] (3/2 - u), counted for z <= 0 writers
    // (near: |phi - pi/2| < 1e-10).  phi = acos(z) is symmetric about the equator, u(-z) = 2 - u(z), so in terms of
    // t = acos(|z|) / (pi/2) in [0, 1] all three are ONE expression per neighbour whatever the sign of z:
    //   current = 1 - |t - 1/2| ;  lower resp. upper = max(t - 1/2, 0), with the single exception the masks leave:
    //   a writer with z > 0 whose phi ROUNDS to pi/2 (t = 1 exactly) fails "phi < pi/2" and gets 0.
    SF_K5_MARK(82, 0);
    double t = sf_acos_abs_quadrants(fmin(fabs(g.lzr), 1.0), k.coef);
    SF_K5_PIN(t);
    SF_K5_MARK(83, 0);
    const double curv = 1.0 - fabs(t - 0.5);
    const bool side = !z_pos | (t < 1.0);
    v_ef = side ? fmax(t - 0.5, 0.0) : 0.0;
    vA = (((1.0 - adc) + cur) + curv) + (1.0 - adth);
}

// The waves of a K5 workgroup are independent (one keypoint and one LDS region each): what orders a wave's LDS phases is
// the in-order execution of its own LDS instructions, so the "barrier" is a compiler fence, never an s_barrier.
#define SF_SHOT_SYNC()                                                                                               \
    do {                                                                                                            \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");                                                      \
        __builtin_amdgcn_wave_barrier();                                                                            \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");                                                      \
    } while (0)
#ifndef SF_SHOT_WPB
#define SF_SHOT_WPB 2 // waves (= keypoints) per workgroup: 1.61 / 1.54 / 1.55 / 1.80 ms at C3 for 1 / 2 / 4 / 8
#endif

// Slot numbers of the cached form's LDS tables (round 6).  Bin b = ((cosine * 8 + azimuth) * 2 + half-space) * 2 + shell lives
// in slot b ^ (2 (b >> 5)).  With slot = b the bank pair of an 8-byte slot is b mod 32 = (azimuth, half-space, shell) WHATEVER
// the cosine bin -- and the 32 lanes an LDS pass serves hold neighbours that follow each other in the cell-sorted order, i.e.
// that lie in the same direction from the keypoint: same octant, same half-space, same shell, told apart by the cosine bin of
// their normals alone.  Their 64-bit election atomics met in a handful of bank pairs (SQ_LDS_BANK_CONFLICT 312 cycles per wave,
// half of the LDS pipe's active cycles; 239 cycles per wave waiting behind an LDS instruction).  XOR-ing bits 1-4 with the
// cosine bin puts the eleven cosine bins of one direction into eleven bank pairs: 206 conflict cycles, 105 waiting, K5 -2 %
// (tools/pmc_k5.sh and tools/ab_libs.sh, same box, three rounds).  Measured beside it: XOR with 3 x the cosine bin, which also
// spreads the shell bit (148 conflict cycles, but the row read-out then has to exchange the halves of its pairs: +10 vector
// instructions per keypoint, and the kernel is bound by those -- 1 % slower than this form); the shell bit moved to the top of
// the slot number (slot = cell + 176 shell: no change, 288 cycles -- the shell does not tell the lanes of a pass apart).
// The XOR stays inside the cosine bin's 32 slots, keeps a pair of adjacent bins adjacent and in order, and commutes with the
// ^ 1 / ^ 2 that reach the other shell / half-space; bit 1 of a slot number is no longer the half-space (bins1 carries it).
// All three 9-bit fields of bins0 at once.  (The team form keeps the plain numbering: with the two more registers the re-mapped
// read-out of its four tables takes, its fused instantiations drop from eight waves per SIMD to seven.)
__device__ inline unsigned shot_swz3(unsigned b)
{
    return b ^ ((b >> 4) & 0x783C1Eu); // every 9-bit field f becomes f ^ (2 (f >> 5))
}

template <int NCH, bool FUSED>
__device__ __forceinline__ void shot_cached_body(const double *__restrict__ rec,
                                                 const double *__restrict__ qx, const double *__restrict__ qy,
                                                 const double *__restrict__ qz, const int64_t *__restrict__ offset,
                                                 const int32_t *__restrict__ cnt, const int32_t *__restrict__ idx,
                                                 const int32_t *__restrict__ qrow, const shot_consts &K,
                                                 double *__restrict__ lrf, int normalize, int64_t min_nb,
                                                 double *__restrict__ out, int64_t q, unsigned long long *slot)
{
    // FUSED: `lrf` holds the raw axes written by k_shot_lrf(raw = 1); the sign votes (shot.py:40-45) are taken
    // here from the gathered neighbours and the finished frame is written back before it is used.
    // LDS: 704 election slots (5.5 KB per wave), used twice.  Phase 1 elects and resolves the writers of
    // S2+S5+S8+S10 (A, 352 slots), S3/S4 (CD, 176) and S6/S7 (EF, 176); phase 2 those of S1 (B, 352) and S9
    // (G, 352).  Halving the footprint (it was 11 KB with all five tables live at once) is what lets the
    // register file, not LDS, set the occupancy.  A CD / EF slot carries ONE value plus a flag in bit 62
    // (unused by doubles below 2.0): the S3/S4 pair of a winner has a single non-zero member, selected by
    // the winner's radial bin, and likewise S6/S7 by its elevation bin.
    SF_K5_MARK(1, NCH);
    unsigned long long *const sA = slot;
    const int lane = threadIdx.x & 63;
    const int64_t s = offset[q];
    const int k = cnt[q];
    const int64_t row = qrow ? qrow[q] : q;
    double *o = out + (int64_t)SF_SHOT_LEN * row;
    const double px = qx[q], py = qy[q], pz = qz[q];

    // (only the election / accumulator half: sX is cleared before each of its two uses; 16 bytes per lane and store)
    for (int b = lane; b < 176; b += 64) reinterpret_cast<ulonglong2 *>(slot)[b] = make_ulonglong2(0ull, 0ull);

    SF_K5_MARK(2, NCH);
    // one gather for all chunks
    double cx[NCH], cy[NCH], cz[NCH], nx[NCH], ny[NCH], nz[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int t = c * 64 + lane;
        const int j = t < k ? SF_LIST_LOAD(idx + s + t) : -1;
        const int jj = j < 0 ? 0 : j;
        double x, y, z;
        sf_load_pn(rec, jj, x, y, z, nx[c], ny[c], nz[c]);
        cx[c] = x - px; // padding lanes (j < 0) carry point 0's offset; they are masked by `on` / d2 = 0 below
        cy[c] = y - py;
        cz[c] = z - pz;
    }
    SF_K5_MARK(3, NCH);
    // the lanes of each chunk that hold a list entry, as wave-uniform masks: votes and gate are mask arithmetic on the scalar
    // unit (a ballot of a bare comparison is the comparison's own result register)
    unsigned long long onm[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int rem = k - 64 * c;
        onm[c] = rem >= 64 ? ~0ull : (rem > 0 ? (1ull << rem) - 1ull : 0ull);
    }
    double E[9];
    if (FUSED) {
        double *lr = lrf + 9 * row;
        double raw[9];
        // Scalar loads, by reading the nine numbers through the constant address space: this wave is the only one that touches
        // this frame, it reads it here and writes it below, so the scalar cache never holds a stale copy of it -- but the
        // compiler, seeing stores to `lrf` in the same kernel, proves that for one of the two inlined bodies only and loads the
        // other's frame into vector registers (every flip below then costs vector instructions).  The pin keeps all nine loads
        // in front of the votes (left alone the scheduler sinks the y loads to their first use: a scalar-memory round trip of
        // their own in the middle of the kernel).
        const __attribute__((address_space(4))) double *clr = (const __attribute__((address_space(4))) double *)lr;
#pragma unroll
        for (int i = 0; i < 9; ++i) raw[i] = clr[i];
#pragma unroll
        for (int i = 0; i < 9; ++i) asm volatile("" : "+s"(raw[i]));
        int xneg = 0, zneg = 0;
#pragma unroll
        for (int c = 0; c < NCH; ++c) { // the query itself (c = 0) votes ">= 0", as in the reference
            const double xo = sf_dot3(cx[c], cy[c], cz[c], raw[0], raw[3], raw[6]);
            const double zo = sf_dot3(cx[c], cy[c], cz[c], raw[2], raw[5], raw[8]);
            xneg += __popcll(__ballot(xo < 0.0) & onm[c]);
            zneg += __popcll(__ballot(zo < 0.0) & onm[c]);
        }
        if (shot_finish_frame(raw, k, xneg, zneg, E) && lane == 0) {
#pragma unroll
            for (int i = 0; i < 9; ++i) lr[i] = E[i]; // (streamed past the L2 like the rows: no difference)
        }
    } else {
#pragma unroll
        for (int i = 0; i < 9; ++i) E[i] = lrf[9 * row + i];
    }

    SF_K5_MARK(4, NCH);
    // gate (shot.py:212): neighbours at non-zero distance, as one mask per chunk (padding lanes carry point 0's offset: masked)
    double d2[NCH];
    unsigned long long posm[NCH];
    int npos = 0;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        d2[c] = (cx[c] * cx[c] + cy[c] * cy[c]) + cz[c] * cz[c];
        posm[c] = __ballot(d2[c] > 0.0) & onm[c];
        npos += __popcll(posm[c]);
    }
    if (!(npos > (int)min_nb)) { // (launch_shot passes min_nb clamped into [-1, 2^31 - 1]: a scalar 32-bit comparison)
        for (int b = lane; b < SF_SHOT_LEN; b += 64) o[b] = 0.0;
        return;
    }
    SF_SHOT_SYNC();
    // Election and accumulation.  sA (352 slots) first elects the writers of S2+S5+S8+S10 by rho, then becomes the
    // ACCUMULATOR of the row: its winners store their (negated) value, and every other statement's winner adds its own
    // (negated) value to the bin it feeds with an LDS float64 atomic add -- the LDS pipe does the additions, the bins are
    // never assembled by the vector ALU.  Every wave of the workgroup (SF_SHOT_WPB of them, one keypoint each) works in its
    // OWN LDS region and never waits for another; a wave's LDS instructions execute in program order, so the order of the
    // additions into a bin -- hence every bit of the row -- is the same in every run, whatever the workgroup size.
    // S3/S4 and S6/S7 need no election of their own: their writer is the farthest neighbour of a cell (cosine, azimuth,
    // half-space) over BOTH radial shells, resp. of a cell (cosine, azimuth, shell) over both half-spaces -- i.e. the
    // farther of the two S2 winners of bins base ^ 1, resp. base ^ 2.  Those two keys are read back after the election,
    // next to the bin's own, and compared: three plain reads replace two 64-bit LDS atomic maxima and two
    // compare-and-swaps per neighbour (the LDS pipe was 77 % busy, more than half of it bank
    // conflicts of the random 8-byte atomics -- tools/pmc_k5.sh).  sX (352 slots) serves S1 (B), then S9 (G).
    unsigned long long *const sX = slot + 352;
    double *const acc = reinterpret_cast<double *>(slot);
    constexpr unsigned long long SHOT_CLAIMED = ~0ull; // no rho has this bit pattern
    constexpr unsigned WON_A = 1u << 28, WON_CD = 1u << 29, WON_EF = 1u << 30; // flags kept in bins1 (bit 31 = valid)
    shot_kept g[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        g[c].bins1 = 0u;
        SF_K5_MARK(5, NCH);
        // (a branch per chunk on purpose: calling the geometry for all lanes puts the chunks' chains into one basic block, the
        // scheduler interleaves them and the kernel needs 87 registers instead of 70 -- 5 waves per SIMD, 1.72 ms; capped at
        // 72 / 80 registers it spills / runs 1.62 ms)
        if (__builtin_amdgcn_inverse_ballot_w64(posm[c])) {
            shot_geometry(cx[c], cy[c], cz[c], d2[c], nx[c], ny[c], nz[c], E, K.half_r, g[c]);
            g[c].bins0 = shot_swz3(g[c].bins0);
            SF_K5_MARK(6, NCH);
            const unsigned long long key = (unsigned long long)__double_as_longlong(g[c].rho);
            atomicMax(&sA[g[c].bins0 & 511u], key);
        }
    }
    SF_SHOT_SYNC();
    SF_K5_MARK(7, NCH);
    // who writes what (all reads of the keys come before the first winner replaces its key by a value)
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        if (__builtin_amdgcn_inverse_ballot_w64(posm[c])) {
            const unsigned long long key = (unsigned long long)__double_as_longlong(g[c].rho);
            const unsigned iA = g[c].bins0 & 511u;
            const bool up = g[c].bins1 & 2u, odd = g[c].bins1 & 1u; // (z > 0; outer shell -- iA is a SLOT number: shot_swz3)
            const unsigned long long own = sA[iA], other_shell = sA[iA ^ 1u], other_half = sA[iA ^ 2u];
            unsigned f = 0u;
            if (own == key) {
                f = WON_A;
                // every rho of the outer shell exceeds every rho of the inner one
                if (odd || other_shell == 0ull) f |= WON_CD;
                // equal distances in the two half-spaces: undefined in the reference (unstable argsort, shot.py:218); z > 0 here
                if (key > other_half || (key == other_half && up)) f |= WON_EF;
            }
            g[c].bins1 |= f;
        }
    }
    // the winners of A store; every neighbour keeps what its other statements may have to add
    double v_cd[NCH], v_ef[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        v_cd[c] = 0.0;
        v_ef[c] = 0.0;
        SF_K5_MARK(8, NCH);
        if (__builtin_amdgcn_inverse_ballot_w64(posm[c])) {
            const unsigned long long key = (unsigned long long)__double_as_longlong(g[c].rho);
            const unsigned iA = g[c].bins0 & 511u;
            double vA, adth;
            shot_weights(g[c], K, vA, v_cd[c], v_ef[c], adth);
            SF_K5_MARK(9, NCH);
            g[c].tdot = adth;
            // compare-and-swap, not a plain store: two neighbours at exactly the same distance (duplicated points) hold the
            // same key, and the ADDS below must come from one of them only (their values are equal; which one of two
            // distinct equidistant neighbours writes last is undefined in the reference too)
            if ((g[c].bins1 & WON_A) && atomicCAS(&sA[iA], key, tag_value(vA)) != key) g[c].bins1 &= ~(WON_A | WON_CD | WON_EF);
        }
    }
    SF_SHOT_SYNC();
    SF_K5_MARK(10, NCH);
    // S3/S4 and S6/S7: the writer adds into the bin with the OTHER radial / elevation bit than its own
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        if (g[c].bins1 & WON_A) {
            const unsigned iA = g[c].bins0 & 511u;
            if ((g[c].bins1 & WON_CD) && v_cd[c] != 0.0) unsafeAtomicAdd(&acc[iA ^ 1u], -v_cd[c]);
            if ((g[c].bins1 & WON_EF) && v_ef[c] != 0.0) unsafeAtomicAdd(&acc[iA ^ 2u], -v_ef[c]);
        }
    }
    SF_SHOT_SYNC();
    SF_K5_MARK(11, NCH);
    // S1 (value |dc|) and S9 (value |dth|): elect in sX, add into the accumulator
#pragma unroll
    for (int stmt = 0; stmt < 2; ++stmt) {
        for (int b = lane; b < 176; b += 64) reinterpret_cast<ulonglong2 *>(sX)[b] = make_ulonglong2(0ull, 0ull);
        SF_SHOT_SYNC();
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            if (__builtin_amdgcn_inverse_ballot_w64(posm[c])) {
                const unsigned long long key = (unsigned long long)__double_as_longlong(g[c].rho);
                atomicMax(&sX[(g[c].bins0 >> (stmt ? 18 : 9)) & 511u], key);
            }
        }
        SF_SHOT_SYNC();
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            if (__builtin_amdgcn_inverse_ballot_w64(posm[c])) {
                const unsigned long long key = (unsigned long long)__double_as_longlong(g[c].rho);
                const unsigned iW = (g[c].bins0 >> (stmt ? 18 : 9)) & 511u;
                const double val = stmt ? g[c].tdot : fabs(g[c].dc); // S1's range mask is always true for cf in 0..10
                if (atomicCAS(&sX[iW], key, SHOT_CLAIMED) == key && val != 0.0) unsafeAtomicAdd(&acc[iW], -val);
            }
        }
        SF_SHOT_SYNC();
    }
    SF_K5_MARK(12, NCH);
    // every slot of the accumulator is +0 (nothing written) or minus the bin's value: the row is acc * (-scale) + 0 (the sum
    // of squares does not see the sign; the "+ 0" makes an empty slot +0 whatever the sign of the scale).
    // A lane takes PAIRS of adjacent bins -- (2 l, 2 l + 1) of each third of the row -- so that the row leaves in three
    // 16-byte-per-lane store instructions instead of six 8-byte ones (and is read back in three LDS instructions): a wave's
    // slot is held until its stores are acknowledged, and the row is issued at the very end of the wave's work -- leaving
    // five of the six store instructions out (timing only) made the kernel 14 % faster, halving their number 1-1.5 %.
    double2 vals[3];
    double ss = 0.0;
#pragma unroll
    for (int u = 0; u < 3; ++u) {
        const int b = 2 * lane + 128 * u;
        // bins b, b + 1 are the slots b ^ g, (b ^ g) + 1 with g = 2 (b >> 5) = 2 (lane >> 4) + 8 u: an aligned pair, in order
        const int sb = ((2 * lane) ^ (2 * (lane >> 4)) ^ (8 * u)) + 128 * u;
        vals[u] = b < 352 ? *reinterpret_cast<const double2 *>(acc + sb) : make_double2(0.0, 0.0);
        ss += vals[u].x * vals[u].x;
        ss += vals[u].y * vals[u].y;
    }
    double nrm, inv_nrm; // (the short root / inverse-root pair: ~1 ulp each, a third of sqrt() followed by a division)
    sf_sqrt_rsqrt_uniform(sf_wave_sum(ss), nrm, inv_nrm);
    const double nscale = nrm > 0.0 ? (normalize ? -inv_nrm : -1.0) : -0.0; // shot.py:301-305
#pragma unroll
    for (int u = 0; u < 3; ++u) {
        const int b = 2 * lane + 128 * u;
        if (b < 352) sf_store_stream2(o + b, __builtin_fma(vals[u].x, nscale, 0.0), __builtin_fma(vals[u].y, nscale, 0.0));
    }
}


// --------------------------------------------------------------------------------------------------
// K5 for lists of ANY length (the keypoints whose own list exceeds the register-cached form: dense regions, and the
// reference's default configuration -- a support subsampled at radius / 10 puts 300-500 voxels' points into a ball on a
// surface scan).  Same arithmetic as the cached form (shot_geometry / shot_weights: the short polynomial forms, the octant
// fast path, S3/S4 and S6/S7 writers read off the S2 election), streamed: the list is swept three times, 128 neighbours in
// flight -- (V) sign votes of the frame + the gate, (1) geometry -> the three elections (64-bit LDS atomic max on rho's bit
// pattern for S2.., S1, S9), (2) geometry again -> weights -> every elected writer claims its (statement, bin) once (a bit
// per slot: two neighbours at exactly the same distance hold the same key) and ADDS its value into a float64 accumulator
// row with ds_add_f64.  LDS per wave: 3 x 352 keys + 352 accumulators + 3 x 352 claim bits = 11.4 KB.  A wave's LDS
// instructions execute in program order and no two lanes of one instruction target the same slot (one claimed writer per
// statement and bin), so every bit of the row is reproducible.  (Until round 4 this was `k_shot`, a two-sweep kernel on
// libm's atan2 / acos with five election tables: 2.3x the time per neighbour of the cached form.)
// --------------------------------------------------------------------------------------------------
#ifndef SF_SHOT_LONG_WPB
#define SF_SHOT_LONG_WPB 2
#endif
struct shot_long_lds {
    unsigned long long keyA[352], keyB[352], keyG[352];
    double acc[352];
    unsigned claim[3][12];
};

template <bool FUSED, bool SEL>
__global__ __launch_bounds__(64 * SF_SHOT_LONG_WPB) void k_shot_long(const double *__restrict__ rec,
                                             const double *__restrict__ qx, const double *__restrict__ qy,
                                             const double *__restrict__ qz, const int64_t *__restrict__ offset,
                                             const int32_t *__restrict__ cnt, const int32_t *__restrict__ idx,
                                             const int32_t *__restrict__ qrow,
                                             int64_t m, shot_consts K, double *__restrict__ lrf, int normalize,
                                             int64_t min_nb, double *__restrict__ out, const int32_t *__restrict__ sel,
                                             int64_t nsel, int64_t view_first, int lo)
{
    __shared__ __attribute__((aligned(16))) shot_long_lds lds_all[SF_SHOT_LONG_WPB];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    shot_long_lds &L = lds_all[wave];
    int64_t q = sf_xcd_block() * SF_SHOT_LONG_WPB + wave;
    if (SEL) {
        if (q >= nsel) return;
        q = (int64_t)sf_uniform(sel[q]) - view_first;
        if (q < 0) return;
    }
    if (q >= m) return;
    const int64_t s = offset[q];
    const int k = sf_uniform(cnt[q]);
    if (k <= lo) return; // (a list the team form of another launch holds: launch_shot)
    const int64_t row = qrow ? qrow[q] : q;
    double *o = out + (int64_t)SF_SHOT_LEN * row;
    const double px = qx[q], py = qy[q], pz = qz[q];
    {
        unsigned long long *w = reinterpret_cast<unsigned long long *>(&L);
        for (int b = lane; b < (int)(sizeof(shot_long_lds) / 8); b += 64) w[b] = 0ull;
    }
    double raw[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (FUSED) {
        const double *lr = lrf + 9 * row;
#pragma unroll
        for (int i = 0; i < 9; ++i) raw[i] = lr[i];
    }
    // ---- sweep V: gate (shot.py:212, 306) and the frame's sign votes (shot.py:40-45) ----
    int npos = 0, xneg = 0, zneg = 0;
    for (int t0 = 0; t0 < k; t0 += 128) {
        int jj[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int t = t0 + 64 * u + lane;
            jj[u] = t < k ? idx[s + t] : -1;
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            double x, y, z;
            sf_load_xyz(rec, jj[u] < 0 ? 0 : jj[u], x, y, z);
            const double cx = x - px, cy = y - py, cz = z - pz;
            const bool on = jj[u] >= 0;
            npos += __popcll(__ballot(on && ((cx * cx + cy * cy) + cz * cz) > 0.0));
            if (FUSED) {
                xneg += __popcll(__ballot(on && sf_dot3(cx, cy, cz, raw[0], raw[3], raw[6]) < 0.0));
                zneg += __popcll(__ballot(on && sf_dot3(cx, cy, cz, raw[2], raw[5], raw[8]) < 0.0));
            }
        }
    }
    double E[9];
    if (FUSED) { // (the frame is written whether or not the descriptor passes the gate, as in the cached form)
        if (shot_finish_frame(raw, k, xneg, zneg, E) && lane == 0) {
#pragma unroll
            for (int i = 0; i < 9; ++i) lrf[9 * row + i] = E[i];
        }
    } else {
#pragma unroll
        for (int i = 0; i < 9; ++i) E[i] = lrf[9 * row + i];
    }
    if (!((int64_t)npos > min_nb)) {
        for (int b = lane; b < SF_SHOT_LEN; b += 64) o[b] = 0.0;
        return;
    }
    SF_SHOT_SYNC(); // (the cleared tables)
    // one step of a sweep: 128 neighbours gathered together, their geometry
    auto geometry128 = [&](int t0, shot_kept (&g)[2]) {
        int jj[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int t = t0 + 64 * u + lane;
            jj[u] = t < k ? idx[s + t] : -1;
        }
        double cx[2], cy[2], cz[2], nx[2], ny[2], nz[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            double x, y, z;
            sf_load_pn(rec, jj[u] < 0 ? 0 : jj[u], x, y, z, nx[u], ny[u], nz[u]);
            cx[u] = x - px; cy[u] = y - py; cz[u] = z - pz;
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            g[u].bins1 = 0u;
            const double d2 = (cx[u] * cx[u] + cy[u] * cy[u]) + cz[u] * cz[u];
            if (jj[u] >= 0 && d2 > 0.0) shot_geometry(cx[u], cy[u], cz[u], d2, nx[u], ny[u], nz[u], E, K.half_r, g[u]);
        }
    };
    // ---- sweep 1: the three elections ----
    for (int t0 = 0; t0 < k; t0 += 128) {
        shot_kept g[2];
        geometry128(t0, g);
#pragma unroll
        for (int u = 0; u < 2; ++u)
            if (g[u].bins1 >> 31) {
                const unsigned long long key = (unsigned long long)__double_as_longlong(g[u].rho);
                atomicMax(&L.keyA[g[u].bins0 & 511u], key);
                atomicMax(&L.keyB[(g[u].bins0 >> 9) & 511u], key);
                atomicMax(&L.keyG[(g[u].bins0 >> 18) & 511u], key);
            }
    }
    SF_SHOT_SYNC();
    // ---- sweep 2: every elected writer claims its slot and adds its value ----
    auto claim = [&](int table, unsigned slot_) -> bool { // true for exactly one of the neighbours holding the winning key
        const unsigned bit = 1u << (slot_ & 31u);
        return (atomicOr(&L.claim[table][slot_ >> 5], bit) & bit) == 0u;
    };
    for (int t0 = 0; t0 < k; t0 += 128) {
        shot_kept g[2];
        geometry128(t0, g);
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (g[u].bins1 >> 31) {
                const unsigned long long key = (unsigned long long)__double_as_longlong(g[u].rho);
                const unsigned iA = g[u].bins0 & 511u, iB = (g[u].bins0 >> 9) & 511u, iG = (g[u].bins0 >> 18) & 511u;
                const bool up = iA & 2u, odd = iA & 1u; // (bit 1: z > 0, bit 0: outer shell)
                const unsigned long long own = L.keyA[iA], other_shell = L.keyA[iA ^ 1u], other_half = L.keyA[iA ^ 2u];
                const bool winB = L.keyB[iB] == key, winG = L.keyG[iG] == key;
                double vA, v_cd, v_ef, adth;
                shot_weights(g[u], K, vA, v_cd, v_ef, adth);
                if (own == key && claim(0, iA)) {
                    unsafeAtomicAdd(&L.acc[iA], vA);
                    // S3/S4: the farthest neighbour of the cell over BOTH shells (every rho of the outer shell exceeds every
                    // rho of the inner one); S6/S7: the farther of the two half-spaces' winners (equal distances: undefined in
                    // the reference -- unstable argsort, shot.py:218 --, z > 0 here, as in the cached form)
                    if ((odd || other_shell == 0ull) && v_cd != 0.0) unsafeAtomicAdd(&L.acc[iA ^ 1u], v_cd);
                    if ((key > other_half || (key == other_half && up)) && v_ef != 0.0) unsafeAtomicAdd(&L.acc[iA ^ 2u], v_ef);
                }
                const double adc = fabs(g[u].dc);
                if (winB && claim(1, iB) && adc != 0.0) unsafeAtomicAdd(&L.acc[iB], adc);
                if (winG && claim(2, iG) && adth != 0.0) unsafeAtomicAdd(&L.acc[iG], adth);
            }
        }
    }
    SF_SHOT_SYNC();
    double vals[6];
    double ss = 0.0;
#pragma unroll
    for (int u = 0; u < 6; ++u) {
        const int b = lane + 64 * u;
        const double v = b < 352 ? L.acc[b] : 0.0;
        vals[u] = v;
        ss += v * v;
    }
    double nrm, inv_nrm;
    sf_sqrt_rsqrt(sf_wave_sum(ss), nrm, inv_nrm);
    const double scale = nrm > 0.0 ? (normalize ? inv_nrm : 1.0) : 0.0; // shot.py:301-305
#pragma unroll
    for (int u = 0; u < 6; ++u) {
        const int b = lane + 64 * u;
        if (b < 352) sf_store_stream(o + b, vals[u] * scale);
    }
}

template <int NCH, bool FUSED>
__global__ __launch_bounds__(64 * SF_SHOT_WPB) void k_shot_cached(const double *__restrict__ rec,
                                                    const double *__restrict__ qx, const double *__restrict__ qy,
                                                    const double *__restrict__ qz, const int64_t *__restrict__ offset,
                                                    const int32_t *__restrict__ cnt, const int32_t *__restrict__ idx,
                                                    const int32_t *__restrict__ qrow,
                                                    int64_t m, shot_consts K, double *__restrict__ lrf,
                                                    int normalize, int64_t min_nb, double *__restrict__ out, int limit,
                                                    const int32_t *__restrict__ sel, int64_t nsel, int64_t view_first)
{
    __shared__ __attribute__((aligned(16))) unsigned long long slots[SF_SHOT_WPB][704];
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    unsigned long long *const slot = slots[wave];
    int64_t q = sf_xcd_block() * SF_SHOT_WPB + wave;
    if (sel) { // (the launch of the lists that need more chunks than the bulk: sf_dispatch::mid_sel, owner numbering)
        if (q >= nsel) return;
        q = (int64_t)sf_uniform(sel[q]) - view_first;
        if (q < 0) return;
    }
    if (q >= m) return;
    // (a keypoint whose own list is longer than this launch's form holds belongs to the second launch: launch_shot)
    if (sf_uniform(cnt[q]) > limit) return;
    // The kernel is instantiated for the LONGEST list of the launch (a 1M-point uniform cloud at 110 neighbours on average
    // has one of 160+), but nearly every keypoint fits one chunk less: a wave-uniform branch picks the body that
    // matches THIS keypoint, so the gather, votes and distance tests of an empty last chunk are never issued.
    if (NCH >= 3 && sf_uniform(cnt[q]) <= 64 * (NCH - 1)) {
        shot_cached_body<(NCH >= 3 ? NCH - 1 : NCH), FUSED>(rec, qx, qy, qz, offset, cnt, idx, qrow, K, lrf, normalize, min_nb, out, q, slot);
        return;
    }
    shot_cached_body<NCH, FUSED>(rec, qx, qy, qz, offset, cnt, idx, qrow, K, lrf, normalize, min_nb, out, q, slot);
}

// --------------------------------------------------------------------------------------------------
// K5 for lists of 256 .. 64 NCH NW points (round 5): a TEAM of waves -- one workgroup -- per keypoint.  The register-cached
// form's work, spread over waves: a list of n chunks is served by a = ceil(n / NCH) waves (the workgroup's other waves end at
// once; a barrier waits for the surviving waves only), wave w holding the chunks w, w + a, w + 2a of the list in registers:
// ONE gather, the geometry evaluated ONCE (the streaming form evaluates it twice, once to elect and once to add, because no
// single wave can keep 12 registers per chunk for a list of any length), the election tables are the workgroup's.  All three
// elections (S2.., S1, S9) run in one phase on three key tables; the winners then replace their key by their value (one
// compare-and-swap claims the slot and stores: duplicated points hold the same key) and the S3/S4 and S6/S7 writers store
// into a value table of their own (two addends per slot at most: commutative), so the row is the sum of the four tables in a
// fixed order -- every bit of it independent of how the waves were scheduled and of the team's size.  Four
// workgroup barriers per keypoint.  LDS: 4 x 352 x 8 B + 256 B = 11.3 KB per team.
// --------------------------------------------------------------------------------------------------
#ifndef SF_TEAM_NCH
#define SF_TEAM_NCH 3 // 70 registers: seven waves per SIMD (tools/ab_long.sh: 2 / 3 / 4 chunks per wave)
#endif
struct shot_team_lds {
    unsigned long long keyA[352], keyB[352], keyG[352]; // keys (rho's bit pattern), then tagged values: S2+S5+S8+S10, S1, S9
    double vx[352];                                      // S3/S4 + S6/S7 by destination bin (two addends at most: see below)
    __attribute__((aligned(16))) int red[16][4];         // per wave: gate count and the two sign votes
};

__device__ inline double shot_untag(unsigned long long x) { return fmax(-__longlong_as_double((long long)x), 0.0); }

template <int NCH, int NW, bool FUSED>
__global__ __launch_bounds__(64 * NW) void k_shot_team(const double *__restrict__ rec, const double *__restrict__ qx,
                                                       const double *__restrict__ qy, const double *__restrict__ qz,
                                                       const int64_t *__restrict__ offset, const int32_t *__restrict__ cnt,
                                                       const int32_t *__restrict__ idx, const int32_t *__restrict__ qrow,
                                                       int64_t m, shot_consts K, double *__restrict__ lrf, int normalize,
                                                       int64_t min_nb, double *__restrict__ out,
                                                       const int32_t *__restrict__ sel, int64_t nsel, int64_t view_first,
                                                       int lo)
{
    __shared__ __attribute__((aligned(16))) shot_team_lds L;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    int64_t q = sf_xcd_block();
    if (sel) {
        if (q >= nsel) return;
        q = (int64_t)sf_uniform(sel[q]) - view_first;
        if (q < 0) return;
    }
    if (q >= m) return;
    const int k = sf_uniform(cnt[q]);
    if (k <= lo || k > 64 * NCH * NW) return; // (another launch's list: launch_shot)
    const int nact = (((k + 63) >> 6) + NCH - 1) / NCH; // waves that hold a part of this list
    if (wave >= nact) return;
    const int nthr = 64 * nact;
    const int64_t s = offset[q];
    const int64_t row = qrow ? qrow[q] : q;
    double *o = out + (int64_t)SF_SHOT_LEN * row;
    const double px = qx[q], py = qy[q], pz = qz[q];
    {
        ulonglong2 *w = reinterpret_cast<ulonglong2 *>(&L);
        for (int b = threadIdx.x; b < 704; b += nthr) w[b] = make_ulonglong2(0ull, 0ull);
    }
    // this wave's chunks of the list: chunk w + a c of the list is its chunk c
    double cx[NCH], cy[NCH], cz[NCH], nx[NCH], ny[NCH], nz[NCH];
    unsigned long long onm[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int first = 64 * (wave + nact * c);
        cx[c] = cy[c] = cz[c] = nx[c] = ny[c] = nz[c] = 0.0;
        onm[c] = 0ull;
        if (first < k) { // (wave-uniform)
            const int t = first + lane;
            const int j = t < k ? SF_LIST_LOAD(idx + s + t) : -1;
            double x, y, z;
            sf_load_pn(rec, j < 0 ? 0 : j, x, y, z, nx[c], ny[c], nz[c]);
            cx[c] = x - px;
            cy[c] = y - py;
            cz[c] = z - pz;
            const int rem = k - first;
            onm[c] = rem >= 64 ? ~0ull : (1ull << rem) - 1ull;
        }
    }
    double raw[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    int xneg = 0, zneg = 0;
    if (FUSED) { // (read by every wave of the team before the barriers below, written back by one lane after them)
        const __attribute__((address_space(4))) double *clr = (const __attribute__((address_space(4))) double *)(lrf + 9 * row);
#pragma unroll
        for (int i = 0; i < 9; ++i) raw[i] = clr[i];
#pragma unroll
        for (int i = 0; i < 9; ++i) asm volatile("" : "+s"(raw[i]));
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            if (onm[c]) {
                const double xo = sf_dot3(cx[c], cy[c], cz[c], raw[0], raw[3], raw[6]);
                const double zo = sf_dot3(cx[c], cy[c], cz[c], raw[2], raw[5], raw[8]);
                xneg += __popcll(__ballot(xo < 0.0) & onm[c]);
                zneg += __popcll(__ballot(zo < 0.0) & onm[c]);
            }
        }
    }
    double d2[NCH];
    unsigned long long posm[NCH];
    int npos = 0;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        d2[c] = (cx[c] * cx[c] + cy[c] * cy[c]) + cz[c] * cz[c];
        posm[c] = __ballot(d2[c] > 0.0) & onm[c];
        npos += __popcll(posm[c]);
    }
    if (lane == 0) {
        L.red[wave][0] = npos;
        L.red[wave][1] = xneg;
        L.red[wave][2] = zneg;
    }
    __syncthreads(); // the cleared tables, the waves' counts
    npos = xneg = zneg = 0;
    for (int w = 0; w < nact; ++w) {
        const int4 part = *reinterpret_cast<const int4 *>(L.red[w]);
        npos += sf_uniform(part.x);
        xneg += sf_uniform(part.y);
        zneg += sf_uniform(part.z);
    }
    double E[9];
    if (FUSED) {
        if (shot_finish_frame(raw, k, xneg, zneg, E) && threadIdx.x == 0) {
#pragma unroll
            for (int i = 0; i < 9; ++i) lrf[9 * row + i] = E[i];
        }
    } else {
#pragma unroll
        for (int i = 0; i < 9; ++i) E[i] = lrf[9 * row + i];
    }
    if (!(npos > (int)min_nb)) { // (the same for every wave of the team)
        for (int b = threadIdx.x; b < SF_SHOT_LEN; b += nthr) o[b] = 0.0;
        return;
    }
    constexpr unsigned WON_B = 1u << 26, WON_G = 1u << 27, WON_A = 1u << 28, WON_CD = 1u << 29, WON_EF = 1u << 30;
    // ---- the three elections ----
    shot_kept g[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        g[c].bins1 = 0u;
        if (__builtin_amdgcn_inverse_ballot_w64(posm[c])) {
            shot_geometry(cx[c], cy[c], cz[c], d2[c], nx[c], ny[c], nz[c], E, K.half_r, g[c]);
            const unsigned long long key = (unsigned long long)__double_as_longlong(g[c].rho);
            atomicMax(&L.keyA[g[c].bins0 & 511u], key);
            atomicMax(&L.keyB[(g[c].bins0 >> 9) & 511u], key);
            atomicMax(&L.keyG[(g[c].bins0 >> 18) & 511u], key);
        }
    }
    __syncthreads();
    // ---- who writes what: every key is read before the first winner replaces one by its value ----
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        if (__builtin_amdgcn_inverse_ballot_w64(posm[c])) {
            const unsigned long long key = (unsigned long long)__double_as_longlong(g[c].rho);
            const unsigned iA = g[c].bins0 & 511u;
            const bool up = iA & 2u, odd = iA & 1u; // (bit 1: z > 0, bit 0: outer shell)
            const unsigned long long own = L.keyA[iA], other_shell = L.keyA[iA ^ 1u], other_half = L.keyA[iA ^ 2u];
            unsigned f = 0u;
            if (own == key) {
                f = WON_A;
                if (odd || other_shell == 0ull) f |= WON_CD; // (as in the cached form: shot_cached_body)
                if (key > other_half || (key == other_half && up)) f |= WON_EF;
            }
            if (L.keyB[(g[c].bins0 >> 9) & 511u] == key) f |= WON_B;
            if (L.keyG[(g[c].bins0 >> 18) & 511u] == key) f |= WON_G;
            g[c].bins1 |= f;
        }
    }
    __syncthreads();
    // ---- the winners' values: one claimed writer per slot ----
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        if (__builtin_amdgcn_inverse_ballot_w64(posm[c])) {
            const unsigned long long key = (unsigned long long)__double_as_longlong(g[c].rho);
            const unsigned iA = g[c].bins0 & 511u;
            double vA, v_cd, v_ef, adth;
            shot_weights(g[c], K, vA, v_cd, v_ef, adth);
            const unsigned f = g[c].bins1;
            if ((f & WON_A) && atomicCAS(&L.keyA[iA], key, tag_value(vA)) == key) {
                // (a bin receives at most one S3/S4 value -- from the winner of bin ^ 1 -- and one S6/S7 value -- from the winner
                // of bin ^ 2: two addends into a slot that starts at +0 give the same sum in either order)
                if ((f & WON_CD) && v_cd != 0.0) unsafeAtomicAdd(&L.vx[iA ^ 1u], v_cd);
                if ((f & WON_EF) && v_ef != 0.0) unsafeAtomicAdd(&L.vx[iA ^ 2u], v_ef);
            }
            if (f & WON_B) atomicCAS(&L.keyB[(g[c].bins0 >> 9) & 511u], key, tag_value(fabs(g[c].dc)));
            if (f & WON_G) atomicCAS(&L.keyG[(g[c].bins0 >> 18) & 511u], key, tag_value(adth));
        }
    }
    __syncthreads();
    // ---- the row: wave 0 sums the four tables in a fixed order, takes the norm, scales and stores (the other waves are done:
    //      with three writers, each repeating the sums and the norm for a third of the stores, the launch took 6 % longer --
    //      the team is bound by the vector instructions it issues, like the cached form) ----
    if (wave != 0) return;
    double2 vals[3];
    double ss = 0.0;
#pragma unroll
    for (int u = 0; u < 3; ++u) {
        const int b = 2 * lane + 128 * u;
        vals[u] = make_double2(0.0, 0.0);
        if (b < 352) {
            const ulonglong2 a = *reinterpret_cast<const ulonglong2 *>(L.keyA + b);
            const ulonglong2 bb = *reinterpret_cast<const ulonglong2 *>(L.keyB + b);
            const ulonglong2 gg = *reinterpret_cast<const ulonglong2 *>(L.keyG + b);
            const double2 vx = *reinterpret_cast<const double2 *>(L.vx + b);
            vals[u].x = ((shot_untag(a.x) + vx.x) + shot_untag(bb.x)) + shot_untag(gg.x);
            vals[u].y = ((shot_untag(a.y) + vx.y) + shot_untag(bb.y)) + shot_untag(gg.y);
        }
        ss += vals[u].x * vals[u].x;
        ss += vals[u].y * vals[u].y;
    }
    double nrm, inv_nrm;
    sf_sqrt_rsqrt_uniform(sf_wave_sum(ss), nrm, inv_nrm);
    const double scale = nrm > 0.0 ? (normalize ? inv_nrm : 1.0) : 0.0; // shot.py:301-305
#pragma unroll
    for (int u = 0; u < 3; ++u) {
        const int b = 2 * lane + 128 * u;
        if (b < 352) sf_store_stream2(o + b, vals[u].x * scale, vals[u].y * scale);
    }
}

} // namespace

// launch K5 on resident buffers; fused != 0: dlrf holds raw axes (k_shot_lrf raw mode) and receives the frames.
// Dispatch by list length, per keypoint (sf_nbrs_dispatch): the register-cached form of the main launch leaves out the
// keypoints whose own list exceeds it, and the streaming form serves exactly those in a second launch.
static int launch_shot(sf_ctx *ctx, sf_cloud *c, sf_nbrs *nb, double *dlrf, int normalize, int64_t min_nb, double *dout,
                       bool fused)
{
    const int64_t m = nb->m;
    if (!m) return SF_OK;
    // (a list holds fewer than 2^31 points: clamped here, "more than min_nb neighbours" is a 32-bit comparison in the kernels)
    min_nb = std::min<int64_t>(std::max<int64_t>(min_nb, -1), 2147483647LL);
    const dim3 grid(sf_xcd_grid(sf_div_up(m, SF_SHOT_WPB))), block(64 * SF_SHOT_WPB), block_streaming(64 * SF_SHOT_LONG_WPB);
#define SF_SHOT_ARGS c->rec, nb->qx, nb->qy, nb->qz, nb->offset, nb->count, nb->idx, nb->qrow, m
    const double r_ = nb->radius;
    if (!ctx->shot_coef) { // (once per context)
        SF_HIP(hipMalloc(&ctx->shot_coef, sizeof(SF_SHOT_COEF)));
        SF_HIP(hipMemcpy(ctx->shot_coef, SF_SHOT_COEF, sizeof(SF_SHOT_COEF), hipMemcpyHostToDevice));
    }
    const shot_consts K{r_, r_ / 2, r_ / 4, r_ * 3 / 4, 1.0 / (r_ / 2), ctx->shot_coef}; // the reference's own expressions (shot.py:95-117, 235)
    const sf_dispatch d = sf_nbrs_dispatch(nb);
#define SF_SHOT_CASE(N)                                                                                              \
    if (fused) { SF_LAUNCH(ctx, "k5_shot", (k_shot_cached<N, true>), grid, block, SF_SHOT_ARGS, K, dlrf, normalize, min_nb, dout, d.limit, (const int32_t *)nullptr, (int64_t)0, (int64_t)0); } \
    else { SF_LAUNCH(ctx, "k5_shot", (k_shot_cached<N, false>), grid, block, SF_SHOT_ARGS, K, dlrf, normalize, min_nb, dout, d.limit, (const int32_t *)nullptr, (int64_t)0, (int64_t)0); }
#define SF_SHOT_STREAM(NAME, SEL, GRID, SELP, NSEL, LO)                                                               \
    if (fused) { SF_LAUNCH(ctx, NAME, (k_shot_long<true, SEL>), dim3(sf_xcd_grid(sf_div_up(GRID, SF_SHOT_LONG_WPB))), block_streaming, SF_SHOT_ARGS, K, dlrf, normalize, min_nb, dout, SELP, NSEL, d.view_first, LO); } \
    else { SF_LAUNCH(ctx, NAME, (k_shot_long<false, SEL>), dim3(sf_xcd_grid(sf_div_up(GRID, SF_SHOT_LONG_WPB))), block_streaming, SF_SHOT_ARGS, K, dlrf, normalize, min_nb, dout, SELP, NSEL, d.view_first, LO); }
    // the team form (k_shot_team): one workgroup of NW waves per selected keypoint, lists of 256 .. 64 NCH NW points
#define SF_SHOT_TEAM(NCH, NW)                                                                                         \
    if (fused) { SF_LAUNCH(ctx, "k5_shot_tail", (k_shot_team<NCH, NW, true>), dim3(sf_xcd_grid(d.n_tail)), dim3(64 * NW), SF_SHOT_ARGS, K, dlrf, normalize, min_nb, dout, d.tail_sel, d.n_tail, d.view_first, 255); } \
    else { SF_LAUNCH(ctx, "k5_shot_tail", (k_shot_team<NCH, NW, false>), dim3(sf_xcd_grid(d.n_tail)), dim3(64 * NW), SF_SHOT_ARGS, K, dlrf, normalize, min_nb, dout, d.tail_sel, d.n_tail, d.view_first, 255); }
    if (d.chunks == 1) { SF_SHOT_CASE(1) }
    else if (d.chunks == 2) { SF_SHOT_CASE(2) }
    else if (d.chunks == 3) { SF_SHOT_CASE(3) }
    else if (d.chunks == 4) { SF_SHOT_CASE(4) }
    else { SF_SHOT_STREAM("k5_shot", false, m, (const int32_t *)nullptr, (int64_t)0, 0) }
    if (d.n_mid) { // the few lists that need more chunks than the bulk: the same form, four chunks
        const dim3 grid_mid(sf_xcd_grid(sf_div_up(d.n_mid, SF_SHOT_WPB)));
        if (fused) { SF_LAUNCH(ctx, "k5_shot_mid", (k_shot_cached<4, true>), grid_mid, block, SF_SHOT_ARGS, K, dlrf, normalize, min_nb, dout, 255, d.mid_sel, d.n_mid, d.view_first); }
        else { SF_LAUNCH(ctx, "k5_shot_mid", (k_shot_cached<4, false>), grid_mid, block, SF_SHOT_ARGS, K, dlrf, normalize, min_nb, dout, 255, d.mid_sel, d.n_mid, d.view_first); }
    }
    if (d.n_tail) {
        // Lists above 255 points: a team of waves per keypoint while the list fits the team's registers (the longest list of
        // the search decides which team sizes are launched at all; every launch walks the selection and takes its own lengths),
        // the streaming form beyond.  Team size: the smallest that holds the longest list of the search (a shorter list uses as many
        // of the team's waves as it needs).
        const int64_t longest = nb->max_count > 255 ? nb->max_count : INT64_MAX; // (unknown: every form is launched)
        constexpr int TN = SF_TEAM_NCH; // chunks a wave of the team holds
        int64_t covered;
        if (longest <= 64 * TN * 4) { SF_SHOT_TEAM(TN, 4) covered = 64 * TN * 4; }
        else if (longest <= 64 * TN * 8) { SF_SHOT_TEAM(TN, 8) covered = 64 * TN * 8; }
        else { SF_SHOT_TEAM(TN, 16) covered = 64 * TN * 16; }
        if (longest > covered) { SF_SHOT_STREAM("k5_shot_tail_stream", true, d.n_tail, d.tail_sel, d.n_tail, (int)covered) }
    }
#undef SF_SHOT_TEAM
#undef SF_SHOT_STREAM
#undef SF_SHOT_CASE
#undef SF_SHOT_ARGS
    return SF_OK;
}

extern "C" int sf_shot(sf_ctx *ctx, sf_cloud *c, sf_nbrs *nb, const double *lrf, int normalize, int64_t min_nb,
                       double *out, int flags)
{
    SF_CHECK(check_nbrs(ctx, c, nb, "sf_shot"));
    if (!lrf || !out) { sf_set_error("sf_shot: null lrf/out"); return SF_ERR_ARG; }
    SF_CHECK(sf_cloud_ensure_sorted_normals(ctx, c));
    const int64_t m = nb->m;
    sf_pool_guard tmp(ctx);
    const double *dlrf;
    double *dout;
    SF_CHECK(stage_in(tmp, lrf, (size_t)m * 9, flags, &dlrf));
    SF_CHECK(stage_out(tmp, out, (size_t)m * SF_SHOT_LEN, flags, &dout));
    SF_CHECK(launch_shot(ctx, c, nb, const_cast<double *>(dlrf), normalize, min_nb, dout, false));
    SF_CHECK(finish_out(ctx, out, (size_t)m * SF_SHOT_LEN, flags, dout));
    return stage_sync(ctx, flags);
}

extern "C" int sf_azimuth_idx(sf_ctx *ctx, const double *x, const double *y, int64_t n, int64_t *idx, int flags)
{
    if (!ctx || !x || !y || !idx || n < 0) { sf_set_error("sf_azimuth_idx: bad argument"); return SF_ERR_ARG; }
    SF_HIP(hipSetDevice(ctx->device));
    sf_pool_guard tmp(ctx);
    const double *dx, *dy;
    SF_CHECK(stage_in(tmp, x, (size_t)n, flags, &dx));
    SF_CHECK(stage_in(tmp, y, (size_t)n, flags, &dy));
    int64_t *dout = idx;
    if (!(flags & SF_OUT_DEVICE)) SF_CHECK(tmp.alloc(&dout, (size_t)n));
    if (n) {
        SF_LAUNCH(ctx, "k5_azimuth_idx", k_azimuth_idx, dim3((unsigned)sf_div_up(n, 256)), dim3(256), dx, dy, n, dout);
        if (dout != idx) SF_HIP(hipMemcpyAsync(idx, dout, (size_t)n * sizeof(int64_t), hipMemcpyDeviceToHost, ctx->stream));
    }
    return stage_sync(ctx, flags);
}

// compute_shot_descriptor, the reference's serial variant (shot.py:310-499): the frame of a keypoint is computed on its
// neighbours at NON-ZERO distance only (the keypoint itself and its duplicates are dropped first, :361-363) and the
// descriptor is always normalised (:496-497).  K4 with skip_zero, then the plain K5.
extern "C" int sf_shot_serial(sf_ctx *ctx, sf_cloud *c, sf_nbrs *nb, int64_t min_nb, double *out, int flags)
{
    SF_CHECK(check_nbrs(ctx, c, nb, "sf_shot_serial"));
    if (!out) { sf_set_error("sf_shot_serial: null out"); return SF_ERR_ARG; }
    SF_CHECK(sf_cloud_ensure_sorted_normals(ctx, c));
    const int64_t m = nb->m;
    sf_pool_guard guard(ctx);
    double *dlrf = nullptr, *dout = out;
    SF_CHECK(guard.alloc(&dlrf, (size_t)m * 9));
    if (!(flags & SF_OUT_DEVICE)) SF_CHECK(guard.alloc(&dout, (size_t)m * SF_SHOT_LEN));
    SF_CHECK(sf_launch_shot_lrf(ctx, c, nb, 0, 1, dlrf));
    SF_CHECK(launch_shot(ctx, c, nb, dlrf, 1, min_nb, dout, false));
    if (dout != out) {
        if (m) SF_HIP(hipMemcpyAsync(out, dout, (size_t)m * SF_SHOT_LEN * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        SF_HIP(hipStreamSynchronize(ctx->stream));
    }
    return SF_OK;
}

// Single-scale SHOT in one go (shot_parallelization.py:135-183: frames and descriptor from the SAME search):
// K4 without its vote sweep, then the fused K5, which votes on the neighbours it gathers anyway.
extern "C" int sf_shot_single_scale(sf_ctx *ctx, sf_cloud *c, sf_nbrs *nb, int normalize, int64_t min_nb, double *lrf,
                                    double *out, int flags)
{
    SF_CHECK(check_nbrs(ctx, c, nb, "sf_shot_single_scale"));
    if (!out) { sf_set_error("sf_shot_single_scale: null out"); return SF_ERR_ARG; }
    SF_CHECK(sf_cloud_ensure_sorted_normals(ctx, c));
    const int64_t m = nb->m;
    const bool out_dev = flags & SF_OUT_DEVICE;
    sf_pool_guard tmp(ctx);
    double *dlrf = lrf, *own_lrf = nullptr, *dout;
    if (!out_dev || !lrf) { // frames wanted on the host, or not wanted at all: scratch on the device
        SF_CHECK(tmp.alloc(&own_lrf, (size_t)m * 9));
        dlrf = own_lrf;
    }
    SF_CHECK(stage_out(tmp, out, (size_t)m * SF_SHOT_LEN, flags, &dout));
    const bool fused = true; // (the streaming K5 votes too: no list is too long for the fused path)
    SF_CHECK(sf_launch_shot_lrf(ctx, c, nb, fused ? 1 : 0, 0, dlrf));
    SF_CHECK(launch_shot(ctx, c, nb, dlrf, normalize, min_nb, dout, fused));
    if (own_lrf && lrf && m) SF_HIP(hipMemcpyAsync(lrf, own_lrf, (size_t)m * 9 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    SF_CHECK(finish_out(ctx, out, (size_t)m * SF_SHOT_LEN, flags, dout));
    if (!out_dev) SF_HIP(hipStreamSynchronize(ctx->stream));
    return SF_OK;
}

// sf_shot_single_scale with the frame moments already computed by sf_spfh_compute_moments on the SAME lists (self
// search; lists of any length: launch_shot dispatches per keypoint): eigen-solves only, then the fused K5.  cov: m x 6 on the device.
extern "C" int sf_shot_from_moments(sf_ctx *ctx, sf_cloud *c, sf_nbrs *nb, const double *cov_dev, int normalize, int64_t min_nb,
                                    double *lrf, double *out, int flags)
{
    SF_CHECK(check_nbrs(ctx, c, nb, "sf_shot_from_moments"));
    if (!out || !cov_dev) { sf_set_error("sf_shot_from_moments: null argument"); return SF_ERR_ARG; }
    if (nb->qrow) {
        sf_set_error("sf_shot_from_moments: needs a self search");
        return SF_ERR_UNSUPPORTED;
    }
    SF_CHECK(sf_cloud_ensure_sorted_normals(ctx, c));
    const int64_t m = nb->m;
    const bool out_dev = flags & SF_OUT_DEVICE;
    sf_pool_guard tmp(ctx);
    double *dlrf = lrf, *own_lrf = nullptr, *dout;
    if (!out_dev || !lrf) {
        SF_CHECK(tmp.alloc(&own_lrf, (size_t)m * 9));
        dlrf = own_lrf;
    }
    SF_CHECK(stage_out(tmp, out, (size_t)m * SF_SHOT_LEN, flags, &dout));
    SF_CHECK(sf_launch_lrf_from_cov(ctx, cov_dev, m, dlrf));
    SF_CHECK(launch_shot(ctx, c, nb, dlrf, normalize, min_nb, dout, true));
    if (own_lrf && lrf && m) SF_HIP(hipMemcpyAsync(lrf, own_lrf, (size_t)m * 9 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    SF_CHECK(finish_out(ctx, out, (size_t)m * SF_SHOT_LEN, flags, dout));
    if (!out_dev) SF_HIP(hipStreamSynchronize(ctx->stream));
    return SF_OK;
}

extern "C" int sf_shot_from_raw_lrf(sf_ctx *ctx, sf_cloud *c, sf_nbrs *nb, double *lrf_dev, int normalize, int64_t min_nb,
                                    double *out_dev)
{
    SF_CHECK(check_nbrs(ctx, c, nb, "sf_shot_from_raw_lrf"));
    if (!lrf_dev || !out_dev) { sf_set_error("sf_shot_from_raw_lrf: null argument"); return SF_ERR_ARG; }
    SF_CHECK(sf_cloud_ensure_sorted_normals(ctx, c));
    return launch_shot(ctx, c, nb, lrf_dev, normalize, min_nb, out_dev, true);
}

