// devmath.h -- fp64 elementwise math for the CDNA4 SGP4/SDP4 kernels.
//
// Replaces the reference's simdMath (src/simdMath.zig L29-212: sincosN / atan2N / modTwoPiN /
// pow15N / pow23N) with a design that fits a lane that OWNS one satellite and walks time:
//   * az_sincos      full-range sincos (2-term Cody-Waite + degree-13/12 minimax), ~1 ulp
//   * az_rotate*     (sin,cos) of angle+d from (sin,cos) of angle for small d by a short Taylor
//                    rotation -- the work-horse: slowly drifting angles (arg of perigee, node,
//                    Kepler corrections, J2 short-period corrections) never need a fresh sincos
//   * az_rcp/az_rsqrt hardware v_rcp_f64 / v_rsq_f64 seeds (~2^-23) + Newton refinement;
//                    the reference's 11+K divides and 4 square roots become 3 rcp + 2 rsqrt
//   * no atan2 and no mod-2pi at all on the near-earth path: u is only ever used through
//                    sin/cos, and (sin u, cos u) are already known exactly
//
// The header is plain C++ so that the SAME source can also be compiled for the host by the
// test-only emulation harness (tests/host_emul); the product only ever compiles it with hipcc.
#pragma once
#include <math.h>

#ifndef AZ_DEVICE
#define AZ_DEVICE __device__ __forceinline__
#endif

#ifdef AZ_HOST_EMUL
// test-only stand-ins for the gfx950 intrinsics (seed precision mimics v_rcp_f64/v_rsq_f64)
static inline double az_hw_rcp(double x) { return (double)(1.0f / (float)x); }
static inline double az_hw_rsq(double x) { return (double)(1.0f / sqrtf((float)x)); }
static inline bool az_any(bool p) { return p; }
static inline double az_rint(double x) { return nearbyint(x); }
#else
AZ_DEVICE double az_hw_rcp(double x) { return __builtin_amdgcn_rcp(x); }
AZ_DEVICE double az_hw_rsq(double x) { return __builtin_amdgcn_rsq(x); }
// wave64 vote: true if the predicate holds on any active lane (s_cmp on the exec-masked ballot)
AZ_DEVICE bool az_any(bool p) { return __builtin_amdgcn_ballot_w64(p) != 0ull; }
AZ_DEVICE double az_rint(double x) { return __builtin_rint(x); }
#endif

#define AZ_PI 3.14159265358979323846
#define AZ_TWOPI 6.28318530717958647692

// ---------------------------------------------------------------- reciprocal / rsqrt
// 1/x to ~1 ulp with ONE cubically convergent step: seed error e0 ~ 2^-23 -> e0^3 ~ 2^-69.
//   r' = r (1 + e + e^2),  e = 1 - x r            (3 FMA-class ops after the seed)
AZ_DEVICE double az_rcp(double x)
{
    const double r = az_hw_rcp(x);
    const double e = fma(-x, r, 1.0);
    return fma(r, fma(e, e, e), r);
}
// 1/x to ~2^-46: enough for a Newton *step* (the step is re-evaluated next trip)
AZ_DEVICE double az_rcp1(double x)
{
    const double r = az_hw_rcp(x);
    return fma(r, fma(-x, r, 1.0), r);
}
// 1/sqrt(x) to ~1 ulp, one cubic step:  y' = y (1 + e/2 + 3 e^2/8),  e = 1 - x y^2
AZ_DEVICE double az_rsqrt(double x)
{
    const double y = az_hw_rsq(x);
    const double e = fma(-(x * y), y, 1.0);
    return fma(y, e * fma(e, 0.375, 0.5), y);
}

// ---------------------------------------------------------------- where polynomial coefficients come from
// A v_fma_f64 takes no 64-bit literal.  Inside a loop the compiler therefore parks every polynomial coefficient in a VGPR
// pair for the whole loop AND copies it (v_mov_b64) in front of every v_fmac that would overwrite it: the deep-space step
// -- six sincos, two 1/8-rad rotations, an atan2, a binomial series -- carried ~70 registers of literals and a hundred
// moves per step that way.  The math below takes its coefficients through an accessor m.mc(k): literals (McLit: host
// emulation, lane = satellite kernels, one-off code) or words of an LDS table read by all lanes at once (kernels.h:
// a ds_read delivers a fresh register the fmac may consume in place -- no residency, no move).
enum AzMathConst {
    MC_2OPI, MC_PIO2HI, MC_PIO2LO,                         // sincos reduction
    MC_S1, MC_S2, MC_S3, MC_S4, MC_S5, MC_S6,              // sin kernel, highest degree first
    MC_C1, MC_C2, MC_C3, MC_C4, MC_C5,                     // cos kernel
    MC_R_Q0, MC_R_Q1, MC_R_Q2, MC_R_Q3, MC_R_P0, MC_R_P1, MC_R_P2, MC_R_P3, // 1/8-rad rotation (az_rotate_med)
    MC_AT_T3, MC_AT_T1, MC_AT_TK, MC_AT_PI4, MC_AT_PI8,    // atan2 reduction: tan(3 pi/16), tan(pi/16), tan(pi/8), pi/4, pi/8
    MC_AT_P0, MC_AT_P1, MC_AT_P2, MC_AT_P3, MC_AT_P4, MC_AT_P5, MC_AT_P6, // atan polynomial, highest degree first
    MC_PIO2, MC_PI,
    MC_A23_0, MC_A23_1, MC_A23_2, MC_A23_3, MC_A23_4,      // (1 + x)^(-2/3) series
    MC_L_Q0, MC_L_Q1, MC_L_Q2, MC_L_Q3, MC_L_Q4, MC_L_Q5, MC_L_Q6,          // 1/2-rad rotation (az_rotate_large): cos - 1
    MC_L_P0, MC_L_P1, MC_L_P2, MC_L_P3, MC_L_P4, MC_L_P5, MC_L_P6,          // ... sin
    MC_NUM
};
#define AZ_MC_VALUES                                                                                                     \
    {0.63661977236758134308, 1.57079632679489655800e+00, 6.12323399573676603587e-17,                                     \
     1.58969099521155010221e-10, -2.50507602534068634195e-08, 2.75573137070700676789e-06, -1.98412698298579493134e-04,   \
     8.33333333332248946124e-03, -1.66666666666666324348e-01,                                                            \
     -1.13596475577881948265e-11, 2.08757232129817482790e-09, -2.75573143513906633035e-07, 2.48015872894767294178e-05,   \
     -1.38888888888741095749e-03,                                                                                        \
     -1.0 / 3628800.0, 1.0 / 40320.0, -1.0 / 720.0, 1.0 / 24.0, 1.0 / 362880.0, -1.0 / 5040.0, 1.0 / 120.0, -1.0 / 6.0,  \
     6.68178637919298879e-01, 1.98912367379658006e-01, 4.14213562373095034e-01, 7.85398163397448279e-01,                 \
     3.92699081698724139e-01,                                                                                            \
     6.83755761589315975e-02, -9.04573884142570867e-02, 1.11099128876421274e-01, -1.42856978924566086e-01,               \
     1.99999998929785122e-01, -3.33333333330710246e-01, 9.99999999999999667e-01,                                         \
     1.57079632679489655800e+00, 3.14159265358979323846,                                                                 \
     -308.0 / 729.0, 110.0 / 243.0, -40.0 / 81.0, 5.0 / 9.0, -2.0 / 3.0,                                                 \
     1.0 / 20922789888000.0, -1.0 / 87178291200.0, 1.0 / 479001600.0, -1.0 / 3628800.0, 1.0 / 40320.0, -1.0 / 720.0,     \
     1.0 / 24.0,                                                                                                         \
     -1.0 / 1307674368000.0, 1.0 / 6227020800.0, -1.0 / 39916800.0, 1.0 / 362880.0, -1.0 / 5040.0, 1.0 / 120.0,          \
     -1.0 / 6.0}
// (MC_C6 = 4.16666666666666019037e-02 and the -0.5 / 1.0 / 0.5 of the kernels stay literals: inline constants or one use)
AZ_DEVICE double az_mc_literal(int k)
{
    constexpr double v[MC_NUM] = AZ_MC_VALUES;
    return v[k];
}
struct McLit {
#ifdef AZ_HOST_EMUL
    inline double mc(int k) const { return az_mc_literal(k); }
#else
    __device__ __forceinline__ double mc(int k) const { return az_mc_literal(k); }
#endif
};

// ---------------------------------------------------------------- full-range sincos
// |x| up to ~1e6 rad with < 2e-16 absolute error (2-term Cody-Waite: k*pio2_lo residual 1e-33*k).
template <class M>
AZ_DEVICE void az_sincos_m(double x, double &s, double &c, const M &m)
{
    const double kf = az_rint(x * m.mc(MC_2OPI));
    double r = fma(-kf, m.mc(MC_PIO2HI), x);
    r = fma(-kf, m.mc(MC_PIO2LO), r);
    const int k = (int)kf;
    const double z = r * r;
    // minimax on [-pi/4, pi/4] (the classic fdlibm kernel coefficients)
    double ps = fma(m.mc(MC_S1), z, m.mc(MC_S2));
    ps = fma(ps, z, m.mc(MC_S3));
    ps = fma(ps, z, m.mc(MC_S4));
    ps = fma(ps, z, m.mc(MC_S5));
    ps = fma(ps, z, m.mc(MC_S6));
    const double sr = fma(ps, z * r, r);
    double pc = fma(m.mc(MC_C1), z, m.mc(MC_C2));
    pc = fma(pc, z, m.mc(MC_C3));
    pc = fma(pc, z, m.mc(MC_C4));
    pc = fma(pc, z, m.mc(MC_C5));
    pc = fma(pc, z, 4.16666666666666019037e-02);
    const double cr = fma(z, fma(pc, z, -0.5), 1.0);
    const bool swap = (k & 1) != 0;
    double ss = swap ? cr : sr;
    double cc = swap ? sr : cr;
    s = (k & 2) ? -ss : ss;
    c = ((k + 1) & 2) ? -cc : cc;
}
AZ_DEVICE void az_sincos(double x, double &s, double &c) { az_sincos_m(x, s, c, McLit()); }

// ---------------------------------------------------------------- register-resident constants
// v_fma_f64 accepts at most ONE scalar/literal source and no 64-bit literal at all, so a Horner
// step fma(d2, C1, C2) with two fp64 constants costs an extra v_mov_b64 every time the compiler
// re-materialises a constant instead of keeping it live.  The four Taylor coefficients of the two
// rotation tiers on the hot path are therefore parked in VGPRs once per kernel; the empty asm makes
// the value opaque so that it cannot be folded back into an immediate.
AZ_DEVICE double az_opaque(double x)
{
#ifndef AZ_HOST_EMUL
    asm volatile("" : "+v"(x));
#endif
    return x;
}
struct RotK {
    double n6, p24, p120, n720; // -1/6, 1/24, 1/120, -1/720
};
AZ_DEVICE RotK az_rotk()
{
    RotK k;
    k.n6 = az_opaque(-1.0 / 6.0);
    k.p24 = az_opaque(1.0 / 24.0);
    k.p120 = az_opaque(1.0 / 120.0);
    k.n720 = az_opaque(-1.0 / 720.0);
    return k;
}

// ---------------------------------------------------------------- small rotations
// (s,c) <- (sin,cos)(angle + d).  Written as s += (s*q + c*p), q = cos d - 1, p = sin d, so the
// rounding error is that of one addition to s (c), not of a product.
#define AZ_ROT_MED 0.125       /* sin to d^9, cos to d^10: truncation d^11/11! < 3e-18, d^12/12! < 2e-20 */
#define AZ_ROT_SMALL 0.0078125 /* 2^-7 : sin to d^5, cos to d^6: d^7/5040 < 4e-19, d^8/40320 < 4e-22       */
#define AZ_ROT_MILLI 0.0009765625 /* 2^-10: sin to d^3, cos to d^4: d^5/120 < 8e-18, d^6/720 < 2e-21        */
#define AZ_ROT_TINY 1.0e-4     /* same polynomial as MILLI; kept as the Newton tail threshold              */

AZ_DEVICE void az_rot_apply(double &s, double &c, double p, double q)
{
    const double ns = s + fma(s, q, c * p);
    const double nc = c + fma(c, q, -(s * p));
    s = ns;
    c = nc;
}
// |d| <= 1/8
template <class M>
AZ_DEVICE void az_rotate_med_m(double &s, double &c, double d, const M &m)
{
    const double d2 = d * d;
    double q = fma(d2, m.mc(MC_R_Q0), m.mc(MC_R_Q1));
    q = fma(d2, q, m.mc(MC_R_Q2));
    q = fma(d2, q, m.mc(MC_R_Q3));
    q = d2 * fma(d2, q, -0.5);
    double p = fma(d2, m.mc(MC_R_P0), m.mc(MC_R_P1));
    p = fma(d2, p, m.mc(MC_R_P2));
    p = fma(d2, p, m.mc(MC_R_P3));
    p = d * fma(d2, p, 1.0);
    az_rot_apply(s, c, p, q);
}
AZ_DEVICE void az_rotate_med(double &s, double &c, double d) { az_rotate_med_m(s, c, d, McLit()); }
// (p,q) = (sin d, cos d - 1) of the 2^-7 tier
AZ_DEVICE void az_pq_small(double d, const RotK &k, double &p, double &q)
{
    const double d2 = d * d;
    q = d2 * fma(d2, fma(d2, k.n720, k.p24), -0.5);
    p = d * fma(d2, fma(d2, k.p120, k.n6), 1.0);
}
// |d| <= 2^-7
AZ_DEVICE void az_rotate_small(double &s, double &c, double d, const RotK &k)
{
    double p, q;
    az_pq_small(d, k, p, q);
    az_rot_apply(s, c, p, q);
}
// |d| <= 2^-10
AZ_DEVICE void az_rotate_tiny(double &s, double &c, double d, const RotK &k)
{
    const double d2 = d * d;
    const double q = d2 * fma(d2, k.p24, -0.5);
    const double p = d * fma(d2, k.n6, 1.0);
    az_rot_apply(s, c, p, q);
}
// sincos(d) + angle addition, any d
AZ_DEVICE void az_rotate_full(double &s, double &c, double d)
{
    double sd, cd;
    az_sincos(d, sd, cd);
    const double ns = fma(s, cd, c * sd);
    const double nc = fma(c, cd, -(s * sd));
    s = ns;
    c = nc;
}
// |d| <= 1/2: sin to d^15, cos to d^16 (truncation 0.5^17/17! = 2e-20); no integer work, no
// range reduction -- cheaper than sincos(d) + angle addition for the first Newton step of an
// eccentric orbit
template <class M>
AZ_DEVICE void az_rotate_large_m(double &s, double &c, double d, const M &m)
{
    const double d2 = d * d;
    double q = fma(d2, m.mc(MC_L_Q0), m.mc(MC_L_Q1));
    q = fma(d2, q, m.mc(MC_L_Q2));
    q = fma(d2, q, m.mc(MC_L_Q3));
    q = fma(d2, q, m.mc(MC_L_Q4));
    q = fma(d2, q, m.mc(MC_L_Q5));
    q = fma(d2, q, m.mc(MC_L_Q6));
    q = d2 * fma(d2, q, -0.5);
    double p = fma(d2, m.mc(MC_L_P0), m.mc(MC_L_P1));
    p = fma(d2, p, m.mc(MC_L_P2));
    p = fma(d2, p, m.mc(MC_L_P3));
    p = fma(d2, p, m.mc(MC_L_P4));
    p = fma(d2, p, m.mc(MC_L_P5));
    p = fma(d2, p, m.mc(MC_L_P6));
    p = d * fma(d2, p, 1.0);
    az_rot_apply(s, c, p, q);
}
AZ_DEVICE void az_rotate_large(double &s, double &c, double d) { az_rotate_large_m(s, c, d, McLit()); }
// any d: wave-uniform choice of the cheapest valid tier (up to four votes: use the az_rotate_le_*
// forms below where the usual magnitude is known); m: where the wide tiers' coefficients come from
template <class M>
AZ_DEVICE void az_rotate_m(double &s, double &c, double d, const RotK &k, const M &m)
{
    const double ad = fabs(d);
    if (!az_any(ad > AZ_ROT_SMALL)) {
        if (!az_any(ad > AZ_ROT_MILLI))
            az_rotate_tiny(s, c, d, k);
        else
            az_rotate_small(s, c, d, k);
    } else if (!az_any(ad > AZ_ROT_MED)) {
        az_rotate_med_m(s, c, d, m);
    } else if (!az_any(ad > 0.5)) {
        az_rotate_large_m(s, c, d, m);
    } else {
        double sd, cd;
        az_sincos_m(d, sd, cd, m);
        const double ns = fma(s, cd, c * sd);
        const double nc = fma(c, cd, -(s * sd));
        s = ns;
        c = nc;
    }
}
AZ_DEVICE void az_rotate(double &s, double &c, double d, const RotK &k) { az_rotate_m(s, c, d, k, McLit()); }
template <class M>
AZ_DEVICE void az_rotate_le_tiny_m(double &s, double &c, double d, const RotK &k, const M &m)
{
    if (!az_any(fabs(d) > AZ_ROT_MILLI))
        az_rotate_tiny(s, c, d, k);
    else
        az_rotate_m(s, c, d, k, m);
}
// one vote for the expected tier, generic fallback otherwise
AZ_DEVICE void az_rotate_le_tiny(double &s, double &c, double d, const RotK &k)
{
    if (!az_any(fabs(d) > AZ_ROT_MILLI))
        az_rotate_tiny(s, c, d, k);
    else
        az_rotate(s, c, d, k);
}
AZ_DEVICE void az_rotate_le_small(double &s, double &c, double d, const RotK &k)
{
    if (!az_any(fabs(d) > AZ_ROT_SMALL))
        az_rotate_small(s, c, d, k);
    else
        az_rotate(s, c, d, k);
}
// (s,c) of a+b from (sa,ca),(sb,cb)
AZ_DEVICE void az_angle_add(double sa, double ca, double sb, double cb, double &s, double &c)
{
    s = fma(sa, cb, ca * sb);
    c = fma(ca, cb, -(sa * sb));
}

// atan2(y, x) to ~5e-16 without libm: ratio of the smaller to the larger magnitude (a in [0,1]), one more reduction
// about tan(k pi/8), k = 0..2 (atan a = k pi/8 + atan((a - c)/(1 + a c)), |z| <= tan(pi/16)), degree-13 odd polynomial
// (near-minimax fit: 4.4e-16), octant fix-ups by selects.  ~40 instructions, no branches; atan2(0, 0) = 0.
// Only the deep-space Lyddane branch needs an angle VALUE (everything else lives on (sin,cos) pairs).
template <class M>
AZ_DEVICE double az_atan2_m(double y, double x, const M &m)
{
    const double ax = fabs(x), ay = fabs(y);
    const double mx = fmax(ax, ay), mn = fmin(ax, ay);
    const double a = (mx > 0.0) ? mn * az_rcp(mx) : 0.0;
    const bool hi = a > m.mc(MC_AT_T3), mid = a > m.mc(MC_AT_T1); // tan(3 pi/16), tan(pi/16)
    const double c = hi ? 1.0 : (mid ? m.mc(MC_AT_TK) : 0.0);     // tan(k pi/8)
    const double off = hi ? m.mc(MC_AT_PI4) : (mid ? m.mc(MC_AT_PI8) : 0.0);
    const double z = (a - c) * az_rcp(fma(a, c, 1.0));
    const double u = z * z;
    double p = fma(u, m.mc(MC_AT_P0), m.mc(MC_AT_P1));
    p = fma(u, p, m.mc(MC_AT_P2));
    p = fma(u, p, m.mc(MC_AT_P3));
    p = fma(u, p, m.mc(MC_AT_P4));
    p = fma(u, p, m.mc(MC_AT_P5));
    p = fma(u, p, m.mc(MC_AT_P6));
    double r = fma(z, p, off);
    if (ay > ax) r = m.mc(MC_PIO2) - r;
    if (x < 0.0) r = m.mc(MC_PI) - r;
    return (y < 0.0) ? -r : r;
}
AZ_DEVICE double az_atan2(double y, double x) { return az_atan2_m(y, x, McLit()); }

// positive modulus (only the deep-space path and GMST need an explicit reduced angle)
AZ_DEVICE double az_mod2pi(double x)
{
    double r = fma(-floor(x * (1.0 / AZ_TWOPI)), AZ_TWOPI, x);
    if (r < 0.0) r += AZ_TWOPI;
    if (r >= AZ_TWOPI) r -= AZ_TWOPI;
    return r;
}
