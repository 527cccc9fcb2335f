// fast_step_f32.h -- the branch-free uniform-grid step (fast_step.h) for fp32 OUTPUTS, two grid points per lane
// (BASELINE config 5: 1M satellites x 10,000 steps, an HBM-bound stress case; the reference itself is fp64 only).
// Two steps: az_sgp4_fast_step_f32 (packed fp32 arithmetic with fp64 islands, opt-in: metres / mm/s; described first) and,
// at the end of the file, az_sgp4_fast_step_f32p (mixed precision, the default: every O(1) quantity in fp64, the small ones
// in packed fp32 -- the accuracy of fp32 storage itself).
//
// TWO grid points per lane.  On gfx950 a plain v_fma_f32 issues at the same rate as v_fma_f64 (one wave instruction
// per four cycles); the fp32 vector peak (157 TFLOP/s, twice the fp64 one) belongs to the PACKED forms v_pk_fma_f32 /
// v_pk_mul_f32 / v_pk_add_f32, which work on a register pair.  A lane therefore carries two adjacent grid points
// (2i, 2i+1) as the two halves of az_f2 values and the whole fp32 part of the step is written on az_f2: one
// instruction, two propagations.
//
// What stays fp64, per grid point: the carried pair of U = M + W (the along-track phase: hundreds of radians over
// 10,000 minutes, and a metre is 1.4e-7 rad; the even point's pair is carried, the odd point's is the even one
// rotated by the one-grid-step increment) and the along-radius chain a -> r (1 - e cos E and the J2 radius factor
// applied to the semi-major axis).  Everything else -- the pairs of M and W (they only enter through terms scaled by
// the eccentricity, < 0.004 here, and a wave re-seeds them from a full fp64 sincos at the start of its segment, at
// most 12 steps back), drag polynomials, Kepler step, short-period terms, orientation, velocity -- is packed fp32
// (v_rcp_f32 needs no refinement at this precision).  Near-circular members
// only; eccentric members and validation failures take the fp64 kernels with rounded stores.  Result accuracy is
// that of fp32 storage plus ~1e-7 relative from the fp32 chain: metres, mm/s
// (tests/test_gpu_round2.py::test_fp32_arithmetic_*).
#pragma once
#include "fast_step.h"

#ifdef AZ_HOST_EMUL
struct az_f2 { float x, y; };
static inline az_f2 operator+(az_f2 a, az_f2 b) { return {a.x + b.x, a.y + b.y}; }
static inline az_f2 operator-(az_f2 a, az_f2 b) { return {a.x - b.x, a.y - b.y}; }
static inline az_f2 operator*(az_f2 a, az_f2 b) { return {a.x * b.x, a.y * b.y}; }
static inline az_f2 operator*(float a, az_f2 b) { return {a * b.x, a * b.y}; }
static inline az_f2 operator*(az_f2 a, float b) { return {a.x * b, a.y * b}; }
static inline az_f2 operator-(az_f2 a) { return {-a.x, -a.y}; }
static inline az_f2 az_fma2(az_f2 a, az_f2 b, az_f2 c) { return {fmaf(a.x, b.x, c.x), fmaf(a.y, b.y, c.y)}; }
static inline float az_rcp32(float x) { return 1.0f / x; }
#else
typedef float az_f2 __attribute__((ext_vector_type(2)));
AZ_DEVICE az_f2 az_fma2(az_f2 a, az_f2 b, az_f2 c) { return __builtin_elementwise_fma(a, b, c); }
AZ_DEVICE float az_rcp32(float x) { return __builtin_amdgcn_rcpf(x); }
#endif
AZ_DEVICE az_f2 az_splat2(float a) { az_f2 r; r.x = a; r.y = a; return r; }
AZ_DEVICE az_f2 az_fma2(float a, az_f2 b, az_f2 c) { return az_fma2(az_splat2(a), b, c); }
AZ_DEVICE az_f2 az_fma2(az_f2 a, az_f2 b, float c) { return az_fma2(a, b, az_splat2(c)); }
AZ_DEVICE az_f2 az_fma2(float a, az_f2 b, float c) { return az_fma2(az_splat2(a), b, az_splat2(c)); }
AZ_DEVICE az_f2 az_rcp2(az_f2 a) { az_f2 r; r.x = az_rcp32(a.x); r.y = az_rcp32(a.y); return r; }
AZ_DEVICE az_f2 az_cvt2(double a, double b) { az_f2 r; r.x = (float)a; r.y = (float)b; return r; }

// per-satellite constants of the packed step.  Used once per step (LDS candidates) / several times (registers):
#define AZ_F32_ONCE(X) \
    X(cc1) X(d2) X(d3) X(d4) X(nl2) X(nl3) X(nl4) X(nl5) X(eta) X(omgcof) X(xmcof) X(xd) X(bc4) X(bc5) X(ecb) X(aycof) \
    X(xlcof) X(xnodcf) X(k_mrt) X(k_c2u) X(k_su) X(k_node) X(k_inc) X(k_rv) X(nodedot) X(sinio) X(cosio) X(sOc) X(cOc) \
    X(udot)
#define AZ_F32_MANY(X) X(x1mth2) X(sdA32) X(cdA32) X(sdW32) X(cdW32) X(step1) X(inv_sab32) X(abase32) X(rv0_32)
enum Fast32Once {
#define X(n) F32_##n,
    AZ_F32_ONCE(X)
#undef X
    F32_NUM
};
struct FastK32Doubles {
    double sab64;                       // sqrt(a_base), head of the along-radius chain
    double sdU, cdU, tc, tmid;          // the increment of U over one lane step, window centres
    double s1U, c1U;                    // the increment of U over ONE grid step (even -> odd point)
    // the mixed-precision step (az_sgp4_fast_step_f32p) keeps the orientation chain in fp64: its starting pairs
    double abase_km64, rv0_64, sOc64, cOc64, sinio64, cosio64; // a_base radius_km, vkmpersec / sqrt(a_base), starting pairs
};
// everything in registers (host emulation, set-up)
struct FastK32 : FastK32Doubles {
#define X(n) float n##_;
    AZ_F32_ONCE(X) AZ_F32_MANY(X)
#undef X
#define X(n) AZ_MEMBER float n() const { return n##_; }
    AZ_F32_ONCE(X) AZ_F32_MANY(X)
#undef X
};
// k_rows_fast32 (one satellite per wave): the once-per-step constants are LDS words read by all lanes at once (a
// packed instruction takes ONE scalar operand; every further constant would have to live in a VGPR for the whole
// loop -- 29 of them are the difference between 4 and 5 waves per SIMD)
struct FastK32Bcast : FastK32Doubles {
    const float *once;
#define X(n) float n##_;
    AZ_F32_MANY(X)
#undef X
#define X(n) AZ_MEMBER float n() const { return n##_; }
    AZ_F32_MANY(X)
#undef X
#define X(n) AZ_MEMBER float n() const { return once[F32_##n]; }
    AZ_F32_ONCE(X)
#undef X
};

// carried state of one lane: M and W pairs of both grid points in fp32, the even point's U pair in fp64
struct FastCarry32 {
    az_f2 sA, cA, sW, cW;
    double sU, cU;
};

// k: constants with the increments of one LANE step (two grid points x 64 lanes = 128 grid steps);
// k1: the same satellite's constants with the increments of ONE grid step; step1: the grid step in minutes
AZ_DEVICE void az_load_fast32(const FastK &k, const FastK &k1, double step1, const AzGrav &g, FastK32 &f)
{
#define X(n) f.n##_ = (float)k.n##_;
    AZ_F32_ONCE(X)
#undef X
    f.x1mth2_ = (float)k.x1mth2_;
    f.sab64 = k.sab_;
    f.sdU = k.sdU_; f.cdU = k.cdU_; f.tc = k.tc_; f.tmid = k.tmid_;
    f.s1U = k1.sdU_; f.c1U = k1.cdU_;
    {
        const double inv_sab = 1.0 / k.sab_, abase = k.sab_ * k.sab_;
        f.abase_km64 = abase * g.radius_km; f.rv0_64 = g.vkmpersec * inv_sab;
        f.inv_sab32_ = (float)inv_sab; f.abase32_ = (float)abase; f.rv0_32_ = (float)f.rv0_64;
    }
    f.sOc64 = k.sOc_; f.cOc64 = k.cOc_; f.sinio64 = k.sinio_; f.cosio64 = k.cosio_;
    f.sdA32_ = (float)k.sdA_; f.cdA32_ = (float)k.cdA_; f.sdW32_ = (float)k.sdW_; f.cdW32_ = (float)k.cdW_;
    f.step1_ = (float)step1;
}

// st: az_seed_fast's pairs for the even point (one lane step before its first grid point); k1: the one-grid-step
// increments.  The odd point's M and W pairs are the even ones rotated by one grid step, in fp64, once.
AZ_DEVICE void az_seed_fast32(const FastCarry &st, const FastK &k1, FastCarry32 &f)
{
    f.sA = az_cvt2(st.sA, fma(st.sA, k1.cdA_, st.cA * k1.sdA_));
    f.cA = az_cvt2(st.cA, fma(st.cA, k1.cdA_, -(st.sA * k1.sdA_)));
    f.sW = az_cvt2(st.sW, fma(st.sW, k1.cdW_, st.cW * k1.sdW_));
    f.cW = az_cvt2(st.cW, fma(st.cW, k1.cdW_, -(st.sW * k1.sdW_)));
    f.sU = st.sU;
    f.cU = st.cU;
}

// the increments of twice the angle (k_prep_inc holds 64 grid steps; a lane of the packed kernel advances by 128)
AZ_DEVICE void az_double_increments(FastK &k)
{
    const double sA = 2.0 * k.sdA_ * k.cdA_, cA = fma(-2.0 * k.sdA_, k.sdA_, 1.0);
    const double sW = 2.0 * k.sdW_ * k.cdW_, cW = fma(-2.0 * k.sdW_, k.sdW_, 1.0);
    k.sdA_ = sA; k.cdA_ = cA; k.sdW_ = sW; k.cdW_ = cW;
}

// (s,c) <- rotated by d, |d| <= 2^-10: sin d = d, cos d - 1 = -d^2/2 to fp32 precision (d^3/6 < 1.6e-10)
AZ_DEVICE void az_rot32_tiny(az_f2 &s, az_f2 &c, az_f2 d)
{
    const az_f2 q = (-0.5f * d) * d;
    const az_f2 ns = az_fma2(c, d, az_fma2(s, q, s));
    c = az_fma2(-s, d, az_fma2(c, q, c));
    s = ns;
}
// |d| <= 1/8: sin to d^5, cos to d^6 (next terms 4.5e-11, 1.5e-12)
AZ_DEVICE void az_rot32_med(az_f2 &s, az_f2 &c, az_f2 d)
{
    const az_f2 d2 = d * d;
    const az_f2 p = d * az_fma2(d2, az_fma2(1.0f / 120.0f, d2, -1.0f / 6.0f), 1.0f);
    const az_f2 q = d2 * az_fma2(d2, az_fma2(-1.0f / 720.0f, d2, 1.0f / 24.0f), -0.5f);
    const az_f2 ns = az_fma2(c, p, az_fma2(s, q, s));
    c = az_fma2(-s, p, az_fma2(c, q, c));
    s = ns;
}
// |d| <= 2^-7: sin to d^3, cos to d^4 (next terms 2.4e-13, 3e-16)
AZ_DEVICE void az_rot32_small(az_f2 &s, az_f2 &c, az_f2 d)
{
    const az_f2 d2 = d * d;
    const az_f2 p = az_fma2((-1.0f / 6.0f) * d2, d, d);
    const az_f2 q = d2 * az_fma2(1.0f / 24.0f, d2, -0.5f);
    const az_f2 ns = az_fma2(c, p, az_fma2(s, q, s));
    c = az_fma2(-s, p, az_fma2(c, q, c));
    s = ns;
}

// One lane step = the two grid points ta and ta + step1.  st: the carried pairs (advanced here by one lane step
// first, as in az_sgp4_fast_step).  r, v: component j of the even / odd point in r[j].x / r[j].y.
// Only inside a window az_fast_window_ok<false> accepted (fast_step.h: the bounds are on the fp64 quantities; the fp32
// ones differ from them by roundings, and no tier has a cliff at its threshold).
// DELTA (quasi-uniform grids, fast_step.h): dl = the two grid points' deviations from the ideal grid, minutes; only the carried
// phase U needs them at fp32 output precision (udot dl ~ 3e-8 rad = 0.2 m; t itself does not resolve 4e-7 min in fp32).
template <bool VEL, bool DELTA = false, class K>
AZ_DEVICE void az_sgp4_fast_step_f32(const K &k, const AzGrav &g, double ta, FastCarry32 &st, az_f2 r[3], az_f2 v[3], az_f2 dl = az_f2())
{
    {
        const az_f2 nsA = az_fma2(k.cdA32(), st.sA, k.sdA32() * st.cA);
        st.cA = az_fma2(k.cdA32(), st.cA, -(k.sdA32() * st.sA));
        st.sA = nsA;
        const az_f2 nsW = az_fma2(k.cdW32(), st.sW, k.sdW32() * st.cW);
        st.cW = az_fma2(k.cdW32(), st.cW, -(k.sdW32() * st.sW));
        st.sW = nsW;
        // U: fp64 (the phase is hundreds of radians; a per-step fp32 rounding would walk away)
        const double nsU = fma(st.sU, k.cdU, st.cU * k.sdU);
        st.cU = fma(st.cU, k.cdU, -(st.sU * k.sdU));
        st.sU = nsU;
    }
    az_f2 lane01;
    lane01.x = 0.0f;
    lane01.y = k.step1();
    const az_f2 t = az_splat2((float)ta) + lane01;
    const az_f2 dtc = az_splat2((float)(ta - k.tc)) + lane01;
    const az_f2 t2 = t * t;

    const az_f2 dm = az_fma2(k.eta(), st.cA, 1.0f);
    const az_f2 th = az_fma2(k.xmcof(), dm * dm * dm, az_fma2(k.omgcof(), t, -k.xd()));
    // 1 - tempa = t (cc1 + t (d2 + t (d3 + t d4))) is small (1e-3 after a week): fp32 is plenty for it
    const az_f2 dev = t * az_fma2(t, az_fma2(t, az_fma2(k.d4(), t, k.d3()), k.d2()), k.cc1());
    az_f2 nl = az_fma2(k.nl2() * dtc, dtc, t2 * (t * az_fma2(t, az_fma2(k.nl5(), t, k.nl4()), k.nl3())));
    if constexpr (DELTA) nl = az_fma2(k.udot(), dl, nl); // U at the actual time (fast_step.h): rides on the rotation by eps below
    // M + th and W - th: both pairs only enter through eccentricity-scaled terms (x 0.004), |th| <= 1/16:
    // sin th = th - th^3/6 (error 8e-9), cos th = 1 - th^2/2 (error 6e-7 x 0.004), one polynomial for both
    const az_f2 th2 = th * th;
    const az_f2 pth = az_fma2((-1.0f / 6.0f) * th2, th, th);
    const az_f2 qth = -0.5f * th2;
    const az_f2 smm = az_fma2(st.cA, pth, az_fma2(st.sA, qth, st.sA));
    const az_f2 sw = az_fma2(-st.cW, pth, az_fma2(st.sW, qth, st.sW));
    const az_f2 cw = az_fma2(st.sW, pth, az_fma2(st.cW, qth, st.cW));
    az_f2 em = az_fma2(-k.bc5(), smm, az_fma2(-k.bc4(), t, k.ecb()));
    em.x = fmaxf(em.x, 1.0e-6f);
    em.y = fmaxf(em.y, 1.0e-6f);

    // along-radius chain in fp64: sqrt(am) = sqrt(a_base) |tempa|
    const double sqrt_am_a = fabs(fma(-k.sab64, (double)dev.x, k.sab64)), sqrt_am_b = fabs(fma(-k.sab64, (double)dev.y, k.sab64));
    const az_f2 sqrt_am = az_cvt2(sqrt_am_a, sqrt_am_b);
    const az_f2 omem2 = az_fma2(-em, em, 1.0f);
    const az_f2 R = az_rcp2(sqrt_am * omem2);
    const az_f2 ra = R * omem2;
    const az_f2 temp = ra * R;

    const az_f2 axnl = em * cw;
    const az_f2 aynl = az_fma2(em, sw, k.aycof() * temp);
    // u0 = U (carried in fp64) + the rest of the drag term + the long-period term
    az_f2 s = az_cvt2(st.sU, fma(st.sU, k.c1U, st.cU * k.s1U));
    az_f2 c = az_cvt2(st.cU, fma(st.cU, k.c1U, -(st.sU * k.s1U)));
    {
        const az_f2 eps = az_fma2(k.xlcof() * temp, axnl, nl);
        az_rot32_med(s, c, eps);
    }

    // Kepler, near-circular: one Newton step from E0 = u (the next correction, (el/2) d0^2 <= 3.2e-8, is below
    // fp32 resolution)
    const az_f2 el2 = az_fma2(axnl, axnl, aynl * aynl);
    const az_f2 rden = az_rcp2(az_fma2(-s, aynl, az_fma2(-c, axnl, 1.0f)));
    const az_f2 d0 = az_fma2(axnl, s, -(aynl * c)) * rden;
    az_rot32_small(s, c, d0);
    const az_f2 ecose = az_fma2(axnl, c, aynl * s);
    const az_f2 esine = az_fma2(axnl, s, -(aynl * c));
    const az_f2 inv_ome = az_rcp2(az_splat2(1.0f) - ecose);
    const az_f2 betal = az_fma2(-0.5f, el2, 1.0f);
    const az_f2 inv_omel2 = az_splat2(1.0f) + el2;
    const az_f2 inv_1pb = az_fma2(0.125f, el2, 0.5f);

    const az_f2 est = esine * inv_1pb;
    const az_f2 sinu = inv_ome * (s - az_fma2(axnl, est, aynl));
    const az_f2 cosu = inv_ome * (c + az_fma2(aynl, est, -axnl));
    const az_f2 sin2u = (sinu + sinu) * cosu;
    const az_f2 cos2u = az_fma2(-2.0f * sinu, sinu, 1.0f);

    const az_f2 inv_am = ra * ra;
    const az_f2 inv_pl = inv_am * inv_omel2;
    const az_f2 temp1 = (float)g.half_j2 * inv_pl;
    const az_f2 temp2 = temp1 * inv_pl;

    // mrt = rl (1 + k_mrt temp2 betal) + k_c2u temp1 cos2u, rl = am (1 - ecose): fp64 (cancellation-free, but the
    // result is the radius itself: an fp32 chain here costs a metre); the two corrections are small and enter as fp32
    const az_f2 fm1 = k.k_mrt() * temp2 * betal;
    const az_f2 add = k.k_c2u() * temp1 * cos2u;
    const double rl_a = sqrt_am_a * sqrt_am_a * (1.0 - (double)ecose.x);
    const double rl_b = sqrt_am_b * sqrt_am_b * (1.0 - (double)ecose.y);
    const double mrt_a = fma(rl_a, (double)fm1.x, rl_a + (double)add.x);
    const double mrt_b = fma(rl_b, (double)fm1.y, rl_b + (double)add.y);
    const az_f2 rs = az_cvt2(mrt_a * g.radius_km, mrt_b * g.radius_km);
    const az_f2 t2s = temp2 * sin2u;
    const az_f2 a_nd = az_fma2(k.k_node(), t2s, az_fma2(k.nodedot(), az_splat2((float)(ta - k.tmid)) + lane01, k.xnodcf() * t2));
    az_f2 ssu = sinu, csu = cosu, sn = az_splat2(k.sOc()), cn = az_splat2(k.cOc()), si = az_splat2(k.sinio()),
          ci = az_splat2(k.cosio());
    az_rot32_tiny(ssu, csu, k.k_su() * t2s);
    az_rot32_med(sn, cn, a_nd);
    az_rot32_tiny(si, ci, k.k_inc() * temp2 * cos2u);

    const az_f2 xmx = -sn * ci, xmy = cn * ci;
    const az_f2 ux = az_fma2(xmx, ssu, cn * csu);
    const az_f2 uy = az_fma2(xmy, ssu, sn * csu);
    const az_f2 uz = si * ssu;
    r[0] = rs * ux;
    r[1] = rs * uy;
    r[2] = rs * uz;
    if (VEL) {
        const az_f2 rv = (float)g.vkmpersec * ra;
        const az_f2 vk = rv * inv_ome;
        const az_f2 nxt = rv * inv_am * temp1;
        const az_f2 mvt = az_fma2(-k.x1mth2() * nxt, sin2u, vk * esine);
        const az_f2 rvdot = az_fma2(nxt, az_fma2(k.x1mth2(), cos2u, k.k_rv()), vk * betal);
        const az_f2 vx = az_fma2(xmx, csu, -(cn * ssu));
        const az_f2 vy = az_fma2(xmy, csu, -(sn * ssu));
        const az_f2 vz = si * csu;
        v[0] = az_fma2(mvt, ux, rvdot * vx);
        v[1] = az_fma2(mvt, uy, rvdot * vy);
        v[2] = az_fma2(mvt, uz, rvdot * vz);
    }
}


// ---------------------------------------------------------------------------------------------------------------------
// The MIXED-precision step: fp32 outputs at (nearly) the accuracy of fp64 arithmetic rounded once at the store, for two
// thirds of its instruction slots.  The packed step above loses its metres and mm/s in the O(1) quantities: every fp32
// rounding of a unit-vector component or of the 7.5 km/s speed factor is 0.4 m / 0.45 mm/s, and the chain from the U pair
// to the velocity has a dozen of them.  Here every O(1) quantity stays fp64, per grid point -- the U pair and each
// rotation APPLIED to it, sin/cos u, the three short-period rotations of (u, node, inclination), the orientation
// products, the radius chain, the speed factors -- while everything SMALL is still computed once for both grid points
// in packed fp32 and converted where it meets an O(1) value: the secular drag terms, the eccentricity vector (x 0.004),
// the Kepler corrections, the (p,q) = (sin d, cos d - 1) of every rotation angle (|d| <= 1/8: an fp32 p is good to
// 7.5e-9 rad = 5 cm), 1/(1 - e cos E) - 1 and 1/sqrt(am) - 1/sqrt(a_base) as short series, the J2 factors.
// Near-circular members inside a window az_fast_window_ok<false> accepted AND whose drag deviation 1 - tempa stays
// below 2^-6 over it (the 1/(1 - dev) series; the plan checks it).
AZ_DEVICE void az_pq32_med(az_f2 d, az_f2 &p, az_f2 &q)
{
    const az_f2 d2 = d * d;
    p = d * az_fma2(d2, az_fma2(1.0f / 120.0f, d2, -1.0f / 6.0f), 1.0f);
    q = d2 * az_fma2(d2, az_fma2(-1.0f / 720.0f, d2, 1.0f / 24.0f), -0.5f);
}
// (s,c) in fp64 rotated by the fp32 pair (p,q)
AZ_DEVICE void az_rot_apply_mixed(double &s, double &c, float p, float q)
{
    const double pd = (double)p, qd = (double)q;
    const double ns = fma(c, pd, fma(s, qd, s));
    c = fma(-s, pd, fma(c, qd, c));
    s = ns;
}
#define AZ_F32P_DEV_MAX 0.015625 /* 1 - tempa: dev^5 < 1e-9 */

// the extra bound of the mixed step over a window [t_a, t_b] (any order): 1 - tempa <= 2^-6
template <class K>
AZ_DEVICE bool az_fast32p_window_ok(const K &k, double t_a, double t_b)
{
    const double T = fmax(fabs(t_a), fabs(t_b));
    const double da = T * fma(T, fma(T, fma(T, fabs(k.d4_), fabs(k.d3_)), fabs(k.d2_)), fabs(k.cc1_));
    return da <= AZ_F32P_DEV_MAX;
}

template <bool VEL, bool DELTA = false, class K>
AZ_DEVICE void az_sgp4_fast_step_f32p(const K &k, const AzGrav &g, double ta, FastCarry32 &st, az_f2 r[3], az_f2 v[3], az_f2 dl = az_f2())
{
    {
        const az_f2 nsA = az_fma2(k.cdA32(), st.sA, k.sdA32() * st.cA);
        st.cA = az_fma2(k.cdA32(), st.cA, -(k.sdA32() * st.sA));
        st.sA = nsA;
        const az_f2 nsW = az_fma2(k.cdW32(), st.sW, k.sdW32() * st.cW);
        st.cW = az_fma2(k.cdW32(), st.cW, -(k.sdW32() * st.sW));
        st.sW = nsW;
        const double nsU = fma(st.sU, k.cdU, st.cU * k.sdU);
        st.cU = fma(st.cU, k.cdU, -(st.sU * k.sdU));
        st.sU = nsU;
    }
    az_f2 lane01;
    lane01.x = 0.0f;
    lane01.y = k.step1();
    const az_f2 t = az_splat2((float)ta) + lane01;
    const az_f2 dtc = az_splat2((float)(ta - k.tc)) + lane01;
    const az_f2 t2 = t * t;

    // ---- small quantities, packed fp32 (as in az_sgp4_fast_step_f32)
    const az_f2 dm = az_fma2(k.eta(), st.cA, 1.0f);
    const az_f2 th = az_fma2(k.xmcof(), dm * dm * dm, az_fma2(k.omgcof(), t, -k.xd()));
    const az_f2 dev = t * az_fma2(t, az_fma2(t, az_fma2(k.d4(), t, k.d3()), k.d2()), k.cc1());
    az_f2 nl = az_fma2(k.nl2() * dtc, dtc, t2 * (t * az_fma2(t, az_fma2(k.nl5(), t, k.nl4()), k.nl3())));
    if constexpr (DELTA) nl = az_fma2(k.udot(), dl, nl); // U at the actual time: rides on the rotation by eps below
    const az_f2 th2 = th * th;
    const az_f2 pth = az_fma2((-1.0f / 6.0f) * th2, th, th);
    const az_f2 qth = -0.5f * th2;
    const az_f2 smm = az_fma2(st.cA, pth, az_fma2(st.sA, qth, st.sA));
    const az_f2 sw = az_fma2(-st.cW, pth, az_fma2(st.sW, qth, st.sW));
    const az_f2 cw = az_fma2(st.sW, pth, az_fma2(st.cW, qth, st.cW));
    az_f2 em = az_fma2(-k.bc5(), smm, az_fma2(-k.bc4(), t, k.ecb()));
    em.x = fmaxf(em.x, 1.0e-6f);
    em.y = fmaxf(em.y, 1.0e-6f);

    // am = a_base (1 - dev)^2; 1/sqrt(am) = (1/sqrt(a_base)) (1 + c1), c1 = dev + dev^2 + dev^3 + dev^4.  Both enter the
    // results as a per-satellite fp64 constant plus a SMALL fp32 correction (below): a_base, vkmpersec / sqrt(a_base)
    const az_f2 c1 = dev * az_fma2(dev, az_fma2(dev, az_fma2(dev, az_splat2(1.0f), 1.0f), 1.0f), 1.0f);
    const az_f2 ra = az_fma2(k.inv_sab32(), c1, k.inv_sab32());
    const az_f2 omd = az_splat2(1.0f) - dev;
    const az_f2 am = (k.abase32() * omd) * omd;
    const az_f2 omem2 = az_fma2(-em, em, 1.0f);
    const az_f2 inv_am = ra * ra;
    const az_f2 temp = inv_am * az_rcp2(omem2);

    const az_f2 axnl = em * cw;
    const az_f2 aynl = az_fma2(em, sw, k.aycof() * temp);

    // ---- the U pairs of the two grid points, fp64; fp32 copies carry the chain U -> E -> u for the SMALL terms, and the
    // angles of that chain are summed: ONE rotation takes each fp64 pair from U to u + (short-period correction)
    const double sU_a = st.sU, cU_a = st.cU;
    const double sU_b = fma(st.sU, k.c1U, st.cU * k.s1U), cU_b = fma(st.cU, k.c1U, -(st.sU * k.s1U));
    az_f2 s = az_cvt2(sU_a, sU_b), c = az_cvt2(cU_a, cU_b);
    const az_f2 eps = az_fma2(k.xlcof() * temp, axnl, nl);
    az_rot32_med(s, c, eps);
    // Kepler, near-circular: Newton step from E0 = u, then the chord step with the same reciprocal (first order)
    const az_f2 el2 = az_fma2(axnl, axnl, aynl * aynl);
    const az_f2 rden = az_rcp2(az_fma2(-s, aynl, az_fma2(-c, axnl, 1.0f)));
    const az_f2 d0 = az_fma2(axnl, s, -(aynl * c)) * rden;
    az_rot32_small(s, c, d0);
    const az_f2 d1 = az_fma2(axnl, s, az_fma2(-aynl, c, -d0)) * rden; // <= (el/2) d0^2 = 3.4e-8 rad: a quarter of a metre
    {
        const az_f2 ns = az_fma2(c, d1, s);
        c = az_fma2(-s, d1, c);
        s = ns;
    }
    const az_f2 ecose = az_fma2(axnl, c, aynl * s);
    const az_f2 esine = az_fma2(axnl, s, -(aynl * c));
    // 1/(1 - ecose) = 1 + y, y = w + w^2 + w^3 + w^4, |w| <= 0.0041 (w^5 < 1.2e-12)
    const az_f2 y = ecose * az_fma2(ecose, az_fma2(ecose, az_fma2(ecose, az_splat2(1.0f), 1.0f), 1.0f), 1.0f);
    const az_f2 bm1 = el2 * az_fma2(-0.125f, el2, -0.5f); // betal - 1
    const az_f2 inv_omel2 = az_splat2(1.0f) + el2;
    const az_f2 inv_1pb = az_fma2(0.125f, el2, 0.5f);
    const az_f2 est = esine * inv_1pb;
    const az_f2 tA = az_fma2(axnl, est, aynl), tB = az_fma2(aynl, est, -axnl);
    // (sin u, cos u) = (s - tA, c + tB) / (1 - ecose) is (s, c) rotated by u - E: sin(u - E) = -(tA c + tB s)(1 + y)
    // (a few 1e-3: the fp32 value is good to 5e-10 rad); the fp32 copies feed sin 2u / cos 2u, which only scale J2 terms
    const az_f2 p2 = -(az_fma2(tA, c, tB * s)) * (az_splat2(1.0f) + y);
    const az_f2 d_ue = az_fma2((1.0f / 6.0f) * p2 * p2, p2, p2);     // asin(p2), |p2| <= 0.0082
    {
        const az_f2 ds = s - tA, dc = c + tB;
        s = az_fma2(ds, y, ds);
        c = az_fma2(dc, y, dc);
    }
    const az_f2 sin2u = (s + s) * c;
    const az_f2 cos2u = az_fma2(-2.0f * s, s, 1.0f);

    const az_f2 inv_pl = inv_am * inv_omel2;
    const az_f2 temp1 = (float)g.half_j2 * inv_pl;
    const az_f2 temp2 = temp1 * inv_pl;
    const az_f2 t2s = temp2 * sin2u;
    // radius = a_base + [am - a_base] - am ecose + rl k_mrt temp2 betal + k_c2u temp1 cos2u (earth radii): the fp64 constant
    // plus corrections of at most 3e-2, summed in fp32
    az_f2 rcorr;
    {
        const az_f2 rl = az_fma2(-am, ecose, am);
        const az_f2 fm1 = k.k_mrt() * temp2 * (az_splat2(1.0f) + bm1);
        rcorr = az_fma2(rl, fm1, (k.k_c2u() * temp1) * cos2u);
        rcorr = az_fma2(-am, ecose, rcorr);
        rcorr = az_fma2(k.abase32() * dev, dev - az_splat2(2.0f), rcorr);
        rcorr = rcorr * (float)g.radius_km;
    }
    const double rs_a = k.abase_km64 + (double)rcorr.x, rs_b = k.abase_km64 + (double)rcorr.y;

    // total rotation of the U pair: drag / long-period term, the two Kepler corrections, u - E, the J2 correction of u
    az_f2 p, q;
    {
        const az_f2 theta = eps + ((d0 + d1) + (d_ue + k.k_su() * t2s));
        az_pq32_med(theta, p, q);
    }
    double ssu_a = sU_a, csu_a = cU_a, ssu_b = sU_b, csu_b = cU_b;
    az_rot_apply_mixed(ssu_a, csu_a, p.x, q.x);
    az_rot_apply_mixed(ssu_b, csu_b, p.y, q.y);

    // node (with its motion across the window) and inclination: per-window / per-satellite fp64 pairs rotated by small angles
    const az_f2 a_nd = az_fma2(k.k_node(), t2s, az_fma2(k.nodedot(), az_splat2((float)(ta - k.tmid)) + lane01, k.xnodcf() * t2));
    double sn_a = k.sOc64, cn_a = k.cOc64, sn_b = k.sOc64, cn_b = k.cOc64;
    az_pq32_med(a_nd, p, q);
    az_rot_apply_mixed(sn_a, cn_a, p.x, q.x);
    az_rot_apply_mixed(sn_b, cn_b, p.y, q.y);
    double si_a = k.sinio64, ci_a = k.cosio64, si_b = k.sinio64, ci_b = k.cosio64;
    {
        const az_f2 d = k.k_inc() * temp2 * cos2u;     // <= 9e-4: sin d = d, cos d - 1 = -d^2/2 (d^3/6 < 1.3e-10)
        const az_f2 qq = (-0.5f * d) * d;
        az_rot_apply_mixed(si_a, ci_a, d.x, qq.x);
        az_rot_apply_mixed(si_b, ci_b, d.y, qq.y);
    }
    const double xmx_a = -sn_a * ci_a, xmy_a = cn_a * ci_a, xmx_b = -sn_b * ci_b, xmy_b = cn_b * ci_b;
    const double ux_a = fma(xmx_a, ssu_a, cn_a * csu_a), uy_a = fma(xmy_a, ssu_a, sn_a * csu_a), uz_a = si_a * ssu_a;
    const double ux_b = fma(xmx_b, ssu_b, cn_b * csu_b), uy_b = fma(xmy_b, ssu_b, sn_b * csu_b), uz_b = si_b * ssu_b;
    r[0] = az_cvt2(rs_a * ux_a, rs_b * ux_b);
    r[1] = az_cvt2(rs_a * uy_a, rs_b * uy_b);
    r[2] = az_cvt2(rs_a * uz_a, rs_b * uz_b);
    if (VEL) {
        // rvdot = rv (1 + y)(1 + bm1) + nxt (x1mth2 cos2u + k_rv), rv = (vkmpersec / sqrt(a_base)) (1 + c1): the fp64 constant
        // plus corrections of a few per cent, summed in fp32
        const az_f2 rv32 = az_fma2(k.rv0_32(), c1, k.rv0_32());
        const az_f2 zz = az_fma2(y, bm1, y + bm1);
        const az_f2 nxt = rv32 * inv_am * temp1;
        az_f2 vcorr = az_fma2(rv32, zz, nxt * az_fma2(k.x1mth2(), cos2u, k.k_rv()));
        vcorr = az_fma2(k.rv0_32(), c1, vcorr);
        const az_f2 mvt = az_fma2(-k.x1mth2() * nxt, sin2u, rv32 * (az_splat2(1.0f) + y) * esine); // <= 0.03 km/s
        const double rvdot_a = k.rv0_64 + (double)vcorr.x, rvdot_b = k.rv0_64 + (double)vcorr.y;
        const double mvt_a = (double)mvt.x, mvt_b = (double)mvt.y;
        const double vx_a = fma(xmx_a, csu_a, -(cn_a * ssu_a)), vy_a = fma(xmy_a, csu_a, -(sn_a * ssu_a)), vz_a = si_a * csu_a;
        const double vx_b = fma(xmx_b, csu_b, -(cn_b * ssu_b)), vy_b = fma(xmy_b, csu_b, -(sn_b * ssu_b)), vz_b = si_b * csu_b;
        v[0] = az_cvt2(fma(mvt_a, ux_a, rvdot_a * vx_a), fma(mvt_b, ux_b, rvdot_b * vx_b));
        v[1] = az_cvt2(fma(mvt_a, uy_a, rvdot_a * vy_a), fma(mvt_b, uy_b, rvdot_b * vy_b));
        v[2] = az_cvt2(fma(mvt_a, uz_a, rvdot_a * vz_a), fma(mvt_b, uz_b, rvdot_b * vz_b));
    }
}
