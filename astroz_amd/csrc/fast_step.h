// fast_step.h -- branch-free near-earth step for uniform time grids ("optimistic" path).
//
// Same quantities as az_sgp4_step (Sgp4Batch.propagateBatchDirect, src/Sgp4Batch.zig L113-157, +
// Sgp4.keplerAndPosVel, src/Sgp4.zig L646-750), restricted to the case nine propagations in ten fall in:
//   * the time grid is uniform, so the three slowly drifting angles advance by the SAME increment
//     between a lane's consecutive steps: (sin,cos) of mdot*dt, argpdot*dt, nodedot*dt are
//     per-satellite constants (prepared once per grid by k_prep_inc) and every carried pair moves by
//     one angle addition (4 FMA-class instructions, no polynomial, no vote);
//   * every small angle sits inside its usual rotation tier, and the orbit is either near-circular
//     (el^2 < 1.6e-5: the two-step Kepler form) or, ECC = true, converges in five Newton trips with fixed tiers.
// Instead of choosing tiers with wave votes and branches the step runs straight through and RETURNS a
// per-lane `bad` predicate; the caller votes once per step and, on a violation, hands the remaining
// grid points to the generic az_sgp4_step loop (results are only stored after the vote, so nothing
// wrong ever reaches memory).  ~210 fp64 instructions per propagation against ~300 executed by the
// generic loop (whose tier votes cost compares, branches and the register moves at every merge).
#pragma once
#include "propagate_device.h"

// per-satellite constants of the fast step (wave-uniform in the lane = time kernel, per lane in the
// lane = satellite kernel).  Folded products differ from the table rows by one rounding at most.
// Names of the constants; the step reads them through accessor methods so that a kernel can decide where
// each one lives (registers, or one LDS word read by all lanes at once).
#define AZ_FASTK_COLD(X)                                                                                     \
    X(cc1) X(d2) X(d3) X(d4) X(nl2) X(nl3) X(nl4) X(nl5) X(eta) X(omgcof) X(xmcof) X(xd) X(bc4) X(bc5) X(ecb) X(sab)
#define AZ_FASTK_HOT(X)                                                                                      \
    X(aycof) X(xlcof) X(xnodcf) X(sinio) X(cosio) X(k_mrt) X(k_c2u) X(k_su) X(k_node) X(k_inc) X(x1mth2) X(k_rv) \
    X(sdA) X(cdA) X(sdW) X(cdW) X(sdO) X(cdO)
enum FastCold {
#define X(n) FC_##n,
    AZ_FASTK_COLD(X)
#undef X
    FC_NUM
};
// everything in registers (lane = satellite kernels, host emulation)
struct FastK {
#define X(n) double n##_;
    AZ_FASTK_COLD(X) AZ_FASTK_HOT(X)
#undef X
#define X(n) AZ_MEMBER double n() const { return n##_; }
    AZ_FASTK_COLD(X) AZ_FASTK_HOT(X)
#undef X
};
// lane = time kernels (one satellite per wave): the 18 constants used several times per step or sitting
// on the critical path stay in (scalar) registers, the 16 once-per-step ones are LDS words read by all
// lanes at once (broadcast ds_read: no VALU slot, no register residency)
struct FastKBcast {
    const double *cold;
#define X(n) double n##_;
    AZ_FASTK_HOT(X)
#undef X
#define X(n) AZ_MEMBER double n() const { return n##_; }
    AZ_FASTK_HOT(X)
#undef X
#define X(n) AZ_MEMBER double n() const { return cold[FC_##n]; }
    AZ_FASTK_COLD(X)
#undef X
};

// carried (sin,cos) pairs: M = mo + mdot t, W = argpo + argpdot t, O = nodeo + nodedot t (the
// xnodcf t^2 part of the node is folded into the J2 node correction, a tiny rotation anyway)
struct FastCarry {
    double sA, cA, sW, cW, sO, cO;
};

// row offsets of the increment table written by k_prep_inc: inc[(AZ_INC_* + 6*which) * n_pad + sat],
// which = 0: dt = 64 grid steps (lane = time kernels), 1: dt = one grid step (lane = satellite kernels)
enum { AZ_INC_sdA, AZ_INC_cdA, AZ_INC_sdW, AZ_INC_cdW, AZ_INC_sdO, AZ_INC_cdO, AZ_INC_NUM };

AZ_DEVICE void az_load_fast(const double *__restrict__ el, size_t n_pad, size_t i, unsigned flags,
                            const double *__restrict__ inc, int which, FastK &k)
{
#define L(f) el[(size_t)F_##f * n_pad + i]
    const bool ho = !(flags & AZ_FLAG_ISIMP);
    const double no = L(no_unkozai);
    k.cc1_ = L(cc1); k.d2_ = L(d2); k.d3_ = L(d3); k.d4_ = L(d4);
    k.nl2_ = no * L(t2cof); k.nl3_ = no * L(t3cof); k.nl4_ = no * L(t4cof); k.nl5_ = no * L(t5cof);
    k.eta_ = L(eta);
    k.omgcof_ = ho ? L(omgcof) : 0.0;
    k.xmcof_ = ho ? L(xmcof) : 0.0;
    k.xd_ = k.xmcof_ * L(delmo);
    k.bc4_ = L(bc4);
    k.bc5_ = ho ? L(bc5) : 0.0;
    k.ecb_ = fma(k.bc5_, L(sinmao), L(ecco));
    k.sab_ = L(sqrt_a_base);
    k.aycof_ = L(aycof); k.xlcof_ = L(xlcof); k.xnodcf_ = L(xnodcf);
    k.sinio_ = L(sinio); k.cosio_ = L(cosio);
    az_j2_factors(L(con41), L(x1mth2), L(x7thm1), k.sinio_, k.cosio_, k.k_mrt_, k.k_c2u_, k.k_su_, k.k_node_,
                  k.k_inc_, k.k_rv_);
    k.x1mth2_ = L(x1mth2);
    const double *q = inc + (size_t)(6 * which) * n_pad + i;
    k.sdA_ = q[(size_t)AZ_INC_sdA * n_pad]; k.cdA_ = q[(size_t)AZ_INC_cdA * n_pad];
    k.sdW_ = q[(size_t)AZ_INC_sdW * n_pad]; k.cdW_ = q[(size_t)AZ_INC_cdW * n_pad];
    k.sdO_ = q[(size_t)AZ_INC_sdO * n_pad]; k.cdO_ = q[(size_t)AZ_INC_cdO * n_pad];
#undef L
}

// seed the carried pairs with full sincos at time t (the step BEFORE the first one to be produced: every
// az_sgp4_fast_step call first advances the pairs by one increment)
AZ_DEVICE void az_seed_fast(const double *__restrict__ el, size_t n_pad, size_t i, double t, FastCarry &st)
{
#define L(f) el[(size_t)F_##f * n_pad + i]
    az_sincos(fma(L(mdot), t, L(mo)), st.sA, st.cA);
    az_sincos(fma(L(argpdot), t, L(argpo)), st.sW, st.cW);
    az_sincos(fma(L(nodedot), t, L(nodeo)), st.sO, st.cO);
#undef L
}

// (s,c) <- (sin,cos)(angle + d): two FMAs per component.  One more rounding than az_rot_apply's
// s + (s q + c p) form (1.1e-16 relative); used where the result is consumed, not carried.
AZ_DEVICE void az_rot_apply2(double &s, double &c, double p, double q)
{
    const double ns = fma(c, p, fma(s, q, s));
    c = fma(-s, p, fma(c, q, c));
    s = ns;
}
// |d| <= 2^-10: sin to d^3, cos to d^2 (dropped d^4/24 < 3.8e-14, d^5/120 < 8e-18)
AZ_DEVICE void az_rotate_tiny2(double &s, double &c, double d, const RotK &k)
{
    const double d2 = d * d;
    const double q = -0.5 * d2;
    const double p = fma(d2 * k.n6, d, d);
    az_rot_apply2(s, c, p, q);
}

// (p,q) = (sin d, cos d - 1) for |d| <= 1/16: sin to d^7, cos to d^8 (d^9/9! < 3e-17, d^10/10! < 3e-19).
// Two instructions more than the 2^-7 tier; used for delomg + delm, which reaches 2^-7 for one member in fifty.
#define AZ_ROT_16TH 0.0625
AZ_DEVICE void az_pq_16th(double d, const RotK &k, double &p, double &q)
{
    const double d2 = d * d;
    q = d2 * fma(d2, fma(d2, fma(d2, 1.0 / 40320.0, k.n720), k.p24), -0.5);
    p = d * fma(d2, fma(d2, fma(d2, -1.0 / 5040.0, k.p120), k.n6), 1.0);
}

// (p,q) for |d| <= 1/8: sin to d^9, cos to d^10 (the polynomial of az_rotate_med).  The along-track drag term
// no*templ grows with t^2: 1/8 rad holds 99.4% of a catalog up to six days from epoch.
AZ_DEVICE void az_pq_med(double d, double &p, double &q)
{
    const double d2 = d * d;
    q = fma(d2, -1.0 / 3628800.0, 1.0 / 40320.0);
    q = fma(d2, q, -1.0 / 720.0);
    q = fma(d2, q, 1.0 / 24.0);
    q = d2 * fma(d2, q, -0.5);
    p = fma(d2, 1.0 / 362880.0, -1.0 / 5040.0);
    p = fma(d2, p, 1.0 / 120.0);
    p = fma(d2, p, -1.0 / 6.0);
    p = d * fma(d2, p, 1.0);
}

// validation thresholds of the fast step
#define AZ_FAST_EL2 1.6e-5
#define AZ_FAST_TEMP2 6.0e-4

// one near-earth propagation on a uniform grid; returns true when an assumption of the fast path does
// not hold for this lane (the caller must then discard r/v and use az_sgp4_step)
template <bool VEL, bool ECC = false, class K = FastK>
AZ_DEVICE bool az_sgp4_fast_step(const K &k, const AzGrav &g, const RotK &rk, double t, FastCarry &st, double r[3],
                                  double v[3])
{
    // advance the carried pairs by their constant increments
    {
        const double nsA = fma(st.sA, k.cdA(), st.cA * k.sdA());
        st.cA = fma(st.cA, k.cdA(), -(st.sA * k.sdA()));
        st.sA = nsA;
        const double nsW = fma(st.sW, k.cdW(), st.cW * k.sdW());
        st.cW = fma(st.cW, k.cdW(), -(st.sW * k.sdW()));
        st.sW = nsW;
        const double nsO = fma(st.sO, k.cdO(), st.cO * k.sdO());
        st.cO = fma(st.cO, k.cdO(), -(st.sO * k.sdO()));
        st.sO = nsO;
    }
    const double sA = st.sA, cA = st.cA;
    const double t2 = t * t;

    // secular gravity + drag (Sgp4Batch.zig L121-154)
    const double dm = fma(k.eta(), cA, 1.0);
    const double th = fma(k.xmcof(), dm * dm * dm, fma(k.omgcof(), t, -k.xd())); // delomg + delm
    const double tempa = fma(-t, fma(t, fma(t, fma(t, k.d4(), k.d3()), k.d2()), k.cc1()), 1.0);
    const double nl = t2 * fma(t, fma(t, fma(t, k.nl5(), k.nl4()), k.nl3()), k.nl2()); // no_unkozai * templ
    bool bad = !(fabs(th) <= AZ_ROT_16TH);
    double p, q;
    az_pq_16th(th, rk, p, q);
    const double smm = fma(cA, p, fma(sA, q, sA));            // sin(M + th)
    const double sw = fma(-st.cW, p, fma(st.sW, q, st.sW));   // (sin,cos)(W - th)
    const double cw = fma(st.sW, p, fma(st.cW, q, st.cW));
    const double em = fmax(fma(-k.bc5(), smm, fma(-k.bc4(), t, k.ecb())), 1.0e-6);

    // am = a_base tempa^2: one reciprocal gives 1/sqrt(am) and 1/(am (1 - em^2))
    const double sqrt_am = k.sab() * fabs(tempa);
    const double am = sqrt_am * sqrt_am;
    const double omem2 = fma(-em, em, 1.0);
    const double R = az_rcp(sqrt_am * omem2);
    const double ra = R * omem2;
    const double temp = ra * R;

    const double axnl = em * cw;
    const double aynl = fma(em, sw, temp * k.aycof());
    // u0 = M + W + no*templ + temp*xlcof*axnl
    double s = fma(sA, st.cW, cA * st.sW);
    double c = fma(cA, st.cW, -(sA * st.sW));
    {
        const double eps = fma(temp * k.xlcof(), axnl, nl);
        bad |= !(fabs(eps) <= AZ_ROT_MED);
        az_pq_med(eps, p, q);
        az_rot_apply2(s, c, p, q);
    }

    const double el2 = fma(axnl, axnl, aynl * aynl);
    double ecose, esine, ome, inv_ome, betal, inv_omel2, inv_1pb;
    if (!ECC) {
        // Kepler, near-circular form (see az_kepler_posvel): Newton step from E0 = u, chord step with the
        // same reciprocal, first-order rotation by the second correction
        bad |= !(el2 <= AZ_FAST_EL2);
        const double rden = az_rcp1(fma(-s, aynl, fma(-c, axnl, 1.0)));
        const double d0 = fma(axnl, s, -(aynl * c)) * rden;
        az_pq_small(d0, rk, p, q); // |d0| <= el/(1-el) < 2^-7
        az_rot_apply2(s, c, p, q);
        const double d1 = fma(axnl, s, fma(-aynl, c, -d0)) * rden;
        {
            const double s1 = fma(c, d1, s);
            c = fma(-s, d1, c);
            s = s1;
        }
        ecose = fma(axnl, c, aynl * s);
        esine = fma(axnl, s, -(aynl * c));
        ome = 1.0 - ecose;
        inv_ome = fma(rden, fma(-ome, rden, 1.0), rden);
        inv_ome = fma(inv_ome, fma(-ome, inv_ome, 1.0), inv_ome);
        betal = fma(el2, fma(el2, -0.125, -0.5), 1.0);    // sqrt(1-x)   (- x^3/16)
        inv_omel2 = fma(el2, el2 + 1.0, 1.0);             // 1/(1-x)     (+ x^3 < 4.1e-15)
        inv_1pb = fma(el2, fma(el2, 0.0625, 0.125), 0.5); // 1/(1+betal) (+ 5x^3/128)
    } else {
        // Kepler, any near-earth eccentricity (el < ~0.33): five Newton trips on E - aynl cosE + axnl sinE = u
        // carrying eps = E - u and rotating (sinE, cosE) by each correction, with the rotation tier FIXED per
        // trip (quadratic convergence: |d| <= 1/2, 1/8, 2^-7, 2^-7, 2^-7) and validated instead of voted; the
        // generic loop's exit criterion el^2 d^4 < 4e-26 must hold for the last correction.
        double eps = 0.0, rden = 1.0, d = 0.0;
#pragma unroll
        for (int it = 0; it < 5; ++it) {
            rden = az_rcp1(fma(-s, aynl, fma(-c, axnl, 1.0)));
            d = fma(axnl, s, fma(-aynl, c, -eps)) * rden;
            eps += d;
            if (it == 0) {
                bad |= !(fabs(d) <= 0.5);
                az_rotate_large(s, c, d);
            } else if (it == 1) {
                bad |= !(fabs(d) <= AZ_ROT_MED);
                az_pq_med(d, p, q);
                az_rot_apply2(s, c, p, q);
            } else {
                bad |= !(fabs(d) <= AZ_ROT_SMALL);
                az_pq_small(d, rk, p, q);
                az_rot_apply2(s, c, p, q);
            }
        }
        const double d2 = d * d;
        bad |= !(el2 * d2 * d2 < 4.0e-26);
        ecose = fma(axnl, c, aynl * s);
        esine = fma(axnl, s, -(aynl * c));
        ome = 1.0 - ecose;
        // the last trip's reciprocal is 1/(1 - ecose) before the final rotation: off by el*d, one Newton step
        inv_ome = fma(rden, fma(-ome, rden, 1.0), rden);
        const double omel2 = 1.0 - el2;
        const double rb = az_rsqrt(omel2);
        betal = omel2 * rb;
        inv_omel2 = rb * rb;
        inv_1pb = az_rcp(1.0 + betal);
    }

    const double est = esine * inv_1pb;
    const double sinu = inv_ome * (s - fma(axnl, est, aynl));
    const double cosu = inv_ome * (c + fma(aynl, est, -axnl));
    const double sin2u = (sinu + sinu) * cosu;
    const double cos2u = fma(-2.0 * sinu, sinu, 1.0);

    const double inv_am = ra * ra;
    const double rl = am * ome;
    const double inv_pl = inv_am * inv_omel2;
    const double temp1 = g.half_j2 * inv_pl;
    const double temp2 = temp1 * inv_pl;
    bad |= !(temp2 <= AZ_FAST_TEMP2);

    const double mrt = fma(rl, fma(k.k_mrt() * temp2, betal, 1.0), k.k_c2u() * temp1 * cos2u);
    const double t2s = temp2 * sin2u;
    // J2 short-period corrections as tiny rotations (each bounded by 1.5 temp2 <= 9e-4); the secular
    // xnodcf t^2 part of the node rides on the node correction
    const double a_nd = fma(k.k_node(), t2s, k.xnodcf() * t2);
    bad |= !(fabs(a_nd) <= AZ_ROT_SMALL);
    double ssu = sinu, csu = cosu, sn = st.sO, cn = st.cO, si = k.sinio(), ci = k.cosio();
    az_rotate_tiny2(ssu, csu, k.k_su() * t2s, rk);
    az_pq_small(a_nd, rk, p, q); // the node correction carries the secular xnodcf t^2 term: one tier up
    az_rot_apply2(sn, cn, p, q);
    az_rotate_tiny2(si, ci, k.k_inc() * temp2 * cos2u, rk);

    const double xmx = -sn * ci, xmy = cn * ci;
    const double ux = fma(xmx, ssu, cn * csu);
    const double uy = fma(xmy, ssu, sn * csu);
    const double uz = si * ssu;
    const double rs = mrt * g.radius_km;
    r[0] = rs * ux;
    r[1] = rs * uy;
    r[2] = rs * uz;
    if (VEL) {
        const double rv = ra * g.vkmpersec;         // vkmpersec / sqrt(am)
        const double vk = rv * inv_ome;             // common factor of rdotl, rvdotl (km/s)
        const double nxt = rv * inv_am * temp1;     // (nm/xke) temp1, km/s
        const double mvt = fma(-nxt * k.x1mth2(), sin2u, vk * esine);
        const double rvdot = fma(nxt, fma(k.x1mth2(), cos2u, k.k_rv()), vk * betal);
        const double vx = fma(xmx, csu, -(cn * ssu));
        const double vy = fma(xmy, csu, -(sn * ssu));
        const double vz = si * csu;
        v[0] = fma(mvt, ux, rvdot * vx);
        v[1] = fma(mvt, uy, rvdot * vy);
        v[2] = fma(mvt, uz, rvdot * vz);
    }
    return bad;
}
