// propagate_device.h -- per-lane SGP4 / SDP4 step.  One lane owns one satellite and walks time.
//
// What it computes is what the reference computes per (satellite, time):
//   near-earth : Sgp4Batch.propagateBatchDirect (src/Sgp4Batch.zig L113-157) +
//                Sgp4.keplerAndPosVel (src/Sgp4.zig L646-750)
//   deep-space : Sdp4Batch.propagateBatchDirect (src/Sdp4Batch.zig L199-343) with the *scalar*
//                error semantics of Sdp4.propagateElementsCarry (src/Sdp4.zig L881-970)
// How it computes it is different (see devmath.h): angles that drift slowly in time are carried as
// (sin,cos) pairs in registers and advanced by small rotations; the Kepler iteration rotates
// (sin E, cos E) by each Newton correction instead of re-evaluating sincos; u = atan2(sinu,cosu)
// is never formed because only sin/cos of u + small corrections are needed.
#pragma once
#include "devmath.h"
#include "fields.h"

// ------------------------------------------------------------------------------------------
// near-earth constants of one satellite, split by how they are used inside the time loop:
//   Sgp4Lane  -- touched several times per step, or on the critical path: VGPR resident (15 doubles)
//   cold[]    -- touched exactly once per step by the secular update (18 doubles): staged in LDS,
//                one column per lane (cold[k*AZ_COLD_STRIDE]: conflict-free ds_read_b64), which
//                frees 36 VGPRs and lifts the occupancy of the VALU-bound kernel
//   mo/argpo/nodeo -- only needed when the carried (sin,cos) pairs are re-seeded: re-read from the
//                element table (L2-resident) on those steps
struct Sgp4Lane {
    double mdot, argpdot, nodedot, xnodcf;
    double aycof, xlcof, sinio, cosio;
    // J2 short-period factors, formed once per tile from con41 / x1mth2 / x7thm1 / sinio / cosio
    double k_mrt, k_c2u, k_su, k_node, k_inc, x1mth2, k_rv;
};
enum Sgp4Cold {
    C_cc1, C_bc4, C_t2cof, C_ecco, C_a_base, C_no_unkozai,
    // higher-order drag (zeroed at load for isimp satellites, which disables every term)
    C_omgcof, C_eta, C_xmcof, C_delmo, C_bc5, C_sinmao, C_d2, C_d3, C_d4, C_t3cof, C_t4cof, C_t5cof,
    C_NUM
};
#ifndef AZ_COLD_STRIDE
#define AZ_COLD_STRIDE 64
#endif
#define C_NUM_MAX 18

// (sin,cos) pairs carried from one time step to the next
struct Sgp4Carry {
    double t_prev;
    double sW, cW; // argpo + argpdot*t
    double sO, cO; // nodeo + nodedot*t + xnodcf*t^2
    double sA, cA; // mo + mdot*t
    // lane = time kernels only: a lane's successive steps are a constant dt apart on a uniform grid,
    // so the rotation of the mean anomaly (radians per step) and of the argument of perigee are the
    // SAME every step -- cached as (sin,cos)(mdot dt) and the (p,q) pair of argpdot dt
    double dt_c, sdA, cdA, pW, qW;
};

// products of the inclination-dependent constants that the short-period step uses
AZ_DEVICE void az_j2_factors(double con41, double x1mth2, double x7thm1, double sI, double cI, double &k_mrt,
                             double &k_c2u, double &k_su, double &k_node, double &k_inc, double &k_rv)
{
    k_mrt = -1.5 * con41;     // mrt   = rl (1 + k_mrt temp2 betal) + k_c2u temp1 cos2u
    k_c2u = 0.5 * x1mth2;
    k_su = -0.25 * x7thm1;    // dsu   = k_su temp2 sin2u
    k_node = 1.5 * cI;        // dnode = k_node temp2 sin2u
    k_inc = 1.5 * cI * sI;    // dinc  = k_inc temp2 cos2u
    k_rv = 1.5 * con41;       // rvdot = rvdotl + nx temp1 (x1mth2 cos2u + k_rv)
}

// where the once-per-step constants live: an LDS column (lane = satellite kernels) ...
#ifdef AZ_HOST_EMUL
#define AZ_MEMBER inline
#else
#define AZ_MEMBER __device__ __forceinline__
#endif
struct ColdLds {
    double *p;
    AZ_MEMBER double operator()(int k) const { return p[k * AZ_COLD_STRIDE]; }
    AZ_MEMBER void set(int k, double v) const { p[k * AZ_COLD_STRIDE] = v; }
    AZ_MEMBER double mc(int k) const { return az_mc_literal(k); } // polynomial coefficients: literals (devmath.h)
    AZ_MEMBER const ColdLds &fresh() const { return *this; }
};
// ... or plain (wave-uniform) values when one wave works on a single satellite (lane = time kernels)
struct ColdRegs {
    double c[C_NUM_MAX];
    AZ_MEMBER double operator()(int k) const { return c[k]; }
    AZ_MEMBER void set(int k, double v) { c[k] = v; }
    AZ_MEMBER double mc(int k) const { return az_mc_literal(k); }
    AZ_MEMBER const ColdRegs &fresh() const { return *this; }
};

template <class Cold>
AZ_DEVICE void az_load_sgp4(const double *__restrict__ el, size_t n_pad, size_t i, unsigned flags,
                            Sgp4Lane &e, Cold &cold)
{
#define L(f) el[(size_t)F_##f * n_pad + i]
#define CS(k, val) cold.set((k), (val))
    e.mdot = L(mdot); e.argpdot = L(argpdot); e.nodedot = L(nodedot); e.xnodcf = L(xnodcf);
    e.aycof = L(aycof); e.xlcof = L(xlcof); e.sinio = L(sinio); e.cosio = L(cosio);
    az_j2_factors(L(con41), L(x1mth2), L(x7thm1), e.sinio, e.cosio, e.k_mrt, e.k_c2u, e.k_su, e.k_node, e.k_inc,
                  e.k_rv);
    e.x1mth2 = L(x1mth2);
    CS(C_cc1, L(cc1)); CS(C_bc4, L(bc4)); CS(C_t2cof, L(t2cof)); CS(C_ecco, L(ecco));
    CS(C_a_base, L(sqrt_a_base)); CS(C_no_unkozai, L(no_unkozai)); // C_a_base slot: sqrt((xke/no)^(2/3))
    const bool ho = !(flags & AZ_FLAG_ISIMP);
    CS(C_omgcof, ho ? L(omgcof) : 0.0); CS(C_eta, L(eta)); CS(C_xmcof, ho ? L(xmcof) : 0.0);
    CS(C_delmo, L(delmo)); CS(C_bc5, ho ? L(bc5) : 0.0); CS(C_sinmao, L(sinmao));
    CS(C_d2, L(d2)); CS(C_d3, L(d3)); CS(C_d4, L(d4));
    CS(C_t3cof, L(t3cof)); CS(C_t4cof, L(t4cof)); CS(C_t5cof, L(t5cof));
#undef CS
#undef L
}

// ------------------------------------------------------------------------------------------
// Kepler solve + short-period corrections + orientation  (Sgp4.zig L675-749)
//   am, em          secular semi-major axis / eccentricity
//   (su0,cu0)       sin/cos of u0 = mm + argpm + temp*xlcof*axnl   (the Newton start)
//   axnl, aynl      equinoctial components
//   (sO,cO)         sin/cos of nodem;  (sI,cI) sin/cos of the (mean) inclination
//   ra = 1/sqrt(am)
// returns mrt (corrected radius, Earth radii)
struct J2Factors {
    double k_mrt, k_c2u, k_su, k_node, k_inc, x1mth2, k_rv;
};

template <bool VEL, class M = McLit>
AZ_DEVICE double az_kepler_posvel(const AzGrav &g, double am, double ra, double axnl, double aynl,
                                  double su0, double cu0, double sO, double cO, double sI, double cI,
                                  const J2Factors &k, const RotK &rk, double r[3], double v[3], const M &m = M())
{
    // Newton on  E - aynl*cosE + axnl*sinE = u  with eps = E - u carried instead of E.
    // Exit test: the step after d would be ~ (el/2) d^2, so stop once el2 * d^4 < (2e-13)^2.
    const double el2 = fma(axnl, axnl, aynl * aynl);
    double s = su0, c = cu0, rden;
    double inv_ome, betal, inv_omel2, inv_1pb, ecose, esine, ome;
    // Nine catalog members in ten have el2 < 1.6e-5 (el < 4e-3); the vote is wave-uniform by
    // construction in the lane = time kernel, and in the lane = satellite kernel the host sorts each
    // workgroup by eccentricity class.
    if (!az_any(el2 > 1.6e-5)) {
        // near-circular: one Newton step from E0 = u (|d0| <= el/(1-el)), then one CHORD step with the
        // same reciprocal: error after step 1 is (el/2) d0^2 <= 3.2e-8, the stale slope is off by
        // el*d0 <= 1.6e-5 relative, so step 2 leaves <= 5e-13 rad (4e-9 km); the rotation by d1 is
        // first order (d1^2/2 < 2e-16).  37 instructions instead of two full trips (62).
        rden = az_rcp1(fma(-s, aynl, fma(-c, axnl, 1.0)));
        const double d0 = fma(axnl, s, -(aynl * c)) * rden;
        az_rotate_le_tiny_m(s, c, d0, rk, m); // |d0| < 2^-10 for e < 1e-3, else the generic vote picks `small`
        const double d1 = fma(axnl, s, fma(-aynl, c, -d0)) * rden;
        const double s1 = fma(c, d1, s);
        c = fma(-s, d1, c);
        s = s1;
        ecose = fma(axnl, c, aynl * s);
        esine = fma(axnl, s, -(aynl * c));
        ome = 1.0 - ecose;
        // 1/(1 - ecose) from the reciprocal of step 1 (off by el*(d0+d1) ~ 1e-5): two Newton steps
        inv_ome = fma(rden, fma(-ome, rden, 1.0), rden);
        inv_ome = fma(inv_ome, fma(-ome, inv_ome, 1.0), inv_ome);
        // betal = sqrt(1 - x), 1/(1 - x), 1/(1 + betal), x = el2: short series, truncation < 3e-16
        betal = fma(el2, fma(el2, -0.125, -0.5), 1.0);                 // 1 - x/2 - x^2/8        (- x^3/16)
        inv_omel2 = fma(el2, fma(el2, fma(el2, 1.0, 1.0), 1.0), 1.0);  // 1 + x + x^2 + x^3      (+ x^4)
        inv_1pb = fma(el2, fma(el2, 0.0625, 0.125), 0.5);              // 1/2 + x/8 + x^2/16     (+ 5x^3/128)
    } else {
        // Exit test: the step after d would be ~ (el/2) d^2, so stop once el2 * d^4 < (2e-13)^2.
        double eps = 0.0;
        bool converged = false; // wave-uniform
        rden = 1.0;
#pragma unroll 1
        for (int it = 0; it < 10; ++it) {
            const double den = fma(-s, aynl, fma(-c, axnl, 1.0));
            const double num = fma(axnl, s, fma(-aynl, c, -eps));
            rden = az_rcp1(den);
            double d = num * rden;
            d = fmin(fmax(d, -0.95), 0.95);
            eps += d;
            if (it == 0)
                az_rotate_m(s, c, d, rk, m);
            else
                az_rotate_le_tiny_m(s, c, d, rk, m);
            const double d2 = d * d;
            if (!az_any(el2 * d2 * d2 >= 4.0e-26)) {
                converged = true;
                break;
            }
        }
        ecose = fma(axnl, c, aynl * s);
        esine = fma(axnl, s, -(aynl * c));
        ome = 1.0 - ecose;
        // `den` of the last trip was 1 - ecose BEFORE the final rotation by d: its reciprocal is off
        // by el*d relative (plus rcp1's 2^-46); one Newton step squares that:
        // (el d)^2 <= el sqrt(el2 d^4) < 2e-13 el at loop exit -- below 1e-8 km even at GEO
        inv_ome = fma(rden, fma(-ome, rden, 1.0), rden);
        if (!converged) inv_ome = az_rcp(ome); // 10 trips without convergence: no such bound
        const double omel2 = 1.0 - el2;
        const double rb = az_rsqrt(omel2);
        betal = omel2 * rb;
        inv_omel2 = rb * rb;
        inv_1pb = az_rcp(1.0 + betal);
    }
    const double inv_am = ra * ra;
    const double rl = am * ome; // inv_ome = am / rl
    const double est = esine * inv_1pb;
    const double sinu = inv_ome * (s - aynl - axnl * est);
    const double cosu = inv_ome * (c - axnl + aynl * est);
    const double sin2u = 2.0 * sinu * cosu;
    const double cos2u = fma(-2.0 * sinu, sinu, 1.0);

    const double inv_pl = inv_am * inv_omel2;
    const double temp1 = g.half_j2 * inv_pl;
    const double temp2 = temp1 * inv_pl;

    const double mrt = fma(rl, fma(k.k_mrt * temp2, betal, 1.0), k.k_c2u * temp1 * cos2u);
    const double t2s = temp2 * sin2u;
    double ssu = sinu, csu = cosu, sn = sO, cn = cO, si = sI, ci = cI;
    // the three corrections are bounded by 1.5*temp2 = 0.75 J2 / pl^2 (8e-4 for pl = 1): one vote
    if (!az_any(temp2 > 6.0e-4)) {
        az_rotate_tiny(ssu, csu, k.k_su * t2s, rk);
        az_rotate_tiny(sn, cn, k.k_node * t2s, rk);
        az_rotate_tiny(si, ci, k.k_inc * temp2 * cos2u, rk);
    } else {
        az_rotate_m(ssu, csu, k.k_su * t2s, rk, m);
        az_rotate_m(sn, cn, k.k_node * t2s, rk, m);
        az_rotate_m(si, ci, k.k_inc * temp2 * cos2u, rk, m);
    }

    const double xmx = -sn * ci, xmy = cn * ci;
    const double ux = fma(xmx, ssu, cn * csu);
    const double uy = fma(xmy, ssu, sn * csu);
    const double uz = si * ssu;
    const double rs = mrt * g.radius_km;
    r[0] = rs * ux;
    r[1] = rs * uy;
    r[2] = rs * uz;
    if (VEL) {
        const double sqrt_am = am * ra;
        const double inv_rl = inv_am * inv_ome;
        const double vk = sqrt_am * inv_rl * g.vkmpersec; // common factor of rdotl, rvdotl (km/s)
        const double nxt = inv_am * ra * temp1 * g.vkmpersec; // (nm/xke) temp1, km/s
        const double mvt = fma(-nxt * k.x1mth2, sin2u, vk * esine);
        const double rvdot = fma(nxt, fma(k.x1mth2, cos2u, k.k_rv), vk * betal);
        const double vx = fma(xmx, csu, -(cn * ssu));
        const double vy = fma(xmy, csu, -(sn * ssu));
        const double vz = si * csu;
        v[0] = fma(mvt, ux, rvdot * vx);
        v[1] = fma(mvt, uy, rvdot * vy);
        v[2] = fma(mvt, uz, rvdot * vz);
    }
    return mrt;
}

// ------------------------------------------------------------------------------------------
// one near-earth propagation.  `first` (wave-uniform) seeds the carried pairs with full sincos.
// `el`/`n_pad`/`sat` locate the satellite's column of the element table for the re-seed loads.
// STRIDE64 = false: consecutive steps of one lane are consecutive grid times (lane = satellite);
// STRIDE64 = true : a lane's consecutive steps are 64 grid times apart (lane = time, k_rows): the
//                   mean anomaly moves by radians between them and is simply re-evaluated, the J2
//                   angles still move by < 2^-7 rad and keep their carried pairs.
template <bool VEL, class Cold, bool STRIDE64 = false>
// cache_inc (STRIDE64 only, wave-uniform): the grid is uniform, so a lane's increments repeat and are worth caching; false on
// irregular grids, where every step would rebuild the cache for nothing (a full sincos of mdot dt: a seventh of the step).
AZ_DEVICE void az_sgp4_step(const Sgp4Lane &e, const Cold &cold, const double *__restrict__ el, size_t n_pad,
                            size_t sat, const AzGrav &g, const RotK &rk, double t, bool first, Sgp4Carry &st,
                            double r[3], double v[3], bool cache_inc = true)
{
#define CL(k) cold(k)
    const double t2 = t * t;
    // slowly drifting angles: advance the carried (sin,cos) pairs
    {
        const double dt = t - st.t_prev;
        const double dW = e.argpdot * dt;
        const double dO = dt * fma(e.xnodcf, t + st.t_prev, e.nodedot);
        const double dA = e.mdot * dt;
        if (first) {
            const double argpo = el[(size_t)F_argpo * n_pad + sat], nodeo = el[(size_t)F_nodeo * n_pad + sat];
            const double mo = el[(size_t)F_mo * n_pad + sat];
            az_sincos(fma(e.argpdot, t, argpo), st.sW, st.cW);
            az_sincos(fma(e.xnodcf, t2, fma(e.nodedot, t, nodeo)), st.sO, st.cO);
            az_sincos(fma(e.mdot, t, mo), st.sA, st.cA);
        } else if (STRIDE64) {
            az_rotate_le_small(st.sO, st.cO, dO, rk);
            if (!cache_inc) {
                // irregular grid: nothing repeats -- rotate W by its own increment, re-evaluate M (radians between a lane's steps)
                az_rotate(st.sW, st.cW, dW, rk);
                az_sincos(fma(e.mdot, t, el[(size_t)F_mo * n_pad + sat]), st.sA, st.cA);
            } else if (az_any(dt != st.dt_c || fabs(dW) > AZ_ROT_SMALL)) {
                // (re)build the cached increments; also the path of any non-uniform grid
                st.dt_c = (fabs(dW) > AZ_ROT_SMALL) ? -1.0e300 : dt;
                az_sincos(dA, st.sdA, st.cdA);
                az_pq_small(dW, rk, st.pW, st.qW);
                az_rotate(st.sW, st.cW, dW, rk);
                az_sincos(fma(e.mdot, t, el[(size_t)F_mo * n_pad + sat]), st.sA, st.cA);
            } else {
                az_rot_apply(st.sW, st.cW, st.pW, st.qW);
                const double ns = fma(st.sA, st.cdA, st.cA * st.sdA);
                st.cA = fma(st.cA, st.cdA, -(st.sA * st.sdA));
                st.sA = ns;
            }
        } else if (!az_any(fmax(fabs(dW), fabs(dO)) > AZ_ROT_MILLI || fabs(dA) > AZ_ROT_MED)) {
            // a one-minute grid lands here: J2 rates are ~1e-4 rad/min, the mean motion < 0.08
            az_rotate_tiny(st.sW, st.cW, dW, rk);
            az_rotate_tiny(st.sO, st.cO, dO, rk);
            az_rotate_med(st.sA, st.cA, dA);
        } else {
            // any other grid: per-angle tier votes (increments formed from dt: no cancellation)
            az_rotate(st.sW, st.cW, dW, rk);
            az_rotate(st.sO, st.cO, dO, rk);
            az_rotate(st.sA, st.cA, dA, rk);
        }
        st.t_prev = t;
    }

    // secular gravity + drag (Sgp4Batch.zig L121-154); isimp lanes carry zeros in the ho terms
    const double sA = st.sA, cA = st.cA; // (sin,cos) of xmdf = mo + mdot*t
    const double dm = fma(CL(C_eta), cA, 1.0);
    const double th = fma(CL(C_omgcof), t, CL(C_xmcof) * (dm * dm * dm - CL(C_delmo))); // delomg + delm
    const double t3 = t2 * t, t4 = t3 * t;
    const double tempa = 1.0 - CL(C_cc1) * t - CL(C_d2) * t2 - CL(C_d3) * t3 - CL(C_d4) * t4;
    // sin(mm), mm = xmdf + th, and (sin,cos)(argpm), argpm = argpdf - th: one (p,q) pair serves both
    double smm, sw = st.sW, cw = st.cW;
    if (!az_any(fabs(th) > AZ_ROT_SMALL)) {
        double p, q;
        az_pq_small(th, rk, p, q);
        smm = sA + fma(sA, q, cA * p);
        az_rot_apply(sw, cw, -p, q);
    } else {
        double sm = sA, cm = cA;
        az_rotate(sm, cm, th, rk);
        smm = sm;
        az_rotate(sw, cw, -th, rk);
    }
    const double tempe = fma(CL(C_bc5), smm - CL(C_sinmao), CL(C_bc4) * t);
    const double templ = fma(CL(C_t2cof), t2, fma(CL(C_t3cof), t3, t4 * fma(t, CL(C_t5cof), CL(C_t4cof))));

    // am = a_base tempa^2 (Sgp4Batch.zig L146), so sqrt(am) = sqrt(a_base) |tempa| exactly: ONE
    // reciprocal R = 1/(sqrt(am) (1 - em^2)) yields 1/sqrt(am) = R (1 - em^2) and
    // temp = 1/(am (1 - em^2)) = R / sqrt(am) -- instead of a reciprocal square root plus a reciprocal
    const double sqrt_am = CL(C_a_base) * fabs(tempa);
    const double am = sqrt_am * sqrt_am;
    const double em = fmax(CL(C_ecco) - tempe, 1.0e-6);
    const double omem2 = fma(-em, em, 1.0);
    const double R = az_rcp(sqrt_am * omem2);
    const double ra = R * omem2;
    const double temp = ra * R;

    const double axnl = em * cw;
    const double aynl = fma(em, sw, temp * e.aycof);
    // u0 = mm + argpm + temp*xlcof*axnl = xmdf + argpdf + no*templ + temp*xlcof*axnl
    double su0, cu0;
    az_angle_add(sA, cA, st.sW, st.cW, su0, cu0);
    az_rotate_le_small(su0, cu0, fma(CL(C_no_unkozai), templ, temp * e.xlcof * axnl), rk);

    const J2Factors k = {e.k_mrt, e.k_c2u, e.k_su, e.k_node, e.k_inc, e.x1mth2, e.k_rv};
    az_kepler_posvel<VEL>(g, am, ra, axnl, aynl, su0, cu0, st.sO, st.cO, e.sinio, e.cosio, k, rk, r, v);
#undef CL
}

// ------------------------------------------------------------------------------------------
// deep space
// Deep-space constants of one satellite, split like the near-earth ones:
//   Sdp4Lane -- secular rates, epoch angles, resonance start values: VGPR resident (24 doubles),
//               indexed by Sdp4Hot so that the same step code also runs from LDS-broadcast constants
//   cold[]   -- the 24 lunar/solar periodic coefficients (one use per step in dpper) and the 13
//               resonance coefficients (one use per integrator evaluation): LDS, one column per lane
enum Sdp4Hot {
    H_mo, H_mdot, H_argpo, H_argpdot, H_nodeo, H_nodedot, H_xnodcf, H_cc1, H_bc4, H_t2cof, H_ecco, H_inclo, H_no_unkozai, H_a_base, H_zmol, H_zmos, H_dedt, H_didt, H_dmdt, H_domdt, H_dnodt, H_xlamo, H_xfact, H_gsto,
    H_NUM
};
// lane = satellite: the 24 hot values live in VGPRs ...
struct Sdp4Lane {
    double v[H_NUM];
    int irez;
    AZ_MEMBER double operator()(int k) const { return v[k]; }
    AZ_MEMBER void set(int k, double x) { v[k] = x; }
};
// ... lane = time (one wave per satellite): every constant is wave-uniform and sits in LDS, one word
// each, read by all lanes at once (broadcast ds_read: no VALU slot, no SGPR/VGPR residency)
struct Sdp4Bcast {
    double *p;
    int irez;
    AZ_MEMBER double operator()(int k) const { return p[k]; }
    AZ_MEMBER void set(int k, double x) const { p[k] = x; }
};
enum Sdp4Cold {
    D_se2, D_se3, D_si2, D_si3, D_sl2, D_sl3, D_sl4, D_sgh2, D_sgh3, D_sgh4, D_sh2, D_sh3,
    D_ee2, D_e3, D_xi2, D_xi3, D_xl2, D_xl3, D_xl4, D_xgh2, D_xgh3, D_xgh4, D_xh2, D_xh3,
    D_d2201, D_d2211, D_d3210, D_d3222, D_d4410, D_d4422, D_d5220, D_d5232, D_d5421, D_d5433,
    D_del1, D_del2, D_del3,
    D_NUM
};
struct Sdp4Carry {
    double atime, xli, xni;
};

template <class Lane, class Cold>
AZ_DEVICE void az_load_sdp4(const double *__restrict__ el, size_t n_pad, size_t i, unsigned flags, Lane &e,
                            const Cold &cold)
{
#define L(f) el[(size_t)F_##f * n_pad + i]
#define HS(f) e.set(H_##f, L(f))
#define CS(f) cold.set(D_##f, L(f))
    HS(mo); HS(mdot); HS(argpo); HS(argpdot); HS(nodeo); HS(nodedot); HS(xnodcf);
    HS(cc1); HS(bc4); HS(t2cof); HS(ecco); HS(inclo); HS(no_unkozai); HS(a_base);
    HS(zmol); HS(zmos); HS(dedt); HS(didt); HS(dmdt); HS(domdt); HS(dnodt);
    HS(xlamo); HS(xfact); HS(gsto);
    e.irez = (int)AZ_FLAG_IREZ(flags);
    CS(se2); CS(se3); CS(si2); CS(si3); CS(sl2); CS(sl3); CS(sl4); CS(sgh2); CS(sgh3); CS(sgh4); CS(sh2); CS(sh3);
    CS(ee2); CS(e3); CS(xi2); CS(xi3); CS(xl2); CS(xl3); CS(xl4); CS(xgh2); CS(xgh3); CS(xgh4); CS(xh2); CS(xh3);
    CS(d2201); CS(d2211); CS(d3210); CS(d3222); CS(d4410); CS(d4422); CS(d5220); CS(d5232); CS(d5421); CS(d5433);
    CS(del1); CS(del2); CS(del3);
#undef CS
#undef HS
#undef L
}

// model constants, src/Sdp4.zig L15-52
#define AZ_ZES 0.01675
#define AZ_ZEL 0.05490
#define AZ_ZNS 1.19459e-5
#define AZ_ZNL 1.5835218e-4
#define AZ_RPTIM 4.37526908801129966e-3
#define AZ_FASX2 0.13130908
#define AZ_FASX4 2.8843198
#define AZ_FASX6 0.37448087
#define AZ_G22 5.7686396
#define AZ_G32 0.95240898
#define AZ_G44 1.8014998
#define AZ_G52 1.0508330
#define AZ_G54 4.4108898
#define AZ_STEPP 720.0
#define AZ_STEP2 259200.0

// resonance accelerations (Sdp4.computeResonanceAccel, src/Sdp4.zig L824-866).  Unlike the
// reference batch kernel (Sdp4Batch.zig L347-425: both branches for every lane) each branch is
// only entered by waves that hold such a satellite; the host orders the deep-space index list by
// resonance class so that waves are uniform.
#define DL(f) cold(D_##f)
template <class Lane, class Cold>
AZ_DEVICE void az_resonance_accel(const Lane &e, const Cold &cold, double xli, double xni,
                                  double atime, double &xndt, double &xnddt, double &xldot)
{
    xldot = xni + e(H_xfact);
    double xndt_h = 0.0, xnddt_h = 0.0;
    // Every phase is k xomi + m xli - G (half-day) or m (xli - fasx) (synchronous) with small integers k, m and model
    // constants G: ONE sincos of xli (and one of xomi) and angle additions give all of them -- 2 sincos + ~90 FMA-class
    // instructions for the ten half-day phases instead of ten sincos, 1 + ~25 instead of three for the synchronous ones.
    // (sin,cos) of the constants are literals; sin(a - G) = sin a cos G - cos a sin G.
    double sl, cl;
    az_sincos_m(xli, sl, cl, cold.fresh());
    const double s2l = 2.0 * sl * cl, c2l = fma(-2.0 * sl, sl, 1.0);
    if (az_any(e.irez == 2)) {
        const double xomi = fma(e(H_argpdot), atime, e(H_argpo));
        double so, co;
        az_sincos_m(xomi, so, co, cold.fresh());
        const double s2o = 2.0 * so * co, c2o = fma(-2.0 * so, so, 1.0);
        double acc_s = 0.0, acc_c = 0.0, acc_c2 = 0.0;
        // term(d, (sa, ca) = (sin,cos) of the phase without its constant, (sg, cg) = (sin,cos) G): d sin(a - G), d cos(a - G)
#define AZ_RES_TERM(d, sa, ca, sg, cg, ACC_C)                        \
    {                                                               \
        const double sn = fma(sa, cg, -((ca) * (sg)));              \
        const double cs = fma(ca, cg, (sa) * (sg));                 \
        acc_s = fma(d, sn, acc_s);                                  \
        ACC_C = fma(d, cs, ACC_C);                                  \
    }
        double sa, ca;
        // sin/cos of the model constants (src/Sdp4.zig L15-52): G22 = 5.7686396, G32 = 0.95240898, G44 = 1.8014998,
        // G52 = 1.0508330, G54 = 4.4108898
        const double sG22 = -4.92139430489155261e-01, cG22 = 8.70516387529729374e-01;
        const double sG32 = 8.14814406163892446e-01, cG32 = 5.79721901870011491e-01;
        const double sG44 = 9.73505778018079915e-01, cG44 = -2.28662415288155479e-01;
        const double sG52 = 8.67837401281277288e-01, cG52 = 4.96848311798841979e-01;
        const double sG54 = -9.54892377615299992e-01, cG54 = -2.96952095753168943e-01;
        az_angle_add(s2o, c2o, sl, cl, sa, ca);                       // 2 xomi + xli
        AZ_RES_TERM(DL(d2201), sa, ca, sG22, cG22, acc_c)
        AZ_RES_TERM(DL(d2211), sl, cl, sG22, cG22, acc_c)            //          xli
        az_angle_add(so, co, sl, cl, sa, ca);                         //   xomi + xli
        AZ_RES_TERM(DL(d3210), sa, ca, sG32, cG32, acc_c)
        AZ_RES_TERM(DL(d5220), sa, ca, sG52, cG52, acc_c)
        az_angle_add(-so, co, sl, cl, sa, ca);                        //  -xomi + xli
        AZ_RES_TERM(DL(d3222), sa, ca, sG32, cG32, acc_c)
        AZ_RES_TERM(DL(d5232), sa, ca, sG52, cG52, acc_c)
        az_angle_add(s2o, c2o, s2l, c2l, sa, ca);                     // 2 xomi + 2 xli
        AZ_RES_TERM(DL(d4410), sa, ca, sG44, cG44, acc_c2)
        AZ_RES_TERM(DL(d4422), s2l, c2l, sG44, cG44, acc_c2)         //          2 xli
        az_angle_add(so, co, s2l, c2l, sa, ca);                       //   xomi + 2 xli
        AZ_RES_TERM(DL(d5421), sa, ca, sG54, cG54, acc_c2)
        az_angle_add(-so, co, s2l, c2l, sa, ca);                      //  -xomi + 2 xli
        AZ_RES_TERM(DL(d5433), sa, ca, sG54, cG54, acc_c2)
#undef AZ_RES_TERM
        xndt_h = acc_s;
        xnddt_h = fma(2.0, acc_c2, acc_c) * xldot;
    }
    double xndt_g = 0.0, xnddt_g = 0.0;
    if (az_any(e.irez == 1)) {
        // synchronous: phases xli - fasx2, 2 (xli - fasx4), 3 (xli - fasx6); fasx2 = 0.13130908, 2 fasx4 = 5.7686396,
        // 3 fasx6 = 1.12344261
        double s3l, c3l;
        az_angle_add(s2l, c2l, sl, cl, s3l, c3l);
        const double sF2 = 1.30932065016401006e-01, cF2 = 9.91391342684885934e-01;
        const double sF4 = -4.92139430489155261e-01, cF4 = 8.70516387529729374e-01;
        const double sF6 = 9.01594990166664223e-01, cF6 = 4.32581175857633338e-01;
        const double s1 = fma(sl, cF2, -(cl * sF2)), c1 = fma(cl, cF2, sl * sF2);
        const double s2 = fma(s2l, cF4, -(c2l * sF4)), c2 = fma(c2l, cF4, s2l * sF4);
        const double s3 = fma(s3l, cF6, -(c3l * sF6)), c3 = fma(c3l, cF6, s3l * sF6);
        const double del1 = DL(del1), del2 = DL(del2), del3 = DL(del3);
        xndt_g = del1 * s1 + del2 * s2 + del3 * s3;
        xnddt_g = (del1 * c1 + 2.0 * del2 * c2 + 3.0 * del3 * c3) * xldot;
    }
    xndt = (e.irez == 2) ? xndt_h : xndt_g;
    xnddt = (e.irez == 2) ? xnddt_h : xnddt_g;
}

// Advance the resonance state (atime, xli, xni) to the last 720-minute boundary before t, with the
// reference's restart rule (src/Sdp4.zig L786-801, src/Sdp4Batch.zig L241-267).  The state reached
// after k steps of +-720 min is a pure function of k, so carrying it, restarting from epoch, or
// loading it from the per-tile seed table (k_deep_seed) are all equivalent.
template <class Lane, class Cold>
AZ_DEVICE void az_resonance_advance(const Lane &e, const Cold &cold, double t, Sdp4Carry &cy)
{
    const bool res = e.irez != 0;
    if (res && (cy.atime == 0.0 || t * cy.atime <= 0.0 || fabs(t) < fabs(cy.atime))) {
        cy.atime = 0.0;
        cy.xni = e(H_no_unkozai);
        cy.xli = e(H_xlamo);
    }
    const double delt = (t > 0.0) ? AZ_STEPP : -AZ_STEPP;
    while (az_any(res && fabs(t - cy.atime) >= AZ_STEPP)) {
        double xndt, xnddt, xldot;
        az_resonance_accel(e, cold, cy.xli, cy.xni, cy.atime, xndt, xnddt, xldot);
        if (res && fabs(t - cy.atime) >= AZ_STEPP) {
            cy.xli += xldot * delt + xndt * AZ_STEP2;
            cy.xni += xndt * delt + xnddt * AZ_STEP2;
            cy.atime += delt;
        }
    }
}

// The accelerations depend on the integrator state only, and the state changes every 720 minutes: a lane that
// walks a time grid keeps them next to the atime they belong to (the state after k integrator steps is a pure function of
// k, so an equal atime means an equal state) and re-evaluates only when its state moved -- two iterations in eleven on a
// one-minute grid instead of every step (3 sincos per evaluation for synchronous members, 10 for half-day ones).
struct Sdp4Acc {
    double atime, xndt, xnddt, xldot;
};
// cy: the chunk's seed on entry, this lane's integrator state for time t on return (reference restart rule and stepping,
// src/Sdp4.zig L786-801); acc: the accelerations at that state.  One evaluation site serves the steps and the final state.
template <class Lane, class Cold>
AZ_DEVICE void az_resonance_cached(const Lane &e, const Cold &cold, double t, Sdp4Carry &cy, Sdp4Acc &acc)
{
    if (cy.atime == 0.0 || t * cy.atime <= 0.0 || fabs(t) < fabs(cy.atime)) {
        cy.atime = 0.0;
        cy.xni = e(H_no_unkozai);
        cy.xli = e(H_xlamo);
    }
    bool have = (cy.atime == acc.atime);
    double xndt = acc.xndt, xnddt = acc.xnddt, xldot = acc.xldot;
    const double delt = (t > 0.0) ? AZ_STEPP : -AZ_STEPP;
    for (;;) {
        const bool adv = fabs(t - cy.atime) >= AZ_STEPP;
        if (!az_any(adv | !have)) break;
        if (az_any(!have)) {
            double a, b, c;
            az_resonance_accel(e, cold, cy.xli, cy.xni, cy.atime, a, b, c);
            if (!have) {
                xndt = a;
                xnddt = b;
                xldot = c;
            }
            have = true;
        }
        if (adv) {
            cy.xli += xldot * delt + xndt * AZ_STEP2;
            cy.xni += xndt * delt + xnddt * AZ_STEP2;
            cy.atime += delt;
            have = false;
        }
    }
    acc.atime = cy.atime;
    acc.xndt = xndt;
    acc.xnddt = xnddt;
    acc.xldot = xldot;
}

// one deep-space propagation; returns 0 / 1 (eccentricity) / 6 (decayed) per the scalar
// reference path (src/Sdp4.zig L914-921, L937-938, L967).  PRE: cy is already this lane's integrator state for t and
// `pre` the accelerations at it (az_resonance_cached); a template flag and a reference, not a nullable pointer: taking
// the address of the caller's struct keeps it in scratch memory.
template <bool VEL, bool PRE, class Lane, class Cold>
AZ_DEVICE int az_sdp4_step_impl(const Lane &e, const Cold &cold, const AzGrav &g, const RotK &rk, double t,
                                Sdp4Carry &cy, double r[3], double v[3], const Sdp4Acc &pre)
{
    const double t2 = t * t;
    const double tempa = 1.0 - e(H_cc1) * t;
    const double tempe = e(H_bc4) * t;
    const double templ = e(H_t2cof) * t2;

    double em = fma(e(H_dedt), t, e(H_ecco));
    double inclm = fma(e(H_didt), t, e(H_inclo));
    double argpm = fma(e(H_argpdot), t, e(H_argpo)) + e(H_domdt) * t;
    double nodem = fma(e(H_xnodcf), t2, fma(e(H_nodedot), t, e(H_nodeo))) + e(H_dnodt) * t;
    double mm = fma(e(H_mdot), t, e(H_mo)) + e(H_dmdt) * t;
    double nm = e(H_no_unkozai);
    int rc = 0;

    // resonance integrator (dspace, src/Sdp4.zig L784-819)
    double a23 = e(H_a_base); // (xke/nm)^(2/3)
    if (az_any(e.irez != 0)) {
        const bool res = e.irez != 0;
        double xndt, xnddt, xldot;
        if (PRE) {
            xndt = pre.xndt;
            xnddt = pre.xnddt;
            xldot = pre.xldot;
        } else {
            az_resonance_advance(e, cold, t, cy);
            az_resonance_accel(e, cold, cy.xli, cy.xni, cy.atime, xndt, xnddt, xldot);
        }
        if (res) {
            const double ft = t - cy.atime;
            nm = cy.xni + xndt * ft + xnddt * ft * ft * 0.5;
            const double xl = cy.xli + xldot * ft + xndt * ft * ft * 0.5;
            const double theta = fma(t, AZ_RPTIM, e(H_gsto)); // only ever used through sin/cos: no modulo
            mm = (e.irez != 2) ? xl - nodem - argpm + theta : xl - 2.0 * nodem + 2.0 * theta;
            nm = e(H_no_unkozai) + (nm - e(H_no_unkozai));
        }
        if (nm <= 0.0) { rc = 6; nm = e(H_no_unkozai); }
        // (xke/nm)^(2/3) = a_base (1+x)^(-2/3), x = nm/no - 1 (|x| ~ 1e-5 for real resonant orbits):
        // binomial series to x^5 (next term 0.4 x^6 < 3e-17 for |x| <= 2e-3), cbrt only beyond that
        const double x = (nm - e(H_no_unkozai)) * az_rcp(e(H_no_unkozai));
        if (!az_any(fabs(x) > 2.0e-3)) {
            double f = fma(x, cold.mc(MC_A23_0), cold.mc(MC_A23_1));
            f = fma(x, f, cold.mc(MC_A23_2));
            f = fma(x, f, cold.mc(MC_A23_3));
            f = fma(x, f, cold.mc(MC_A23_4));
            a23 = e(H_a_base) * fma(x, f, 1.0);
        } else {
            const double q = g.xke / nm;
            a23 = cbrt(q * q);
        }
    }

    const double am = a23 * tempa * tempa;
    em -= tempe;
    if (rc == 0 && (em >= 1.0 || em < -0.001)) rc = 1;
    if (em < 1.0e-6) em = 1.0e-6;
    if (rc == 0 && am < 0.95) rc = 6;
    mm += e(H_no_unkozai) * templ;
    // (no mod 2pi: the angles are only consumed by sin/cos, except inside the Lyddane branch)

    // lunar-solar periodics (dpper, src/Sdp4.zig L681-759)
    double sI, cI; // sin/cos of the perturbed inclination
    {
        double zm = fma(AZ_ZNS, t, e(H_zmos));
        double szm, czm, sinzf, coszf;
        az_sincos_m(zm, szm, czm, cold.fresh());
        sinzf = szm; coszf = czm;
        az_rotate_med_m(sinzf, coszf, 2.0 * AZ_ZES * szm, cold.fresh()); // zf = zm + 2 zes sin zm  (|.| <= 0.0335)
        double f2 = fma(0.5 * sinzf, sinzf, -0.25), f3 = -0.5 * sinzf * coszf;
        const double ses = DL(se2) * f2 + DL(se3) * f3;
        const double sis = DL(si2) * f2 + DL(si3) * f3;
        const double sls = DL(sl2) * f2 + DL(sl3) * f3 + DL(sl4) * sinzf;
        const double sghs = DL(sgh2) * f2 + DL(sgh3) * f3 + DL(sgh4) * sinzf;
        const double shs = DL(sh2) * f2 + DL(sh3) * f3;
        zm = fma(AZ_ZNL, t, e(H_zmol));
        az_sincos_m(zm, szm, czm, cold.fresh());
        sinzf = szm; coszf = czm;
        az_rotate_med_m(sinzf, coszf, 2.0 * AZ_ZEL * szm, cold.fresh()); // |.| <= 0.1098
        f2 = fma(0.5 * sinzf, sinzf, -0.25);
        f3 = -0.5 * sinzf * coszf;
        const double sel = DL(ee2) * f2 + DL(e3) * f3;
        const double sil = DL(xi2) * f2 + DL(xi3) * f3;
        const double sll = DL(xl2) * f2 + DL(xl3) * f3 + DL(xl4) * sinzf;
        const double sghl = DL(xgh2) * f2 + DL(xgh3) * f3 + DL(xgh4) * sinzf;
        const double shl = DL(xh2) * f2 + DL(xh3) * f3;
        const double pe = ses + sel, pinc = sis + sil, pl = sls + sll;
        double pgh = sghs + sghl, ph = shs + shl;

        inclm += pinc;
        em += pe;
        az_sincos_m(inclm, sI, cI, cold.fresh());
        const double sinip = sI, cosip = cI;
        const bool lyd = inclm < 0.2;
        if (az_any(lyd)) {
            // Lyddane modification for near-equatorial orbits (src/Sdp4.zig L731-757), in closed form.  The reference
            // forms alfdp = sinip sin(node) + ph cos(node) + pinc cosip sin(node), betdp likewise with cos, takes
            // node' = atan2(alfdp, betdp) on the branch nearest the old node xnoh = node mod 2 pi, and then
            // argp' = xls + dls - mm' - cosip node'.  Rotating (betdp, alfdp) back by xnoh,
            //      alfdp cos(xnoh) - betdp sin(xnoh) = ph,        betdp cos(xnoh) + alfdp sin(xnoh) = sinip + pinc cosip,
            // so  node' = xnoh + delta,  delta = atan2(ph, sinip + pinc cosip)  (principal value = the nearest branch), and
            //      argp' = argp + pgh - pinc xnoh sinip - cosip delta,          mm' = mm + pl
            // -- no sincos of the node, one small polynomial atan2 instead of libm's.  (xnoh's VALUE in [0, 2 pi) enters
            // the reference's dls term, hence the one explicit modulus.)
            if (lyd) {
                const double xnoh = az_mod2pi(nodem);
                const double delta = az_atan2_m(ph, fma(pinc, cosip, sinip), cold.fresh());
                argpm += pgh - pinc * xnoh * sinip - cosip * delta;
                nodem += delta;
                mm += pl;
            }
        }
        if (!lyd) {
            ph *= az_rcp(sinip);
            pgh -= cosip * ph;
            argpm += pgh;
            nodem += ph;
            mm += pl;
        }
    }

    if (inclm < 0.0) {
        inclm = -inclm;
        sI = -sI;
        nodem += AZ_PI;
        argpm -= AZ_PI;
    }
    if (em < 1.0e-6) em = 1.0e-6;
    if (rc == 0 && em >= 1.0) rc = 1;
    if (rc != 0) em = 0.5; // keep the arithmetic below finite; the row is zero-filled by the caller

    const double cI2 = cI * cI;
    const double aycof = -0.5 * g.j3oj2 * sI;
    const double den = (fabs(cI + 1.0) > 1.5e-12) ? 1.0 + cI : 1.5e-12;
    const double xlcof = -0.25 * g.j3oj2 * sI * fma(5.0, cI, 3.0) * az_rcp(den);
    J2Factors k;
    az_j2_factors(fma(3.0, cI2, -1.0), 1.0 - cI2, fma(7.0, cI2, -1.0), sI, cI, k.k_mrt, k.k_c2u, k.k_su, k.k_node,
                  k.k_inc, k.k_rv);
    k.x1mth2 = 1.0 - cI2;

    const double ra = az_rsqrt(am);
    const double temp = ra * ra * az_rcp(fma(-em, em, 1.0));
    double sw, cw, sO, cO, su0, cu0;
    az_sincos_m(argpm, sw, cw, cold.fresh());
    az_sincos_m(nodem, sO, cO, cold.fresh());
    const double axnl = em * cw;
    const double aynl = fma(em, sw, temp * aycof);
    az_sincos_m(mm + argpm + temp * xlcof * axnl, su0, cu0, cold.fresh());

    const double mrt = az_kepler_posvel<VEL>(g, am, ra, axnl, aynl, su0, cu0, sO, cO, sI, cI, k, rk, r, v, cold.fresh());
    if (rc == 0 && mrt < 1.0) rc = 6;
    return rc;
}
template <bool VEL, class Lane, class Cold>
AZ_DEVICE int az_sdp4_step(const Lane &e, const Cold &cold, const AzGrav &g, const RotK &rk, double t,
                           Sdp4Carry &cy, double r[3], double v[3])
{
    const Sdp4Acc none = {0.0, 0.0, 0.0, 0.0};
    return az_sdp4_step_impl<VEL, false>(e, cold, g, rk, t, cy, r, v, none);
}
template <bool VEL, class Lane, class Cold>
AZ_DEVICE int az_sdp4_step_pre(const Lane &e, const Cold &cold, const AzGrav &g, const RotK &rk, double t,
                               Sdp4Carry &cy, double r[3], double v[3], const Sdp4Acc &acc)
{
    return az_sdp4_step_impl<VEL, true>(e, cold, g, rk, t, cy, r, v, acc);
}
#undef DL

// ------------------------------------------------------------------------------------------
// output frames (Constellation.zig L54-56, L489-506; WorldCoordinateSystem.zig L98-121)
AZ_DEVICE void az_to_ecef(double p[3], double sg, double cg)
{
    const double x = fma(p[0], cg, p[1] * sg);
    const double y = fma(p[1], cg, -(p[0] * sg));
    p[0] = x;
    p[1] = y;
}

// ECEF -> (lat rad, lon rad, alt km), WGS84.  The reference iterates lat <- atan2(z + e2 N(lat) sin lat, rho) to a fixed point
// (src/WorldCoordinateSystem.zig L98-121: <= 10 trips, exit once the latitude moves by less than 1e-12; the map contracts by
// e2 = 0.0067 per trip).  The same latitude comes out of Bowring's form on the PARAMETRIC latitude beta, tan beta = (1 - f)
// tan lat:  tan lat = (z + e'2 b sin^3 beta) / (rho - e2 a cos^3 beta), which converges cubically -- from beta0 =
// atan2(z, (1 - f) rho) the formula alone is good to 8e-9 rad, after ONE refinement of beta to 2e-16 at every altitude from
// 150 to 80,000 km (checked against long-double arithmetic) -- and is carried on (sin,cos) pairs: (sin,cos) of atan2(Y, X) is
// (Y, X) / hypot(Y, X), so there is no trigonometric call before the ONE polynomial atan2 each for latitude and longitude.
// Three reciprocal square roots and ~25 multiply-adds in place of round 3's six fixed-point trips (twelve rsqrt): ~130
// instructions instead of ~230.  Altitude as the reference forms it: rho / cos(lat) - N.
AZ_DEVICE void az_ecef_to_geodetic(double p[3])
{
    const double f = 1.0 / 298.257223563;
    const double e2 = 2.0 * f - f * f;
    const double a = 6378.137;
    const double omf = 1.0 - f;
    const double ep2b = e2 / (1.0 - e2) * (a * omf), e2a = e2 * a; // e'^2 b, e^2 a
    const double x = p[0], y = p[1], z = p[2];
    const double rho2 = fma(x, x, y * y);
    const double rho = rho2 * az_rsqrt(fmax(rho2, 1.0e-300));
    double sb = z, cb = omf * rho;                                      // beta0
    double ih = az_rsqrt(fmax(fma(sb, sb, cb * cb), 1.0e-300));
    sb *= ih; cb *= ih;
    double num = fma(ep2b * sb * sb, sb, z), den = fma(-e2a * cb * cb, cb, rho);
    sb = omf * num; cb = den;                                           // beta1
    ih = az_rsqrt(fmax(fma(sb, sb, cb * cb), 1.0e-300));
    sb *= ih; cb *= ih;
    num = fma(ep2b * sb * sb, sb, z); den = fma(-e2a * cb * cb, cb, rho);
    ih = az_rsqrt(fmax(fma(num, num, den * den), 1.0e-300));
    const double s = num * ih, c = den * ih;                            // (sin,cos) of the latitude
    const double N = a * az_rsqrt(fma(-e2 * s, s, 1.0));
    p[0] = az_atan2(s, c);
    p[1] = az_atan2(y, x);
    p[2] = rho * az_rcp(fmax(c, 6.123233995736766e-17)) - N; // (cos(pi/2) in fp64, the reference's divisor on the axis)
}
