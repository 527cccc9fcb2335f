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
// Instead of choosing tiers with wave votes and branches the step runs straight through.  Its assumptions are bounded
// once per time WINDOW from the satellite's constants alone (az_fast_window_ok: rigorous bounds on every tiered angle,
// on el^2 and on the J2 factor over the whole window; evaluated per staged grid by k_plan_windows), so a window that
// passes needs no per-step check; the eccentric form additionally RETURNS a per-lane `bad` predicate for its Newton
// iteration, on which the caller votes and hands the remaining grid points to the generic az_sgp4_step loop (results
// are only stored after the vote, so nothing wrong ever reaches memory).  197 fp64 instructions per propagation against
// ~300 executed by the generic loop (whose tier votes cost compares, branches and the register moves at every merge).
#pragma once
#include <type_traits>

#include "propagate_device.h"

// per-satellite constants of the fast step (wave-uniform in the lane = time kernel, per lane in the
// lane = satellite kernel).  Folded products differ from the table rows by one rounding at most.
// Names of the constants; the step reads them through accessor methods so that a kernel can decide where
// each one lives (registers, or one LDS word read by all lanes at once).
#define AZ_FASTK_COLD(X)                                                                                     \
    X(cc1) X(d2) X(d3) X(d4) X(nl2) X(nl3) X(nl4) X(nl5) X(eta) X(omgcof) X(xmcof) X(xd) X(bc4) X(bc5) X(ecb) X(sab)
#define AZ_FASTK_HOT(X)                                                                                      \
    X(aycof) X(xlcof) X(xnodcf) X(sinio) X(cosio) X(k_mrt) X(k_c2u) X(k_su) X(k_node) X(k_inc) X(x1mth2) X(k_rv) \
    X(sdA) X(cdA) X(sdW) X(cdW) X(tc) X(sdU) X(cdU) X(tmid) X(nodedot) X(sOc) X(cOc)
enum FastCold {
#define X(n) FC_##n,
    AZ_FASTK_COLD(X)
#undef X
    FC_NUM
};
// QUASI-uniform grids (DELTA): times[i] = t0 + i step + delta_i with |delta_i| <= AZ_DELTA_MAX minutes -- what the reference's
// own API produces, times = ((jd + fr) - reference_jd) * 1440 (bindings/python/astroz/api.py L300-302, src/Constellation.zig
// L266-269): jd + fr at 2.46e6 days is quantised to 2^-31 day, so its grids are uniform only to ~4e-7 min.  The carried
// pairs still advance by constant increments along the IDEAL grid; the step is evaluated at the ACTUAL time t0 + i step +
// delta_i and corrects the phases to first order in delta (second order: (0.075 rad/min x 4e-6 min)^2 / 2 = 4.5e-14 rad):
// the rates of M, of W and of the carried phase U (mdot + argpdot + 2 kappa tc).  Three more once-per-step constants.
#define AZ_FASTK_DELTA(X) X(mdot) X(argpdot) X(udot)
enum FastColdX {
    FCX_first_ = FC_NUM - 1,
#define X(n) FCX_##n,
    AZ_FASTK_DELTA(X)
#undef X
    FCX_NUM
};
#define AZ_DELTA_MAX 4.0e-6 /* minutes: largest deviation from the ideal grid of the TIGHT form (DELTA = 1: fp32 deviations) */
// DELTA = 2, the WIDE form: grids that are uniform up to a jitter of seconds (time stamps of a periodic process): |delta_i| <=
// AZ_DELTA_WIDE_MAX.  The deviations are staged as fp64 (a 0.3-min value in fp32 is off by 2e-8 min = 1e-9 rad of mean anomaly);
// U is corrected exactly as before (it is linear in t apart from the drag terms, which see the actual t); the M pair by a
// second-order rotation in the near-circular form and by the 1/16-rad polynomial in the eccentric form; the W pair to first
// order (argpdot delta < 3e-5).
#define AZ_DELTA_WIDE_MAX 0.5
#define AZ_DELTA_SEG 768    /* grid points per time segment whose deviations a wave / tile stages in LDS (fp32) */
// everything in registers (lane = satellite kernels, host emulation)
struct FastK {
    static constexpr bool SCALAR = false;
#define X(n) double n##_;
    AZ_FASTK_COLD(X) AZ_FASTK_HOT(X) AZ_FASTK_DELTA(X)
#undef X
#define X(n) AZ_MEMBER double n() const { return n##_; }
    AZ_FASTK_COLD(X) AZ_FASTK_HOT(X) AZ_FASTK_DELTA(X)
#undef X
};
// lane = time kernels (one satellite per wave): the 18 constants used several times per step or sitting
// on the critical path stay in (scalar) registers, the 16 once-per-step ones are LDS words read by all
// lanes at once (broadcast ds_read: no VALU slot, no register residency)
struct FastKBcast {
    static constexpr bool SCALAR = true; // the hot constants are wave-uniform (SGPR pairs)
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
#define X(n) AZ_MEMBER double n() const { return cold[FCX_##n]; }
    AZ_FASTK_DELTA(X)
#undef X
};

// lane = satellite kernels: hot constants in registers, the once-per-step ones in a per-lane LDS column
// (cold[k * AZ_COLD_STRIDE], conflict-free ds_read_b64) -- 32 VGPRs less, which is what keeps k_propagate's fast
// loop free of scratch spills (a spilled register's reload waits on vmcnt, i.e. on the output stores in flight)
struct FastKCol {
    static constexpr bool SCALAR = false;
    const double *cold;
#define X(n) double n##_;
    AZ_FASTK_HOT(X)
#undef X
#define X(n) AZ_MEMBER double n() const { return n##_; }
    AZ_FASTK_HOT(X)
#undef X
#define X(n) AZ_MEMBER double n() const { return cold[FC_##n * AZ_COLD_STRIDE]; }
    AZ_FASTK_COLD(X)
#undef X
#define X(n) AZ_MEMBER double n() const { return 0.0; } /* (the lane = satellite kernel has no DELTA form) */
    AZ_FASTK_DELTA(X)
#undef X
};

// carried (sin,cos) pairs: M = mo + mdot t, W = argpo + argpdot t, and
// U = M + W + the part of the along-track drag phase kappa t^2 (kappa = no_unkozai t2cof) that is LINEAR about
// the centre tc of the wave's time window:  kappa t^2 = kappa tc^2 + 2 kappa tc (t - tc) + kappa (t - tc)^2.
// The first two terms ride on U (a linear phase: constant increment), only kappa (t - tc)^2 stays in the small
// rotation of the step -- so a high-drag member a week from epoch (kappa t^2 of a radian) is as cheap and as
// valid as a fresh one.  tc = 0 (the plain form U = M + W) whenever kappa t^2 is small over the whole window.
// The node needs no carried pair at all: it moves by a few hundredths of a radian across a window, so (sin,cos) of the node at
// the window centre tmid are per-window constants and nodedot (t - tmid), like the secular xnodcf t^2, rides on
// the J2 short-period rotation of the node (one FMA).
struct FastCarry {
    double sA, cA, sW, cW, sU, cU;
};

// row offsets of the increment table written by k_prep_inc: inc[(AZ_INC_* + AZ_INC_NUM*which) * n_pad + sat],
// which = 0: dt = 64 grid steps (lane = time kernels), 1: dt = one grid step (lane = satellite kernels)
enum { AZ_INC_sdA, AZ_INC_cdA, AZ_INC_sdW, AZ_INC_cdW, AZ_INC_NUM };

// everything of a FastK but the window constants; (sdA, cdA), (sdW, cdW): the increments of M and W over one lane step
AZ_DEVICE void az_load_fast_with(const double *__restrict__ el, size_t n_pad, size_t i, unsigned flags, double sdA, double cdA,
                                 double sdW, double cdW, FastK &k)
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
    k.sdA_ = sdA; k.cdA_ = cdA;
    k.sdW_ = sdW; k.cdW_ = cdW;
    k.nodedot_ = L(nodedot);
    k.mdot_ = L(mdot); k.argpdot_ = L(argpdot);
    k.udot_ = k.mdot_ + k.argpdot_; // (tc = 0; az_fast_udot once the window's tc is known)
#undef L
}
// ... with the increments k_prep_inc tabulated for the staged grid
AZ_DEVICE void az_load_fast(const double *__restrict__ el, size_t n_pad, size_t i, unsigned flags,
                            const double *__restrict__ inc, int which, FastK &k)
{
    const double *q = inc + (size_t)(AZ_INC_NUM * which) * n_pad + i;
    az_load_fast_with(el, n_pad, i, flags, q[(size_t)AZ_INC_sdA * n_pad], q[(size_t)AZ_INC_cdA * n_pad], q[(size_t)AZ_INC_sdW * n_pad],
                      q[(size_t)AZ_INC_cdW * n_pad], k);
}
// rate of the carried phase U about the window centre tc (DELTA: the first-order correction of U)
AZ_DEVICE void az_fast_udot(FastK &k) { k.udot_ = fma(2.0 * k.nl2_, k.tc_, k.mdot_ + k.argpdot_); }

// Window set-up: the centre tc about which the drag phase is expanded and the increment (sin,cos) of U for
// steps of dt.  [t_a, t_b]: the tsince range of the window (any order).  Wave-uniform in the lane = time kernels.
template <class K>
AZ_DEVICE void az_fast_window(const double *__restrict__ el, size_t n_pad, size_t i, double t_a, double t_b, double dt, K &k)
{
#define L(f) el[(size_t)F_##f * n_pad + i]
    const double kappa = L(no_unkozai) * L(t2cof);
    const double tmax = fmax(fabs(t_a), fabs(t_b));
    k.tmid_ = 0.5 * (t_a + t_b);
    az_sincos(fma(k.nodedot_, k.tmid_, L(nodeo)), k.sOc_, k.cOc_);
    if (!az_any(fabs(kappa) * tmax * tmax > 0.05)) {
        // U = M + W: its increment is the sum of the two increments
        k.tc_ = 0.0;
        k.sdU_ = fma(k.sdA_, k.cdW_, k.cdA_ * k.sdW_);
        k.cdU_ = fma(k.cdA_, k.cdW_, -(k.sdA_ * k.sdW_));
    } else {
        k.tc_ = 0.5 * (t_a + t_b);
        az_sincos((L(mdot) + L(argpdot) + 2.0 * kappa * k.tc_) * dt, k.sdU_, k.cdU_);
    }
    if constexpr (std::is_same<K, FastK>::value) az_fast_udot(k);
#undef L
}

// ---- the per-satellite record of the lane = time kernels (one satellite per wave).  az_load_fast's folded products are the
// same for every wave that ever works on the satellite, so k_prep_rec evaluates them ONCE per staged grid into an
// array-of-structures record; a wave then reads its 33 constants with a handful of wide scalar loads (hot fields: straight
// into SGPRs, no arithmetic on uniform values in the vector ALUs, no v_readfirstlane) and one per-lane vector load (lane j
// fetches cold field j for the LDS table: no SGPR -> VGPR moves).  Layout: [FCX_NUM cold fields in FastCold / FastColdX order |
// AZ_FASTK_HOT_REC fields | the four angles of the seeds], padded to a 64-byte multiple.
#define AZ_FASTK_HOT_REC(X)                                                                                  \
    X(aycof) X(xlcof) X(xnodcf) X(sinio) X(cosio) X(k_mrt) X(k_c2u) X(k_su) X(k_node) X(k_inc) X(x1mth2) X(k_rv) \
    X(sdA) X(cdA) X(sdW) X(cdW) X(nodedot)
enum FastRec {
    FR_COLD0 = 0,
    FR_HOT0 = FCX_NUM,
#define X(n) FR_##n,
    FR_first_hot_ = FR_HOT0 - 1,
    AZ_FASTK_HOT_REC(X)
#undef X
    FR_mo, FR_mdot, FR_argpo, FR_argpdot,
    FR_USED,
    FR_NUM = (FR_USED + 7) & ~7
};
AZ_DEVICE void az_fast_rec_store(const double *__restrict__ el, size_t n_pad, size_t i, const FastK &k, double *__restrict__ rec)
{
#define X(n) rec[FC_##n] = k.n##_;
    AZ_FASTK_COLD(X)
#undef X
    rec[FCX_mdot] = k.mdot_; rec[FCX_argpdot] = k.argpdot_; rec[FCX_udot] = k.mdot_ + k.argpdot_; // (udot: tc = 0; the wave adds 2 kappa tc)
#define X(n) rec[FR_##n] = k.n##_;
    AZ_FASTK_HOT_REC(X)
#undef X
#define L(f) el[(size_t)F_##f * n_pad + i]
    rec[FR_mo] = L(mo); rec[FR_mdot] = L(mdot); rec[FR_argpo] = L(argpo); rec[FR_argpdot] = L(argpdot);
#undef L
    for (int j = FR_USED; j < FR_NUM; ++j) rec[j] = 0.0;
}
// the wave-uniform part of a FastKBcast from the record (scalar loads)
template <class K, class P = const double *__restrict__>
AZ_DEVICE void az_fast_rec_hot(P rec, K &k)
{
#define X(n) k.n##_ = rec[FR_##n];
    AZ_FASTK_HOT_REC(X)
#undef X
}
// az_seed_fast from the record, the sincos coefficients through an accessor (an LDS table in the kernels); kappa = nl2
template <class M>
AZ_DEVICE void az_seed_fast_rec(const double *__restrict__ rec, double kappa, double t, double tc, FastCarry &st, const M &m)
{
    az_sincos_m(fma(rec[FR_mdot], t, rec[FR_mo]), st.sA, st.cA, m);
    az_sincos_m(fma(rec[FR_argpdot], t, rec[FR_argpo]), st.sW, st.cW, m);
    if (!az_any(tc != 0.0)) {
        st.sU = fma(st.sA, st.cW, st.cA * st.sW);
        st.cU = fma(st.cA, st.cW, -(st.sA * st.sW));
    } else {
        az_sincos_m(fma(rec[FR_mdot] + rec[FR_argpdot] + 2.0 * kappa * tc, t, rec[FR_mo] + rec[FR_argpo] - kappa * tc * tc), st.sU, st.cU, m);
    }
}

// seed the carried pairs with full sincos at time t (the step BEFORE the first one to be produced: every
// az_sgp4_fast_step call first advances the pairs by one increment); tc from az_fast_window
AZ_DEVICE void az_seed_fast(const double *__restrict__ el, size_t n_pad, size_t i, double t, double tc, FastCarry &st)
{
#define L(f) el[(size_t)F_##f * n_pad + i]
    az_sincos(fma(L(mdot), t, L(mo)), st.sA, st.cA);
    az_sincos(fma(L(argpdot), t, L(argpo)), st.sW, st.cW);
    if (!az_any(tc != 0.0)) {
        st.sU = fma(st.sA, st.cW, st.cA * st.sW);
        st.cU = fma(st.cA, st.cW, -(st.sA * st.sW));
    } else {
        const double kappa = L(no_unkozai) * L(t2cof);
        // (mo + argpo - kappa tc^2) + (mdot + argpdot + 2 kappa tc) t
        az_sincos(fma(L(mdot) + L(argpdot) + 2.0 * kappa * tc, t, L(mo) + L(argpo) - kappa * tc * tc), st.sU, st.cU);
    }
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
// Taylor coefficients of the rotation polynomials, read through an accessor: literals (the compiler keeps them
// where it likes: lane = satellite kernels, host emulation) or LDS words read by all lanes at once.  v_fma_f64
// takes no 64-bit literal and one scalar operand, so a Horner chain's constants otherwise end up parked in VGPRs
// for the whole loop -- 8 coefficients = 16 VGPRs, a sixth wave per SIMD in k_rows_fast.
enum RotCoef { RC_n6, RC_p24, RC_p120, RC_n720, RC_n5040, RC_p40320, RC_p362880, RC_n3628800, RC_p8,
               RC_n11f, RC_p12f, RC_p13f, RC_n14f, RC_n15f, RC_p16f, RC_NUM }; // ... -1/11!, 1/12!, 1/13!, -1/14!, -1/15!, 1/16!
#define AZ_RC_VALUES                                                                                                      \
    {-1.0 / 6.0, 1.0 / 24.0, 1.0 / 120.0, -1.0 / 720.0, -1.0 / 5040.0, 1.0 / 40320.0, 1.0 / 362880.0, -1.0 / 3628800.0, 0.125,     \
     -1.0 / 39916800.0, 1.0 / 479001600.0, 1.0 / 6227020800.0, -1.0 / 87178291200.0, -1.0 / 1307674368000.0,                   \
     1.0 / 20922789888000.0}
struct RotCoefLit {
    AZ_MEMBER double operator()(int k) const
    {
        constexpr double v[RC_NUM] = AZ_RC_VALUES;
        return v[k];
    }
};
struct RotCoefLds {
    const double *p;
    AZ_MEMBER double operator()(int k) const { return p[k]; }
};
template <class W>
AZ_DEVICE void az_rotcoef_store(W write)
{
    const RotCoefLit lit;
#pragma unroll
    for (int k = 0; k < RC_NUM; ++k) write(k, lit(k));
}

// |d| <= 2^-10: sin to d^3, cos to d^2 (dropped d^4/24 < 3.8e-14, d^5/120 < 8e-18)
template <class RC>
AZ_DEVICE void az_rotate_tiny2(double &s, double &c, double d, const RC &k)
{
    const double d2 = d * d;
    const double q = -0.5 * d2;
    const double p = fma(d2 * k(RC_n6), d, d);
    az_rot_apply2(s, c, p, q);
}
// (p,q) = (sin d, cos d - 1) of the 2^-7 tier (the polynomial of devmath.h's az_pq_small)
template <class RC>
AZ_DEVICE void az_fpq_small(double d, const RC &k, double &p, double &q)
{
    const double d2 = d * d;
    q = d2 * fma(d2, fma(d2, k(RC_n720), k(RC_p24)), -0.5);
    p = d * fma(d2, fma(d2, k(RC_p120), k(RC_n6)), 1.0);
}

// (p,q) for |d| <= 1/16: sin to d^7, cos to d^8 (d^9/9! < 3e-17, d^10/10! < 3e-19).
// Two instructions more than the 2^-7 tier; used for delomg + delm, which reaches 2^-7 for one member in fifty.
#define AZ_ROT_16TH 0.0625
template <class RC>
AZ_DEVICE void az_pq_16th(double d, const RC &k, double &p, double &q)
{
    const double d2 = d * d;
    q = d2 * fma(d2, fma(d2, fma(d2, k(RC_p40320), k(RC_n720)), k(RC_p24)), -0.5);
    p = d * fma(d2, fma(d2, fma(d2, k(RC_n5040), k(RC_p120)), k(RC_n6)), 1.0);
}

// (p,q) for |d| <= 1/2: sin to d^15, cos to d^16 (the polynomial of az_rotate_large, truncation 0.5^17/17! = 2e-20), the
// coefficients through the accessor: fifteen 64-bit literals otherwise sit in VGPR pairs across the whole loop
template <class RC>
AZ_DEVICE void az_pq_large(double d, const RC &k, double &p, double &q)
{
    const double d2 = d * d;
    q = fma(d2, k(RC_p16f), k(RC_n14f));
    q = fma(d2, q, k(RC_p12f));
    q = fma(d2, q, k(RC_n3628800));
    q = fma(d2, q, k(RC_p40320));
    q = fma(d2, q, k(RC_n720));
    q = fma(d2, q, k(RC_p24));
    q = d2 * fma(d2, q, -0.5);
    p = fma(d2, k(RC_n15f), k(RC_p13f));
    p = fma(d2, p, k(RC_n11f));
    p = fma(d2, p, k(RC_p362880));
    p = fma(d2, p, k(RC_n5040));
    p = fma(d2, p, k(RC_p120));
    p = fma(d2, p, k(RC_n6));
    p = d * fma(d2, p, 1.0);
}

// (p,q) for |d| <= 1/8: sin to d^9, cos to d^10 (the polynomial of az_rotate_med).  Used for the rest of the
// along-track drag term and for the node's motion across a time window.
template <class RC>
AZ_DEVICE void az_pq_med(double d, const RC &k, double &p, double &q)
{
    const double d2 = d * d;
    q = fma(d2, k(RC_n3628800), k(RC_p40320));
    q = fma(d2, q, k(RC_n720));
    q = fma(d2, q, k(RC_p24));
    q = d2 * fma(d2, q, -0.5);
    p = fma(d2, k(RC_p362880), k(RC_n5040));
    p = fma(d2, p, k(RC_p120));
    p = fma(d2, p, k(RC_n6));
    p = d * fma(d2, p, 1.0);
}


// (s,c) <- (sin,cos)(angle + delta) for a carried pair and its constant increment (sd,cd) = (sin,cos)(delta).
// SCALAR (one satellite per wave: sd, cd are SGPR pairs): written out as four in-place instructions.  Left to the
// compiler the same four operations come out as v_fmac (VOP2: the addend is the destination), whose result lands in
// the product's register, so that every loop-carried pair costs two v_mov_b64 per iteration to get back into place.
template <bool SCALAR>
AZ_DEVICE void az_pair_advance(double &s, double &c, double sd, double cd)
{
#ifndef AZ_HOST_EMUL
    if (SCALAR) {
        double m, m2;
        asm("v_mul_f64 %2, %1, %4\n\t"
            "v_mul_f64 %3, %0, %4\n\t"
            "v_fma_f64 %0, %0, %5, %2\n\t"
            "v_fma_f64 %1, %1, %5, -%3"
            : "+v"(s), "+v"(c), "=&v"(m), "=&v"(m2)
            : "s"(sd), "s"(cd));
        return;
    }
#endif
    const double ns = fma(s, cd, c * sd);
    c = fma(c, cd, -(s * sd));
    s = ns;
}

// (p,q) for |d| <= 1/16 where the rotated pair is only ever used scaled by the eccentricity (< 0.004 in the
// near-circular form): sin to d^5, cos to d^4 -- truncation d^7/5040 < 7.4e-13 and d^6/720 < 8.3e-11, times 0.004
template <class RC>
AZ_DEVICE void az_pq_ecc_scaled(double d, const RC &k, double &p, double &q)
{
    const double d2 = d * d;
    q = d2 * fma(d2, k(RC_p24), -0.5);
    p = d * fma(d2, fma(d2, k(RC_p120), k(RC_n6)), 1.0);
}
// (p,q) for |d| <= 0.0041 (the first Kepler correction of a near-circular orbit, el/(1 - el)): sin to d^3, cos to d^4
// (d^5/120 < 9.7e-15, d^6/720 < 6.6e-18)
template <class RC>
AZ_DEVICE void az_fpq_milli(double d, const RC &k, double &p, double &q)
{
    const double d2 = d * d;
    q = d2 * fma(d2, k(RC_p24), -0.5);
    p = fma(d2 * k(RC_n6), d, d);
}

// validation thresholds of the fast step
#define AZ_FAST_EL2 1.6e-5
#define AZ_FAST_TEMP2 6.0e-4

// Validation, once per time window instead of once per step.  The step below is straight-line code for the case
// "every small angle sits inside its polynomial tier, the orbit is near-circular (ECC = false)".  Each of those
// conditions is bounded here rigorously over the whole window [t_a, t_b] of tsince values from the satellite's
// constants alone (triangle inequalities: |cos| <= 1, |t| <= T, monotone reciprocal bounds), so a window that passes
// cannot violate any of them at any step, and the loop carries no compare, no vote and no branch for them (round 2
// voted on six compares per step).  A window that fails goes to the generic kernel as a whole; the bounds are a few
// per cent wider than the quantities themselves, which moves well under 1 % more segments there.
// ECC = true additionally validates its Newton iteration per step (az_sgp4_fast_step's return value).
// dmax: largest |delta_i| of a quasi-uniform grid (0: exactly uniform): every time bound widens by it, eps by |udot| dmax.
template <bool ECC>
AZ_DEVICE bool az_fast_window_ok(const FastK &k, const AzGrav &g, double t_a, double t_b, double dmax = 0.0)
{
    const double T = fmax(fabs(t_a), fabs(t_b)) + dmax;
    const double ae = fabs(k.eta_);
    // th = xmcof ((1 + eta cos M)^3 - (1 + eta cos mo)^3) + omgcof t
    const double th = fma(fabs(k.xmcof_), ae * fma(2.0 * ae, ae, 6.0), fabs(k.omgcof_) * T);
    // tempa = 1 - t (cc1 + t (d2 + t (d3 + t d4)))
    const double da = T * fma(T, fma(T, fma(T, fabs(k.d4_), fabs(k.d3_)), fabs(k.d2_)), fabs(k.cc1_));
    const double em = fabs(k.ecb_) + fma(fabs(k.bc4_), T, fabs(k.bc5_));
    bool ok = (th <= AZ_ROT_16TH) & (da <= 0.25) & (em <= 0.9);
    const double sam = k.sab_ * (1.0 - da);                  // sqrt(am) >= sam
    const double inv_am = 1.0 / (sam * sam);                 // 1/am <= inv_am
    const double temp = inv_am / fma(-em, em, 1.0);          // temp = 1/(am (1 - em^2))
    const double el = fma(temp, fabs(k.aycof_), em);         // el <= em + temp |aycof|
    const double el2 = el * el;
    if (!ECC) ok &= el2 <= AZ_FAST_EL2;
    ok &= el2 <= 0.81;
    // eps = temp xlcof axnl + nl2 (t - tc)^2 + t^3 (nl3 + t (nl4 + t nl5))
    const double dc = fmax(fabs(t_a - k.tc_), fabs(t_b - k.tc_)) + dmax;
    const double eps = fma(temp * fabs(k.xlcof_), em, fma(fabs(k.nl2_) * dc, dc, T * T * T * fma(T, fma(T, fabs(k.nl5_), fabs(k.nl4_)), fabs(k.nl3_)))) +
                       fabs(k.udot_) * dmax;
    ok &= eps <= AZ_ROT_MED;
    const double inv_pl = inv_am / (1.0 - el2);
    const double temp2 = g.half_j2 * inv_pl * inv_pl;
    ok &= temp2 <= AZ_FAST_TEMP2;
    // a_nd = k_node temp2 sin2u + nodedot (t - tmid) + xnodcf t^2
    const double a_nd = fma(fabs(k.k_node_), temp2, fma(fabs(k.nodedot_), 0.5 * fabs(t_b - t_a) + dmax, fabs(k.xnodcf_) * T * T));
    ok &= a_nd <= AZ_ROT_MED;
    return ok; // (every comparison is false for a NaN operand)
}

// one near-earth propagation on a uniform grid, inside a window az_fast_window_ok accepted.  Returns true when
// the eccentric form's Newton iteration left its assumptions for this lane (ECC = true only; the caller must then
// discard r/v and use az_sgp4_step); the near-circular form always returns false.
// DELTA (0 none, 1 tight, 2 wide): the grid is quasi-uniform; t is the ACTUAL time of this point and dl = t - (its place on the
// ideal grid the carried pairs advance along), |dl| <= AZ_DELTA_MAX (tight) / AZ_DELTA_WIDE_MAX (wide).
template <bool VEL, bool ECC = false, int DELTA = 0, class K = FastK, class RC = RotCoefLit>
AZ_DEVICE bool az_sgp4_fast_step(const K &k, const AzGrav &g, const RC &rk, double t, FastCarry &st, double r[3],
                                  double v[3], double dl = 0.0)
{
    // advance the carried pairs by their constant increments
    az_pair_advance<K::SCALAR>(st.sA, st.cA, k.sdA(), k.cdA());
    az_pair_advance<K::SCALAR>(st.sW, st.cW, k.sdW(), k.cdW());
    az_pair_advance<K::SCALAR>(st.sU, st.cU, k.sdU(), k.cdU());
    double sA = st.sA, cA = st.cA, sW = st.sW, cW = st.cW;
    if constexpr ((DELTA == 1 && ECC) || DELTA == 2) {
        // M and W at the actual time.  Tight form: first order in dl, eccentric members only (the near-circular form does
        // without: both pairs only enter scaled by em < 0.004 -- argpdot dl em < 1e-12 rad -- and through th, whose product
        // with em is bounded by the drag coefficients alone: d(th) em < 1e-13).  Wide form: every member; M of an eccentric
        // member by the 1/16-rad polynomial (|mdot dl| <= 0.0375).
        const double a = k.mdot() * dl, b = k.argpdot() * dl;
        if constexpr (DELTA == 2 && ECC) {
            double pa, qa;
            az_pq_16th(a, rk, pa, qa);
            az_rot_apply2(sA, cA, pa, qa);
        } else if constexpr (DELTA == 2) {
            // second order: cos M feeds th = delomg + delm, which turns the eccentricity vector -- for a low, high-drag member
            // th reaches 1e-2 rad, and a^2 / 2 = 7e-4 of it times em is a micrometre-per-second error; a^4 / 24 is not
            az_rotate_tiny2(sA, cA, a, rk);
        } else {
            sA = fma(st.cA, a, st.sA); cA = fma(-st.sA, a, st.cA);
        }
        if constexpr (DELTA == 2 && ECC) az_rotate_tiny2(sW, cW, b, rk); // (b^2 / 2 = 5e-10 times em = 0.25 would be decimetres)
        else { sW = fma(st.cW, b, st.sW); cW = fma(-st.sW, b, st.cW); }
    }
    const double t2 = t * t;

    // secular gravity + drag (Sgp4Batch.zig L121-154)
    const double dm = fma(k.eta(), cA, 1.0);
    const double th = fma(k.xmcof(), dm * dm * dm, fma(k.omgcof(), t, -k.xd())); // delomg + delm
    const double tempa = fma(-t, fma(t, fma(t, fma(t, k.d4(), k.d3()), k.d2()), k.cc1()), 1.0);
    // no_unkozai * templ without the part carried by U: kappa (t - tc)^2 + t^3 (nl3 + t (nl4 + t nl5))
    const double dtc = t - k.tc();
    double nl = fma(k.nl2() * dtc, dtc, t2 * (t * fma(t, fma(t, k.nl5(), k.nl4()), k.nl3())));
    // U at the actual time: udot dl joins the angle of the small rotation U takes further down -- added HERE, where dl is still
    // at hand (kept live down to that rotation it costs the tile kernel, which runs at its register limit, two VGPRs)
    if constexpr (DELTA != 0) nl = fma(k.udot(), dl, nl);
    bool bad = false;
    double p, q;
    if (ECC) az_pq_16th(th, rk, p, q);
    else az_pq_ecc_scaled(th, rk, p, q);                       // M + th and W - th only enter scaled by em < 0.004
    const double smm = fma(cA, p, fma(sA, q, sA));            // sin(M + th)
    const double sw = fma(-cW, p, fma(sW, q, sW));            // (sin,cos)(W - th)
    const double cw = fma(sW, p, fma(cW, q, cW));
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
    // u0 = U + (rest of no*templ) + temp*xlcof*axnl: |.| <= 1/8, sin to d^7 and cos to d^8 (d^9/9! < 2.1e-14)
    double s = st.sU, c = st.cU;
    {
        const double eps = fma(temp * k.xlcof(), axnl, nl);
        az_pq_16th(eps, rk, p, q);
        az_rot_apply2(s, c, p, q);
    }

    const double el2 = fma(axnl, axnl, aynl * aynl);
    double ecose, esine, ome, inv_ome, betal, inv_omel2, inv_1pb;
    if (!ECC) {
        // Kepler, near-circular form (see az_kepler_posvel): Newton step from E0 = u, chord step with the
        // same reciprocal, first-order rotation by the second correction
        const double rden = az_rcp1(fma(-s, aynl, fma(-c, axnl, 1.0)));
        const double d0 = fma(axnl, s, -(aynl * c)) * rden;
        az_fpq_milli(d0, rk, p, q); // |d0| <= el/(1-el) < 0.0041
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
        // 1/(1 - ecose) from the reciprocal of the Newton step (1 - ecose before the two rotations: off by
        // x ~ el (d0 + d1) <= 1.7e-5 relative): rden (1 + x + x^2), x = 1 - ome rden   (x^3 < 5e-15)
        const double x = fma(-ome, rden, 1.0);
        inv_ome = fma(rden, fma(x, x, x), rden);
        betal = fma(el2, fma(el2, -0.125, -0.5), 1.0);    // sqrt(1-x)   (- x^3/16)
        inv_omel2 = el2 + 1.0;                            // 1/(1-x): x^2 < 2.6e-10 relative on the J2 terms (< 1e-3)
        inv_1pb = fma(el2, rk(RC_p8), 0.5);               // 1/(1+betal): x^2/16 < 1.6e-11, times esine < 0.004
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
                az_pq_large(d, rk, p, q);
                az_rot_apply2(s, c, p, q);
            } else if (it == 1) {
                bad |= !(fabs(d) <= AZ_ROT_MED);
                az_pq_med(d, rk, p, q);
                az_rot_apply2(s, c, p, q);
            } else {
                bad |= !(fabs(d) <= AZ_ROT_SMALL);
                az_fpq_small(d, rk, p, q);
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

    const double mrt = fma(rl, fma(k.k_mrt() * temp2, betal, 1.0), k.k_c2u() * temp1 * cos2u);
    const double t2s = temp2 * sin2u;
    // J2 short-period corrections as tiny rotations (each bounded by 1.5 temp2 <= 9e-4); the node's own motion
    // about the window centre, nodedot (t - tmid) + xnodcf t^2, rides on the node correction
    const double a_nd = fma(k.k_node(), t2s, fma(k.nodedot(), t - k.tmid(), k.xnodcf() * t2));
    double ssu = sinu, csu = cosu, sn = k.sOc(), cn = k.cOc(), si = k.sinio(), ci = k.cosio();
    az_rotate_tiny2(ssu, csu, k.k_su() * t2s, rk);
    az_pq_16th(a_nd, rk, p, q); // J2 correction + the node's motion across the window (up to ~0.03 rad over +-400 min; <= 1/8)
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
