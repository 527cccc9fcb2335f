// fast_step_f32.h -- the branch-free uniform-grid step (fast_step.h) in fp32 arithmetic, for fp32 OUTPUTS
// (BASELINE config 5: 1M satellites x 10,000 steps, an HBM-bound stress case; the reference itself is fp64 only).
//
// What stays fp64: the three carried angle pairs M, W, U (their phase runs to hundreds of radians over 10,000
// minutes: 12 fp64 instructions per step) and the
// along-radius chain a -> r (the semi-major axis, 1 - e cos E and the J2 radius factor: ~12) -- together ~30 fp64
// instructions.  Everything else -- drag polynomials, Kepler step, short-period terms, orientation, velocity --
// is fp32 (v_fma_f32 issues at twice the fp64 rate on gfx950; v_rcp_f32 needs no refinement at this precision):
// ~170 instructions.  Near-circular members only; eccentric members and validation failures take the fp64 kernels
// with rounded stores.  Result accuracy is that of fp32 storage plus ~1e-7 relative from the fp32 chain: metres,
// mm/s (tests/test_gpu_round2.py::test_fp32_arithmetic_*).
#pragma once
#include "fast_step.h"

struct FastK32 {
#define X(n) float n##_;
    AZ_FASTK_COLD(X) AZ_FASTK_HOT(X)
#undef X
    double sab64, cc1d, d2d, d3d, d4d;  // fp64: the along-radius chain (sqrt(a_base), drag polynomial of the semi-major axis)
    double sdA, cdA, sdW, cdW, sdU, cdU, tc, tmid; // fp64: the carried-angle increments, window centres
};

AZ_DEVICE void az_load_fast32(const FastK &k, FastK32 &f)
{
#define X(n) f.n##_ = (float)k.n##_;
    AZ_FASTK_COLD(X) AZ_FASTK_HOT(X)
#undef X
    f.sab64 = k.sab_; f.cc1d = k.cc1_; f.d2d = k.d2_; f.d3d = k.d3_; f.d4d = k.d4_;
    f.sdA = k.sdA_; f.cdA = k.cdA_; f.sdW = k.sdW_; f.cdW = k.cdW_;
    f.sdU = k.sdU_; f.cdU = k.cdU_; f.tc = k.tc_; f.tmid = k.tmid_;
}

#ifdef AZ_HOST_EMUL
static inline float az_rcp32(float x) { return 1.0f / x; }
#else
AZ_DEVICE float az_rcp32(float x) { return __builtin_amdgcn_rcpf(x); }
#endif

// (s,c) <- rotated by d, |d| <= 2^-10: sin d = d, cos d - 1 = -d^2/2 to fp32 precision (d^3/6 < 1.6e-10)
AZ_DEVICE void az_rot32_tiny(float &s, float &c, float d)
{
    const float q = -0.5f * d * d;
    const float ns = fmaf(c, d, fmaf(s, q, s));
    c = fmaf(-s, d, fmaf(c, q, c));
    s = ns;
}
// |d| <= 1/8: sin to d^5, cos to d^6 (next terms 4.5e-11, 1.5e-12)
AZ_DEVICE void az_rot32_med(float &s, float &c, float d)
{
    const float d2 = d * d;
    const float p = d * fmaf(d2, fmaf(d2, 1.0f / 120.0f, -1.0f / 6.0f), 1.0f);
    const float q = d2 * fmaf(d2, fmaf(d2, -1.0f / 720.0f, 1.0f / 24.0f), -0.5f);
    const float ns = fmaf(c, p, fmaf(s, q, s));
    c = fmaf(-s, p, fmaf(c, q, c));
    s = ns;
}
// |d| <= 2^-7: sin to d^3, cos to d^4 (next terms 2.4e-13, 3e-16)
AZ_DEVICE void az_rot32_small(float &s, float &c, float d)
{
    const float d2 = d * d;
    const float p = fmaf(d2 * (-1.0f / 6.0f), d, d);
    const float q = d2 * fmaf(d2, 1.0f / 24.0f, -0.5f);
    const float ns = fmaf(c, p, fmaf(s, q, s));
    c = fmaf(-s, p, fmaf(c, q, c));
    s = ns;
}

template <bool VEL>
AZ_DEVICE bool az_sgp4_fast_step_f32(const FastK32 &k, const AzGrav &g, double t64, FastCarry &st, float r[3], float v[3])
{
    // carried pairs: fp64 (the phase is hundreds of radians; a per-step fp32 rounding would walk away)
    {
        const double nsA = fma(st.sA, k.cdA, st.cA * k.sdA);
        st.cA = fma(st.cA, k.cdA, -(st.sA * k.sdA));
        st.sA = nsA;
        const double nsW = fma(st.sW, k.cdW, st.cW * k.sdW);
        st.cW = fma(st.cW, k.cdW, -(st.sW * k.sdW));
        st.sW = nsW;
        const double nsU = fma(st.sU, k.cdU, st.cU * k.sdU);
        st.cU = fma(st.cU, k.cdU, -(st.sU * k.sdU));
        st.sU = nsU;
    }
    const float t = (float)t64;
    const float sA = (float)st.sA, cA = (float)st.cA, sW = (float)st.sW, cW = (float)st.cW;
    const float t2 = t * t;

    const float dm = fmaf(k.eta_, cA, 1.0f);
    const float th = fmaf(k.xmcof_, dm * dm * dm, fmaf(k.omgcof_, t, -k.xd_));
    const double tempa64 = fma(-t64, fma(t64, fma(t64, fma(t64, k.d4d, k.d3d), k.d2d), k.cc1d), 1.0);
    const float dtc = (float)(t64 - k.tc);
    const float nl = fmaf(k.nl2_ * dtc, dtc, t2 * (t * fmaf(t, fmaf(t, k.nl5_, k.nl4_), k.nl3_)));
    bool bad = !(fabsf(th) <= (float)AZ_ROT_16TH);
    float smm = sA, cmm = cA, sw = sW, cw = cW;
    az_rot32_med(smm, cmm, th);
    az_rot32_med(sw, cw, -th);
    const float em = fmaxf(fmaf(-k.bc5_, smm, fmaf(-k.bc4_, t, k.ecb_)), 1.0e-6f);

    // along-radius chain in fp64: sqrt(am) = sqrt(a_base) |tempa|
    const double sqrt_am64 = k.sab64 * fabs(tempa64);
    const double am64 = sqrt_am64 * sqrt_am64;
    const float sqrt_am = (float)sqrt_am64;
    const float omem2 = fmaf(-em, em, 1.0f);
    const float R = az_rcp32(sqrt_am * omem2);
    const float ra = R * omem2;
    const float temp = ra * R;

    const float axnl = em * cw;
    const float aynl = fmaf(em, sw, temp * k.aycof_);
    // u0 = U (carried in fp64) + the rest of the drag term + the long-period term
    float s = (float)st.sU, c = (float)st.cU;
    {
        const float eps = fmaf(temp * k.xlcof_, axnl, nl);
        bad |= !(fabsf(eps) <= (float)AZ_ROT_MED);
        az_rot32_med(s, c, eps);
    }

    // Kepler, near-circular: one Newton step from E0 = u (the next correction, (el/2) d0^2 <= 3.2e-8, is below
    // fp32 resolution)
    const float el2 = fmaf(axnl, axnl, aynl * aynl);
    bad |= !(el2 <= (float)AZ_FAST_EL2);
    const float rden = az_rcp32(fmaf(-s, aynl, fmaf(-c, axnl, 1.0f)));
    const float d0 = fmaf(axnl, s, -(aynl * c)) * rden;
    az_rot32_small(s, c, d0);
    const float ecose = fmaf(axnl, c, aynl * s);
    const float esine = fmaf(axnl, s, -(aynl * c));
    const float ome = 1.0f - ecose;
    const float inv_ome = az_rcp32(ome);
    const float betal = fmaf(el2, -0.5f, 1.0f);
    const float inv_omel2 = 1.0f + el2;
    const float inv_1pb = fmaf(el2, 0.125f, 0.5f);

    const float est = esine * inv_1pb;
    const float sinu = inv_ome * (s - fmaf(axnl, est, aynl));
    const float cosu = inv_ome * (c + fmaf(aynl, est, -axnl));
    const float sin2u = (sinu + sinu) * cosu;
    const float cos2u = fmaf(-2.0f * sinu, sinu, 1.0f);

    const float inv_am = ra * ra;
    const float inv_pl = inv_am * inv_omel2;
    const float temp1 = (float)g.half_j2 * inv_pl;
    const float temp2 = temp1 * inv_pl;
    bad |= !(temp2 <= (float)AZ_FAST_TEMP2);

    // mrt = rl (1 + k_mrt temp2 betal) + k_c2u temp1 cos2u, rl = am (1 - ecose): fp64 (cancellation-free, but the
    // result is the radius itself: an fp32 chain here costs a metre)
    const double rl64 = am64 * (1.0 - (double)ecose);
    const double mrt64 = fma(rl64, (double)fmaf(k.k_mrt_ * temp2, betal, 1.0f), (double)(k.k_c2u_ * temp1 * cos2u));
    const float t2s = temp2 * sin2u;
    const float a_nd = fmaf(k.k_node_, t2s, fmaf(k.nodedot_, (float)(t64 - k.tmid), k.xnodcf_ * t2));
    bad |= !(fabsf(a_nd) <= (float)AZ_ROT_MED);
    float ssu = sinu, csu = cosu, sn = k.sOc_, cn = k.cOc_, si = k.sinio_, ci = k.cosio_;
    az_rot32_tiny(ssu, csu, k.k_su_ * t2s);
    az_rot32_med(sn, cn, a_nd);
    az_rot32_tiny(si, ci, k.k_inc_ * temp2 * cos2u);

    const float xmx = -sn * ci, xmy = cn * ci;
    const float ux = fmaf(xmx, ssu, cn * csu);
    const float uy = fmaf(xmy, ssu, sn * csu);
    const float uz = si * ssu;
    const float rs = (float)(mrt64 * g.radius_km);
    r[0] = rs * ux;
    r[1] = rs * uy;
    r[2] = rs * uz;
    if (VEL) {
        const float rv = ra * (float)g.vkmpersec;
        const float vk = rv * inv_ome;
        const float nxt = rv * inv_am * temp1;
        const float mvt = fmaf(-nxt * k.x1mth2_, sin2u, vk * esine);
        const float rvdot = fmaf(nxt, fmaf(k.x1mth2_, cos2u, k.k_rv_), vk * betal);
        const float vx = fmaf(xmx, csu, -(cn * ssu));
        const float vy = fmaf(xmy, csu, -(sn * ssu));
        const float vz = si * csu;
        v[0] = fmaf(mvt, ux, rvdot * vx);
        v[1] = fmaf(mvt, uy, rvdot * vy);
        v[2] = fmaf(mvt, uz, rvdot * vz);
    }
    return bad;
}
