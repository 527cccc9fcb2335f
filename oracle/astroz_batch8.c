/*
 * astroz_batch8.c -- CPU BASELINE: a restatement of the reference's multithreaded SIMD CPU *design*
 * for the near-earth constellation path (TEST / BENCHMARK INFRASTRUCTURE ONLY, like the rest of
 * oracle/: only tests/ and bench.py's cpu_baseline leg may use it; never the product).
 *
 * The reference vectorises 8 satellites per AVX-512 register (Sgp4Batch.BatchElements(8),
 * src/Sgp4Batch.zig L15-110), evaluates the whole step branch-free on 8 lanes with polynomial
 * sin/cos and atan2 (src/simdMath.zig L29-171), leaves the Kepler-Newton loop when ALL lanes have
 * converged (src/Sgp4.zig L693), and threads over time ranges (time-major output) or batch ranges
 * (satellite-major output) (src/Constellation.zig L327-434).  This file restates that design in C
 * with GCC vector extensions (8 x fp64 = one zmm register with -march=native on AVX-512 hosts);
 * every function cites the lines it follows.  It is NOT the parity oracle: its atan2 polynomial is
 * only good to ~1e-7 rad, exactly like the reference's (simdMath.zig L124-125) -- the reference's own
 * batch tests assert 0.01 km / 1e-6 km/s (src/Sgp4Batch.zig L259-296), and tests/test_oracle_golden.py
 * holds this file to the same tolerance against the scalar oracle.
 */
#include "astroz_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef double v8 __attribute__((vector_size(64)));
typedef long long v8i __attribute__((vector_size(64)));
#define NL 8

static inline v8 splat(double x) { return (v8){x, x, x, x, x, x, x, x}; }
static inline v8i splati(long long x) { return (v8i){x, x, x, x, x, x, x, x}; }
/* @select(f64, mask, a, b): mask lanes are all-ones / zero (GCC vector comparison result) */
static inline v8 sel(v8i m, v8 a, v8 b) { return (v8)(((v8i)a & m) | ((v8i)b & ~m)); }
static inline v8 vabs(v8 x) { return (v8)((v8i)x & splati(0x7fffffffffffffffLL)); }
static inline v8 vmax(v8 a, v8 b) { return sel(a > b, a, b); }
static inline v8 vmin(v8 a, v8 b) { return sel(a < b, a, b); }
static inline v8 vsqrt(v8 x)
{
    v8 r;
    for (int i = 0; i < NL; i++) r[i] = sqrt(x[i]); /* vsqrtpd */
    return r;
}
static inline v8 vfloor(v8 x)
{
    v8 r;
    for (int i = 0; i < NL; i++) r[i] = floor(x[i]); /* vrndscalepd */
    return r;
}
static inline int all_lt(v8 a, v8 b)
{
    v8i m = a < b;
    long long acc = -1;
    for (int i = 0; i < NL; i++) acc &= m[i];
    return acc != 0;
}
static inline int any_lt(v8 a, v8 b)
{
    v8i m = a < b;
    long long acc = 0;
    for (int i = 0; i < NL; i++) acc |= m[i];
    return acc != 0;
}

/* simdMath.sincosN, src/simdMath.zig L29-95: k = round(x 2/pi) by the 1.5*2^52 trick, two-term
 * Cody-Waite reduction, degree-13 / degree-12 polynomials, quadrant fix-up with integer sign flips */
static inline void sincos8(v8 x, v8 *s, v8 *c)
{
    const v8 magic = splat(6755399441055744.0);
    const v8 kr = x * splat(0.63661977236758134308) + magic - magic;
    v8 r = x - splat(1.5707963267948966) * kr;
    r = r - splat(6.123233995736766e-17) * kr;
    v8i k;
    for (int i = 0; i < NL; i++) k[i] = (long long)kr[i];
    const v8 r2 = r * r;
    v8 sp = splat(1.6058936490373178302326e-10) * r2 + splat(-2.5052106798274583895303e-08);
    sp = sp * r2 + splat(2.7557319210152756118515e-06);
    sp = sp * r2 + splat(-1.9841269841201840457725e-04);
    sp = sp * r2 + splat(8.3333333333333225058715e-03);
    sp = sp * r2 + splat(-1.6666666666666666574148e-01);
    const v8 sr = sp * (r2 * r) + r;
    v8 cp = splat(2.0876756987868089233269e-09) * r2 + splat(-2.7557319223933824788682e-07);
    cp = cp * r2 + splat(2.4801587301587286645498e-05);
    cp = cp * r2 + splat(-1.3888888888888872762458e-03);
    cp = cp * r2 + splat(4.1666666666666665319411e-02);
    cp = cp * r2 + splat(-4.9999999999999999999583e-01);
    const v8 cr = cp * r2 + splat(1.0);
    const v8i swap = (k & splati(1)) != splati(0);
    v8 ss = sel(swap, cr, sr), cc = sel(swap, sr, cr);
    const v8i ssign = (k & splati(2)) << 62;
    const v8i csign = ((k + splati(1)) & splati(2)) << 62;
    *s = (v8)((v8i)ss ^ ssign);
    *c = (v8)((v8i)cc ^ csign);
}

/* simdMath.modTwoPiN, L110-121 */
static inline v8 mod2pi8(v8 x)
{
    const double twopi = 6.283185307179586476925287;
    const v8 n = vfloor(x * splat(1.0 / twopi));
    v8 r = x - splat(twopi) * n;
    return sel(r < splat(0.0), r + splat(twopi), r);
}

/* simdMath.atan2N, L124-171: min/max ratio, odd degree-17 polynomial (~1e-7 rad), octant fix-up */
static inline v8 atan2_8(v8 y, v8 x)
{
    const v8 ax = vabs(x), ay = vabs(y);
    const v8 mx = vmax(ax, ay), mn = vmin(ax, ay);
    const v8 t = mn / vmax(mx, splat(1.0e-30));
    const v8 t2 = t * t;
    v8 a = splat(0.0028662257);
    a = a * t2 + splat(-0.0161657367);
    a = a * t2 + splat(0.0429096138);
    a = a * t2 + splat(-0.0752896400);
    a = a * t2 + splat(0.1065626393);
    a = a * t2 + splat(-0.1420889944);
    a = a * t2 + splat(0.1999355085);
    a = a * t2 + splat(-0.3333314528);
    a = a * t2 + splat(1.0);
    a = a * t;
    a = sel(ay > ax, splat(1.57079632679489661923) - a, a);
    a = sel(x < splat(0.0), splat(3.14159265358979323846) - a, a);
    return sel(y < splat(0.0), -a, a);
}

/* Sgp4Batch.BatchElements(8), src/Sgp4Batch.zig L15-75 (35 per-satellite fields + isimp mask) */
typedef struct {
    v8 mo, mdot, argpo, argpdot, nodeo, nodedot, xnodcf;
    v8 cc1, bc4, bc5, t2cof, omgcof, eta, xmcof, delmo, sinmao, d2, d3, d4, t3cof, t4cof, t5cof;
    v8 a_base, ecco, no_unkozai, inclo, aycof, xlcof, con41, x1mth2, x7thm1, sinio, cosio;
    v8 isimp; /* 0.0 / 1.0 */
    v8 offset; /* epoch offset minutes (Constellation.zig L423-426) */
} batch8;

/* Sgp4.keplerAndPosVel, src/Sgp4.zig L646-750 (N = 8) */
static inline void kepler_posvel8(const orc_grav *g, double vkmpersec, v8 am, v8 em, v8 mm, v8 argpm, v8 nodem,
                                  const batch8 *b, v8 out[6])
{
    const v8 one = splat(1.0);
    const v8 temp = one / (am * (one - em * em));
    v8 sa, ca;
    sincos8(argpm, &sa, &ca);
    const v8 axnl = em * ca;
    const v8 aynl = em * sa + temp * b->aycof;
    const v8 xl = mm + argpm + nodem + temp * b->xlcof * axnl;
    v8 u = xl - nodem;
    v8 eo1 = u, se = splat(0.0), ce = one;
    for (int it = 0; it < 10; it++) {
        sincos8(eo1, &se, &ce);
        const v8 delta = (u - aynl * ce + axnl * se - eo1) / (one - ce * axnl - se * aynl);
        eo1 = eo1 + vmax(splat(-0.95), vmin(splat(0.95), delta));
        if (all_lt(vabs(delta), splat(1.0e-12))) break; /* @reduce(.And, ...), L693 */
    }
    const v8 ecose = axnl * ce + aynl * se;
    const v8 esine = axnl * se - aynl * ce;
    const v8 el2 = axnl * axnl + aynl * aynl;
    const v8 pl = am * (one - el2);
    const v8 betal = vsqrt(one - el2);
    const v8 rl = am * (one - ecose);
    const v8 sqam = vsqrt(am);
    const v8 rdotl = sqam * esine / rl;
    const v8 rvdotl = vsqrt(pl) / rl;
    const v8 aor = am / rl;
    const v8 est = esine / (one + betal);
    const v8 sinu = aor * (se - aynl - axnl * est);
    const v8 cosu = aor * (ce - axnl + aynl * est);
    u = atan2_8(sinu, cosu);
    const v8 sin2u = splat(2.0) * sinu * cosu;
    const v8 cos2u = one - splat(2.0) * sinu * sinu;
    const v8 temp1 = splat(0.5 * g->j2) / pl;
    const v8 temp2 = temp1 / pl;
    const v8 nm = splat(g->xke) / (am * sqam); /* pow15N, simdMath.zig L174-176 */
    const v8 mrt = rl * (one - splat(1.5) * temp2 * betal * b->con41) + splat(0.5) * temp1 * b->x1mth2 * cos2u;
    const v8 su = u - splat(0.25) * temp2 * b->x7thm1 * sin2u;
    const v8 xnode = nodem + splat(1.5) * temp2 * b->cosio * sin2u;
    const v8 xinc = b->inclo + splat(1.5) * temp2 * b->cosio * b->sinio * cos2u;
    const v8 mvt = rdotl - nm * temp1 * b->x1mth2 * sin2u / splat(g->xke);
    const v8 rvdot = rvdotl + nm * temp1 * (b->x1mth2 * cos2u + splat(1.5) * b->con41) / splat(g->xke);
    v8 ssu, csu, sn, cn, si, ci;
    sincos8(su, &ssu, &csu);
    sincos8(xnode, &sn, &cn);
    sincos8(xinc, &si, &ci);
    const v8 xmx = -sn * ci, xmy = cn * ci;
    const v8 ux = xmx * ssu + cn * csu, uy = xmy * ssu + sn * csu, uz = si * ssu;
    const v8 vx = xmx * csu - cn * ssu, vy = xmy * csu - sn * ssu, vz = si * csu;
    const v8 rs = mrt * splat(g->radius_km);
    out[0] = rs * ux;
    out[1] = rs * uy;
    out[2] = rs * uz;
    out[3] = (mvt * ux + rvdot * vx) * splat(vkmpersec);
    out[4] = (mvt * uy + rvdot * vy) * splat(vkmpersec);
    out[5] = (mvt * uz + rvdot * vz) * splat(vkmpersec);
}

/* Sgp4Batch.propagateBatchDirect, src/Sgp4Batch.zig L113-157; returns 0, or 1 when any lane decayed
 * (the reference fails the whole batch, L147) */
static inline int step8(const orc_grav *g, double vkmpersec, const batch8 *b, v8 ts, v8 out[6])
{
    const v8 one = splat(1.0), zero = splat(0.0);
    const v8 t2 = ts * ts, t3 = t2 * ts, t4 = t3 * ts;
    v8 tempa = one - b->cc1 * ts;
    v8 tempe = b->bc4 * ts;
    v8 templ = b->t2cof * t2;
    const v8 xmdf = b->mo + b->mdot * ts;
    const v8 argpdf = b->argpo + b->argpdot * ts;
    v8 nodem = b->nodeo + b->nodedot * ts + b->xnodcf * t2;
    v8 sx, cx;
    sincos8(xmdf, &sx, &cx);
    const v8 dmt = one + b->eta * cx;
    const v8 ho = b->omgcof * ts + b->xmcof * (dmt * dmt * dmt - b->delmo);
    const v8i hom = b->isimp == zero;
    v8 mm = xmdf + sel(hom, ho, zero);
    v8 argpm = argpdf - sel(hom, ho, zero);
    tempa = sel(hom, tempa - b->d2 * t2 - b->d3 * t3 - b->d4 * t4, tempa);
    v8 smm, cmm;
    sincos8(mm, &smm, &cmm);
    tempe = sel(hom, tempe + b->bc5 * (smm - b->sinmao), tempe);
    templ = sel(hom, templ + b->t3cof * t3 + t4 * (b->t4cof + ts * b->t5cof), templ);
    const v8 am = b->a_base * tempa * tempa;
    const v8 em = vmax(b->ecco - tempe, splat(1.0e-6));
    if (any_lt(em, splat(1.0e-6))) return 1;
    mm = mod2pi8(mm + b->no_unkozai * templ);
    nodem = mod2pi8(nodem);
    argpm = mod2pi8(argpm);
    kepler_posvel8(g, vkmpersec, am, em, mm, argpm, nodem, b, out);
    return 0;
}

/* Sgp4Batch.initBatchElements (L78-110) from already-initialised scalar elements; the last batch is
 * padded with copies of the last satellite (Constellation.zig L145-147) */
static void fill_batch(batch8 *b, const orc_sat *sats, const double *offsets, size_t base, size_t n)
{
    for (int l = 0; l < NL; l++) {
        const size_t i = base + (size_t)l < n ? base + (size_t)l : n - 1;
        const orc_sat *s = &sats[i];
#define F(dst, src) b->dst[l] = s->src
        F(mo, mo); F(mdot, mdot); F(argpo, argpo); F(argpdot, argpdot); F(nodeo, nodeo); F(nodedot, nodedot);
        F(xnodcf, xnodcf); F(cc1, cc1); F(t2cof, t2cof); F(omgcof, omgcof); F(eta, eta); F(xmcof, xmcof);
        F(delmo, delmo); F(sinmao, sinmao); F(d2, d2); F(d3, d3); F(d4, d4); F(t3cof, t3cof); F(t4cof, t4cof);
        F(t5cof, t5cof); F(a_base, a_base); F(ecco, ecco); F(no_unkozai, no_unkozai); F(inclo, inclo);
        F(aycof, aycof); F(xlcof, xlcof); F(con41, con41); F(x1mth2, x1mth2); F(x7thm1, x7thm1); F(sinio, sinio);
        F(cosio, cosio);
#undef F
        b->bc4[l] = s->bstar * s->cc4;
        b->bc5[l] = s->bstar * s->cc5;
        b->isimp[l] = s->isimp ? 1.0 : 0.0;
        b->offset[l] = offsets ? offsets[i] : 0.0;
    }
}

/* Constellation.propagateConstellation (L541-605) + propagateImpl / unifiedSgp4Range / sgp4Core
 * (L327-434): near-earth members only, TEME, both layouts; threads over time ranges (time-major) or
 * batch ranges (satellite-major).  Satellites must be near-earth and initialised (orc_sat_init == 0).
 * Returns the number of failed (zero-filled) batch steps. */
size_t orc_batch8_propagate(const orc_sat *sats, size_t n_sats, const double *times, size_t n_times,
                            const double *offsets, double *pos, double *vel, int layout, size_t stride, int nthreads)
{
    if (n_sats == 0 || n_times == 0) return 0;
    if (stride == 0) stride = n_sats;
    const size_t nb = (n_sats + NL - 1) / NL;
    batch8 *bt = (batch8 *)aligned_alloc(64, sizeof(batch8) * nb);
    for (size_t k = 0; k < nb; k++) fill_batch(&bt[k], sats, offsets, k * NL, n_sats);
    const orc_grav g = sats[0].g;
    const double vk = sats[0].vkmpersec;
    size_t failed = 0;
    const long work = (long)(layout == ORC_SAT_MAJOR ? nb : n_times);
    (void)nthreads;
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(nthreads > 1 ? nthreads : 1) reduction(+ : failed)
#endif
    for (long w = 0; w < work; w++) {
        const size_t k0 = layout == ORC_SAT_MAJOR ? (size_t)w : 0, k1 = layout == ORC_SAT_MAJOR ? (size_t)w + 1 : nb;
        const size_t t0 = layout == ORC_SAT_MAJOR ? 0 : (size_t)w, t1 = layout == ORC_SAT_MAJOR ? n_times : (size_t)w + 1;
        for (size_t k = k0; k < k1; k++)
            for (size_t t = t0; t < t1; t++) {
                v8 out[6];
                const int rc = step8(&g, vk, &bt[k], splat(times[t]) + bt[k].offset, out);
                if (rc) {
                    failed++;
                    for (int c = 0; c < 6; c++) out[c] = splat(0.0);
                }
                for (int l = 0; l < NL; l++) {
                    const size_t s = k * NL + (size_t)l;
                    if (s >= n_sats) break;
                    const size_t ob = (layout == ORC_SAT_MAJOR) ? (s * n_times + t) * 3 : (t * stride + s) * 3;
                    pos[ob] = out[0][l];
                    pos[ob + 1] = out[1][l];
                    pos[ob + 2] = out[2][l];
                    if (vel) {
                        vel[ob] = out[3][l];
                        vel[ob + 1] = out[4][l];
                        vel[ob + 2] = out[5][l];
                    }
                }
            }
    }
    free(bt);
    return failed;
}
