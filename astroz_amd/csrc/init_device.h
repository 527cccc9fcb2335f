// init_device.h -- per-satellite element initialisation, one lane per satellite.
//
// Computes what the reference computes once per TLE on the host:
//   Sgp4.initElements  (src/Sgp4.zig L108-180: extractMeanElements L192-202, recoverMeanMotion
//                       L206-228, computeTrigTerms/PolyTerms L232-249, computeSecularRates L253-284,
//                       computeDragCoefficients L301-382, computeHigherOrderDrag L394-417)
//   Sdp4.initElements  (src/Sdp4.zig L174-274: gstime L277-285, dscom L344-499, dsinit L525-657)
//   classification     (Constellation.zig L115-126: period > 225 min -> deep space)
// and writes the rows of the device element table (fields.h).  Running it on the GPU keeps every
// floating-point operation of the product on the device and makes catalog ingest (SURVEY 8f4:
// 1M TLEs for config 5) a single coalesced pass.
#pragma once
#include "devmath.h"
#include "fields.h"

AZ_DEVICE double az_poly3(double x, double c0, double c1, double c2, double c3)
{
    // same term order as the reference's polyEval (sum of c_i * x^i, ascending)
    double r = c0, xn = x;
    r += c1 * xn;
    xn *= x;
    r += c2 * xn;
    xn *= x;
    r += c3 * xn;
    return r;
}

AZ_DEVICE double az_gstime(double jdut1)
{
// plain IEEE operations, no FMA contraction: the argument reaches ~1.5e5 rad before the modulo, so a
// single fused rounding would move the result by 3e-11 rad relative to any host evaluation
#pragma clang fp contract(off)
    const double tut1 = (jdut1 - 2451545.0) / 36525.0;
    double temp = -6.2e-6 * tut1 * tut1 * tut1 + 0.093104 * tut1 * tut1 +
                  (876600.0 * 3600.0 + 8640184.812866) * tut1 + 67310.54841;
    temp = fmod(temp * (AZ_PI / 180.0) / 240.0, AZ_TWOPI);
    if (temp < 0.0) temp += AZ_TWOPI;
    return temp;
}

AZ_DEVICE double az_pmod(double x, double m)
{
    double r = fmod(x, m);
    if (r < 0.0) r += m;
    return r;
}

// raw: TLE-unit inputs (AzRawField order).  el/n_pad/i: destination column.  returns the flag word.
AZ_DEVICE unsigned az_init_satellite(const double raw[AZ_NUM_RAW], const AzGrav &g, double *__restrict__ el,
                                     size_t n_pad, size_t i)
{
#define S(f, val) el[(size_t)F_##f * n_pad + i] = (val)
    const double deg2rad = AZ_PI / 180.0;
    const double epoch_jd = raw[R_epoch_jd];
    const double no_kozai = raw[R_mm_revday] * AZ_TWOPI / 1440.0;
    const double ecco = raw[R_ecc];
    const double inclo = raw[R_incl_deg] * deg2rad;
    const double nodeo = raw[R_raan_deg] * deg2rad;
    const double argpo = raw[R_argp_deg] * deg2rad;
    const double mo = raw[R_ma_deg] * deg2rad;
    const double bstar = raw[R_bstar];
    S(epoch_jd, epoch_jd); S(no_kozai, no_kozai); S(ecco, ecco); S(inclo, inclo); S(nodeo, nodeo);
    S(argpo, argpo); S(mo, mo); S(bstar, bstar);

    unsigned flags = 0;
    if (!(ecco >= 0.0 && ecco < 1.0)) flags = 1; // InvalidEccentricity

    // un-Kozai the mean motion
    const double cosio = cos(inclo), sinio = sin(inclo);
    const double cosio2 = cosio * cosio, cosio4 = cosio2 * cosio2;
    const double x3thm1 = 3.0 * cosio2 - 1.0;
    const double eosq = ecco * ecco;
    const double omeosq = 1.0 - eosq;
    const double rteosq = sqrt(omeosq);
    double no_unkozai, a;
    {
        const double a1 = pow(g.xke / no_kozai, 2.0 / 3.0);
        const double del1 = 0.75 * g.j2 * x3thm1 / (a1 * a1 * rteosq * omeosq);
        const double ao = a1 * (1.0 - del1 * (1.0 / 3.0 + del1 * (1.0 + 134.0 / 81.0 * del1)));
        const double delo = 0.75 * g.j2 * x3thm1 / (ao * ao * rteosq * omeosq);
        no_unkozai = no_kozai / (1.0 + delo);
        a = pow(g.xke / no_unkozai, 2.0 / 3.0);
    }
    if (flags == 0 && a * (1.0 - ecco) < 1.0) flags = 6; // SatelliteDecayed
    const bool deep = (AZ_TWOPI / no_unkozai) > 225.0;
    S(no_unkozai, no_unkozai); S(a, a); S(sinio, sinio); S(cosio, cosio);

    const double con41 = x3thm1, con42 = 1.0 - 5.0 * cosio2;
    const double x1mth2 = 1.0 - cosio2, x7thm1 = 7.0 * cosio2 - 1.0;
    S(con41, con41); S(x1mth2, x1mth2); S(x7thm1, x7thm1);

    // secular rates
    const double pinvsq = 1.0 / ((a * omeosq) * (a * omeosq));
    const double temp1 = 1.5 * g.j2 * pinvsq * no_unkozai;
    const double temp2 = 0.5 * temp1 * g.j2 * pinvsq;
    const double temp3 = -0.46875 * g.j4 * pinvsq * pinvsq * no_unkozai;
    const double mdot = no_unkozai + 0.5 * temp1 * rteosq * con41 +
                        0.0625 * temp2 * rteosq * (13.0 - 78.0 * cosio2 + 137.0 * cosio4);
    const double argpdot = -0.5 * temp1 * con42 + 0.0625 * temp2 * (7.0 - 114.0 * cosio2 + 395.0 * cosio4) +
                           temp3 * (3.0 - 36.0 * cosio2 + 49.0 * cosio4);
    const double xhdot1 = -temp1 * cosio;
    const double nodedot =
        xhdot1 + (0.5 * temp2 * (4.0 - 19.0 * cosio2) + 2.0 * temp3 * (3.0 - 7.0 * cosio2)) * cosio;
    S(mdot, mdot); S(argpdot, argpdot); S(nodedot, nodedot);

    // drag coefficients
    const double perige = (a * (1.0 - ecco) - 1.0) * g.radius_km;
    double sp = 78.0;
    if (perige < 156.0) sp = (perige < 98.0) ? 20.0 : perige - 78.0;
    const double qtemp = (120.0 - sp) / g.radius_km;
    const double sfour = sp / g.radius_km + 1.0;
    const double qzms24 = qtemp * qtemp * qtemp * qtemp;
    const double tsi = 1.0 / (a - sfour);
    const double eta = a * ecco * tsi;
    const double etasq = eta * eta, eeta = ecco * eta;
    const double psisq = fabs(1.0 - etasq);
    const double tsi2 = tsi * tsi;
    const double coef = qzms24 * (tsi2 * tsi2);
    const double coef1 = coef / pow(psisq, 3.5);
    const double cc2 = coef1 * no_unkozai *
                       (a * (1.0 + 1.5 * etasq + eeta * (4.0 + etasq)) +
                        0.375 * g.j2 * tsi / psisq * con41 * (8.0 + 3.0 * etasq * (8.0 + etasq)));
    const double cc1 = bstar * cc2;
    const double cc3 = (ecco > 1.0e-4) ? -2.0 * coef * tsi * g.j3oj2 * no_unkozai * sinio / ecco : 0.0;
    const double cc4 = 2.0 * no_unkozai * coef1 * a * omeosq *
                       (eta * (2.0 + 0.5 * etasq) + ecco * (0.5 + 2.0 * etasq) -
                        g.j2 * tsi / (a * psisq) *
                            (-3.0 * con41 * (1.0 - 2.0 * eeta + etasq * (1.5 - 0.5 * eeta)) +
                             0.75 * x1mth2 * (2.0 * etasq - eeta * (1.0 + etasq)) * cos(2.0 * argpo)));
    const double cc5 = 2.0 * coef1 * a * omeosq * (1.0 + 2.75 * (etasq + eeta) + eeta * etasq);
    S(cc1, cc1); S(bc4, bstar * cc4); S(bc5, bstar * cc5);
    S(xnodcf, 3.5 * omeosq * xhdot1 * cc1);
    S(t2cof, 1.5 * cc1);
    {
        const double num = -0.25 * g.j3oj2 * sinio * (3.0 + 5.0 * cosio);
        const double den = (fabs(cosio + 1.0) > 1.5e-12) ? 1.0 + cosio : 1.5e-12;
        S(xlcof, num / den);
    }
    S(aycof, -0.5 * g.j3oj2 * sinio);
    {
        const double dt = 1.0 + eta * cos(mo);
        S(delmo, dt * dt * dt);
    }
    S(sinmao, sin(mo));
    S(xmcof, (ecco > 1.0e-4) ? -(2.0 / 3.0) * coef * bstar / eeta : 0.0);
    S(omgcof, bstar * cc3 * cos(argpo));
    S(eta, eta);
    {
        const double ratio = g.xke / no_unkozai;
        S(a_base, cbrt(ratio * ratio));
        S(sqrt_a_base, cbrt(ratio)); // sqrt(am) = sqrt_a_base * |tempa|: the step needs no square root of am
    }

    // higher-order drag (near-earth with perigee >= 220 km only)
    const bool isimp = deep || perige < 220.0;
    if (isimp) {
        S(d2, 0.0); S(d3, 0.0); S(d4, 0.0); S(t3cof, 0.0); S(t4cof, 0.0); S(t5cof, 0.0);
    } else {
        const double s1 = 78.0 / g.radius_km + 1.0;
        const double ts = 1.0 / (a - s1);
        const double cc1sq = cc1 * cc1;
        const double d2 = 4.0 * a * ts * cc1sq;
        const double tmp = d2 * ts * cc1 / 3.0;
        const double d3 = (17.0 * a + s1) * tmp;
        const double d4 = 0.5 * tmp * a * ts * (221.0 * a + 31.0 * s1) * cc1;
        S(d2, d2); S(d3, d3); S(d4, d4);
        S(t3cof, d2 + 2.0 * cc1sq);
        S(t4cof, 0.25 * (3.0 * d3 + cc1 * (12.0 * d2 + 10.0 * cc1sq)));
        S(t5cof, 0.2 * (3.0 * d4 + 12.0 * cc1 * d3 + 6.0 * d2 * d2 + 15.0 * cc1sq * (2.0 * d2 + cc1sq)));
    }
    if (isimp) flags |= AZ_FLAG_ISIMP;
    flags |= (ecco < 0.0025 ? 0u : (ecco < 0.0075 ? 1u : (ecco < 0.1 ? 2u : 3u))) << 12;

    // ---------------------------------------------------------------- deep space
    unsigned irez = 0;
    double ls[2][12]; // periodic coefficients: [0] solar, [1] lunar
    double zmol = 0, zmos = 0, dedt = 0, didt = 0, dmdt = 0, domdt = 0, dnodt = 0;
    double d2201 = 0, d2211 = 0, d3210 = 0, d3222 = 0, d4410 = 0, d4422 = 0, d5220 = 0, d5232 = 0,
           d5421 = 0, d5433 = 0, del1 = 0, del2 = 0, del3 = 0, xlamo = 0, xfact = 0, gsto = 0;
    for (int p = 0; p < 2; ++p)
        for (int k = 0; k < 12; ++k) ls[p][k] = 0.0;

    if (deep && AZ_FLAG_ERR(flags) == 0) {
        flags |= AZ_FLAG_DEEP;
        gsto = az_gstime(epoch_jd);
        const double day = epoch_jd - 2415020.0;
        const double nm = no_unkozai;
        const double snodm = sin(nodeo), cnodm = cos(nodeo);
        const double sinomm = sin(argpo), cosomm = cos(argpo);
        const double sinim = sinio, cosim = cosio;
        const double emsq = eosq;
        const double rtemsq = rteosq;

        const double xnodce = az_pmod(4.5236020 - 9.2422029e-4 * day, AZ_TWOPI);
        const double stem = sin(xnodce), ctem = cos(xnodce);
        const double zcosil = 0.91375164 - 0.03568096 * ctem;
        const double zsinil = sqrt(1.0 - zcosil * zcosil);
        const double zsinhl = 0.089683511 * stem / zsinil;
        const double zcoshl = sqrt(1.0 - zsinhl * zsinhl);
        const double gam = 5.8351514 + 0.0019443680 * day;
        double zx = 0.39785416 * stem / zsinil;
        const double zy = zcoshl * ctem + 0.91744867 * zsinhl * stem;
        zx = atan2(zx, zy);
        zx += gam - xnodce;
        const double zcosgl = cos(zx), zsingl = sin(zx);
        const double xnoi = 1.0 / nm;
        const double betasq = 1.0 - emsq;

        double zcosg = 0.1945905, zsing = -0.98088458, zcosi = 0.91744867, zsini = 0.39785416;
        double zcosh = cnodm, zsinh = snodm, cc = 2.9864797e-6;
        // accumulated secular terms per body
        double b_s1[2], b_s2[2], b_s3[2], b_s4[2], b_s5[2];
        double b_z1[2], b_z3[2], b_z11[2], b_z13[2], b_z21[2], b_z23[2], b_z31[2], b_z33[2];
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
            const double a1 = zcosg * zcosh + zsing * zcosi * zsinh;
            const double a3 = -zsing * zcosh + zcosg * zcosi * zsinh;
            const double a7 = -zcosg * zsinh + zsing * zcosi * zcosh;
            const double a8 = zsing * zsini;
            const double a9 = zsing * zsinh + zcosg * zcosi * zcosh;
            const double a10 = zcosg * zsini;
            const double a2 = cosim * a7 + sinim * a8;
            const double a4 = cosim * a9 + sinim * a10;
            const double a5 = -sinim * a7 + cosim * a8;
            const double a6 = -sinim * a9 + cosim * a10;
            const double x1 = a1 * cosomm + a2 * sinomm;
            const double x2 = a3 * cosomm + a4 * sinomm;
            const double x3 = -a1 * sinomm + a2 * cosomm;
            const double x4 = -a3 * sinomm + a4 * cosomm;
            const double x5 = a5 * sinomm, x6 = a6 * sinomm, x7 = a5 * cosomm, x8 = a6 * cosomm;

            const double z31 = 12.0 * x1 * x1 - 3.0 * x3 * x3;
            const double z32 = 24.0 * x1 * x2 - 6.0 * x3 * x4;
            const double z33 = 12.0 * x2 * x2 - 3.0 * x4 * x4;
            const double z1v = 3.0 * (a1 * a1 + a2 * a2) + z31 * emsq;
            const double z2v = 6.0 * (a1 * a3 + a2 * a4) + z32 * emsq;
            const double z3v = 3.0 * (a3 * a3 + a4 * a4) + z33 * emsq;
            const double z11 = -6.0 * a1 * a5 + emsq * (-24.0 * x1 * x7 - 6.0 * x3 * x5);
            const double z12 =
                -6.0 * (a1 * a6 + a3 * a5) + emsq * (-24.0 * (x2 * x7 + x1 * x8) - 6.0 * (x3 * x6 + x4 * x5));
            const double z13 = -6.0 * a3 * a6 + emsq * (-24.0 * x2 * x8 - 6.0 * x4 * x6);
            const double z21 = 6.0 * a2 * a5 + emsq * (24.0 * x1 * x5 - 6.0 * x3 * x7);
            const double z22 =
                6.0 * (a4 * a5 + a2 * a6) + emsq * (24.0 * (x2 * x5 + x1 * x6) - 6.0 * (x4 * x7 + x3 * x8));
            const double z23 = 6.0 * a4 * a6 + emsq * (24.0 * x2 * x6 - 6.0 * x4 * x8);
            const double z1 = z1v + z1v + betasq * z31;
            const double z2 = z2v + z2v + betasq * z32;
            const double z3 = z3v + z3v + betasq * z33;
            const double s3 = cc * xnoi;
            const double s2 = -0.5 * s3 / rtemsq;
            const double s4 = s3 * rtemsq;
            const double s1 = -15.0 * ecco * s4;
            const double s5 = x1 * x3 + x2 * x4;
            const double s6 = x2 * x3 + x1 * x4;
            const double s7 = x2 * x4 - x1 * x3;
            const double ze = (pass == 0) ? 0.01675 : 0.05490;

            double *c = ls[pass];
            c[0] = 2.0 * s1 * s6;                         // e2
            c[1] = 2.0 * s1 * s7;                         // e3
            c[2] = 2.0 * s2 * z12;                        // i2
            c[3] = 2.0 * s2 * (z13 - z11);                // i3
            c[4] = -2.0 * s3 * z2;                        // l2
            c[5] = -2.0 * s3 * (z3 - z1);                 // l3
            c[6] = -2.0 * s3 * (-21.0 - 9.0 * emsq) * ze; // l4
            c[7] = 2.0 * s4 * z32;                        // gh2
            c[8] = 2.0 * s4 * (z33 - z31);                // gh3
            c[9] = -18.0 * s4 * ze;                       // gh4
            c[10] = -2.0 * s2 * z22;                      // h2
            c[11] = -2.0 * s2 * (z23 - z21);              // h3

            b_s1[pass] = s1; b_s2[pass] = s2; b_s3[pass] = s3; b_s4[pass] = s4; b_s5[pass] = s5;
            b_z1[pass] = z1; b_z3[pass] = z3; b_z11[pass] = z11; b_z13[pass] = z13;
            b_z21[pass] = z21; b_z23[pass] = z23; b_z31[pass] = z31; b_z33[pass] = z33;

            if (pass == 0) {
                zcosg = zcosgl; zsing = zsingl; zcosi = zcosil; zsini = zsinil;
                zcosh = zcoshl * cnodm + zsinhl * snodm;
                zsinh = snodm * zcoshl - cnodm * zsinhl;
                cc = 4.7968065e-7;
            }
        }
        zmol = az_pmod(4.7199672 + 0.22997150 * day - gam, AZ_TWOPI);
        zmos = az_pmod(6.2565837 + 0.017201977 * day, AZ_TWOPI);

        // dsinit: secular rates
        const double zns = 1.19459e-5, znl = 1.5835218e-4, rptim = 4.37526908801129966e-3;
        const bool near_eq = (inclo < 5.2359877e-2) || (inclo > AZ_PI - 5.2359877e-2);
        const double ses = b_s1[0] * zns * b_s5[0];
        const double sis = b_s2[0] * zns * (b_z11[0] + b_z13[0]);
        const double sls = -zns * b_s3[0] * (b_z1[0] + b_z3[0] - 14.0 - 6.0 * emsq);
        const double sghs = b_s4[0] * zns * (b_z31[0] + b_z33[0] - 6.0);
        double shs = -zns * b_s2[0] * (b_z21[0] + b_z23[0]);
        if (near_eq) shs = 0.0;
        if (sinim != 0.0) shs = shs / sinim;
        const double sgs = sghs - cosim * shs;
        dedt = ses + b_s1[1] * znl * b_s5[1];
        didt = sis + b_s2[1] * znl * (b_z11[1] + b_z13[1]);
        dmdt = sls - znl * b_s3[1] * (b_z1[1] + b_z3[1] - 14.0 - 6.0 * emsq);
        const double sghl = b_s4[1] * znl * (b_z31[1] + b_z33[1] - 6.0);
        double shll = -znl * b_s2[1] * (b_z21[1] + b_z23[1]);
        if (near_eq) shll = 0.0;
        domdt = sgs + sghl;
        dnodt = shs;
        if (sinim != 0.0) {
            domdt -= cosim / sinim * shll;
            dnodt += shll / sinim;
        }

        if (nm >= 0.00826 && nm <= 0.00924 && ecco >= 0.5)
            irez = 2;
        else if (nm >= 0.0034906585 && nm <= 0.0052359877)
            irez = 1;

        const double sini2 = sinio * sinio, cosisq = cosio2;
        const double xpidot = argpdot + nodedot;
        const double aonv = 1.0 / a;
        if (irez == 1) {
            const double g200 = 1.0 + eosq * (-2.5 + 0.8125 * eosq);
            const double g310 = 1.0 + 2.0 * eosq;
            const double g300 = 1.0 + eosq * (-6.0 + 6.60937 * eosq);
            const double f220 = 0.75 * (1.0 + cosio) * (1.0 + cosio);
            const double f311 = 0.9375 * sini2 * (1.0 + 3.0 * cosio) - 0.75 * (1.0 + cosio);
            double f330 = 1.0 + cosio;
            f330 = 1.875 * f330 * f330 * f330;
            const double t1 = 3.0 * nm * nm * aonv * aonv;
            del2 = 2.0 * t1 * f220 * g200 * 1.7891679e-6;
            del3 = 3.0 * t1 * f330 * g300 * 2.2123015e-7 * aonv;
            del1 = t1 * f311 * g310 * 2.1460748e-6 * aonv;
            xlamo = az_pmod(mo + nodeo + argpo - gsto, AZ_TWOPI);
            xfact = mdot + xpidot - rptim + dmdt + domdt + dnodt - no_unkozai;
        } else if (irez == 2) {
            const double e = ecco;
            const bool lo = e <= 0.65;
            const double g201 = -0.306 - (e - 0.64) * 0.440;
            const double g211 = lo ? az_poly3(e, 3.616, -13.2470, 16.2900, 0.0)
                                   : az_poly3(e, -72.099, 331.819, -508.738, 266.724);
            const double g310 = lo ? az_poly3(e, -19.302, 117.3900, -228.4190, 156.591)
                                   : az_poly3(e, -346.844, 1582.851, -2415.925, 1246.113);
            const double g322 = lo ? az_poly3(e, -18.9068, 109.7927, -214.6334, 146.5816)
                                   : az_poly3(e, -342.585, 1554.908, -2366.899, 1215.972);
            const double g410 = lo ? az_poly3(e, -41.122, 242.6940, -471.0940, 313.953)
                                   : az_poly3(e, -1052.797, 4758.686, -7193.992, 3651.957);
            const double g422 = lo ? az_poly3(e, -146.407, 841.8800, -1629.014, 1083.435)
                                   : az_poly3(e, -3581.690, 16178.110, -24462.770, 12422.520);
            double g520;
            if (lo)
                g520 = az_poly3(e, -532.114, 3017.977, -5740.032, 3708.276);
            else if (e > 0.715)
                g520 = az_poly3(e, -5149.66, 29936.92, -54087.36, 31324.56);
            else
                g520 = 1464.74 - 4664.75 * e + 3763.64 * e * e;
            const bool lo7 = e < 0.7;
            const double g521 = lo7 ? az_poly3(e, -822.71072, 4568.6173, -8491.4146, 5337.524)
                                    : az_poly3(e, -51752.104, 218913.95, -309468.16, 146349.42);
            const double g532 = lo7 ? az_poly3(e, -853.66600, 4690.2500, -8624.7700, 5341.400)
                                    : az_poly3(e, -40023.880, 170470.89, -242699.48, 115605.82);
            const double g533 = lo7 ? az_poly3(e, -919.22770, 4988.6100, -9064.7700, 5542.21)
                                    : az_poly3(e, -37995.780, 161616.52, -229838.20, 109377.94);
            const double c = cosio, si = sinio;
            const double f220 = 0.75 * (1.0 + 2.0 * c + cosisq);
            const double f221 = 1.5 * sini2;
            const double f321 = 1.875 * si * (1.0 - 2.0 * c - 3.0 * cosisq);
            const double f322 = -1.875 * si * (1.0 + 2.0 * c - 3.0 * cosisq);
            const double f441 = 35.0 * sini2 * f220;
            const double f442 = 39.3750 * sini2 * sini2;
            const double f522 = 9.84375 * si *
                                (sini2 * (1.0 - 2.0 * c - 5.0 * cosisq) +
                                 0.33333333 * (-2.0 + 4.0 * c + 6.0 * cosisq));
            const double f523 = si * (4.92187512 * sini2 * (-2.0 - 4.0 * c + 10.0 * cosisq) +
                                      6.56250012 * (1.0 + 2.0 * c - 3.0 * cosisq));
            const double f542 = 29.53125 * si * (2.0 - 8.0 * c + cosisq * (-12.0 + 8.0 * c + 10.0 * cosisq));
            const double f543 = 29.53125 * si * (-2.0 - 8.0 * c + cosisq * (12.0 + 8.0 * c - 10.0 * cosisq));
            double t1 = 3.0 * nm * nm * aonv * aonv;
            double t = t1 * 1.7891679e-6;
            d2201 = t * f220 * g201;
            d2211 = t * f221 * g211;
            t1 *= aonv;
            t = t1 * 3.7393792e-7;
            d3210 = t * f321 * g310;
            d3222 = t * f322 * g322;
            t1 *= aonv;
            t = 2.0 * t1 * 7.3636953e-9;
            d4410 = t * f441 * g410;
            d4422 = t * f442 * g422;
            t1 *= aonv;
            t = t1 * 1.1428639e-7;
            d5220 = t * f522 * g520;
            d5232 = t * f523 * g532;
            t = 2.0 * t1 * 2.1765803e-9;
            d5421 = t * f542 * g521;
            d5433 = t * f543 * g533;
            xlamo = az_pmod(mo + nodeo + nodeo - gsto - gsto, AZ_TWOPI);
            xfact = mdot + dmdt + 2.0 * (nodedot + dnodt - rptim) - no_unkozai;
        }
        flags |= irez << 10;
    }
    S(se2, ls[0][0]); S(se3, ls[0][1]); S(si2, ls[0][2]); S(si3, ls[0][3]); S(sl2, ls[0][4]);
    S(sl3, ls[0][5]); S(sl4, ls[0][6]); S(sgh2, ls[0][7]); S(sgh3, ls[0][8]); S(sgh4, ls[0][9]);
    S(sh2, ls[0][10]); S(sh3, ls[0][11]);
    S(ee2, ls[1][0]); S(e3, ls[1][1]); S(xi2, ls[1][2]); S(xi3, ls[1][3]); S(xl2, ls[1][4]);
    S(xl3, ls[1][5]); S(xl4, ls[1][6]); S(xgh2, ls[1][7]); S(xgh3, ls[1][8]); S(xgh4, ls[1][9]);
    S(xh2, ls[1][10]); S(xh3, ls[1][11]);
    S(zmol, zmol); S(zmos, zmos); S(dedt, dedt); S(didt, didt); S(dmdt, dmdt); S(domdt, domdt);
    S(dnodt, dnodt);
    S(d2201, d2201); S(d2211, d2211); S(d3210, d3210); S(d3222, d3222); S(d4410, d4410);
    S(d4422, d4422); S(d5220, d5220); S(d5232, d5232); S(d5421, d5421); S(d5433, d5433);
    S(del1, del1); S(del2, del2); S(del3, del3); S(xlamo, xlamo); S(xfact, xfact); S(gsto, gsto);
#undef S
    return flags;
}
