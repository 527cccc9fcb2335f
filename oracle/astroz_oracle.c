/*
 * astroz_oracle.c -- CPU ORACLE (test infrastructure only; see astroz_oracle.h).
 *
 * Scalar fp64 restatement of the reference's scalar SGP4/SDP4 path.  Each function cites the
 * reference file:line whose algorithm it follows.  libm sin/cos/atan2/fmod/pow/cbrt, no SIMD.
 */
#include "astroz_oracle.h"

#include <ctype.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

#define PI 3.14159265358979323846
#define TWOPI (2.0 * PI)
#define DEG2RAD (PI / 180.0)

/* Zig's @mod on floats: result has the sign of the divisor (here always > 0). */
static double pmod(double x, double m)
{
    double r = fmod(x, m);
    if (r < 0.0) r += m;
    return r;
}

size_t orc_sizeof_tle(void) { return sizeof(orc_tle); }
size_t orc_sizeof_sat(void) { return sizeof(orc_sat); }

/* ------------------------------------------------------------------ gravity (constants.zig L41-64) */
void orc_get_grav(int which, orc_grav *g)
{
    if (which == ORC_WGS72) {
        g->radius_km = 6378.135;
        g->mu = 398600.8;
        g->j2 = 0.001082616;
        g->j3 = -0.00000253881;
        g->j4 = -0.00000165597;
        g->xke = 0.0743669161331734132;
        g->tumin = 13.44683969695931;
        g->j3oj2 = -0.00234506972242078;
    } else {
        g->radius_km = 6378.137;
        g->mu = 398600.5;
        g->j2 = 0.00108262998905;
        g->j3 = -0.00000253215306;
        g->j4 = -0.00000161098761;
        g->xke = 0.07436685316871385;
        g->tumin = 13.446851082044981;
        g->j3oj2 = -0.00233899967218727;
    }
}

/* ------------------------------------------------------------------ TLE text (Tle.zig L49-101, L277-304) */

/* copy columns [a,b) with surrounding blanks removed; returns length */
static size_t field(const char *line, int a, int b, char *buf)
{
    while (a < b && line[a] == ' ') a++;
    while (b > a && line[b - 1] == ' ') b--;
    size_t n = (size_t)(b - a);
    memcpy(buf, line + a, n);
    buf[n] = 0;
    return n;
}

static int parse_double(const char *s, double *out)
{
    if (!*s) return 0;
    char *end = NULL;
    double v = strtod(s, &end);
    if (end == s || *end != 0) return 0;
    *out = v;
    return 1;
}

static int parse_long(const char *s, long *out)
{
    if (!*s) return 0;
    char *end = NULL;
    long v = strtol(s, &end, 10);
    if (end == s || *end != 0) return 0;
    *out = v;
    return 1;
}

/* Datetime.zig L222-231 */
double orc_year_doy_to_jd(int full_year, double doy)
{
    double y = (double)full_year;
    double a = floor((14.0 - 1.0) / 12.0);
    double yy = y + 4800.0 - a;
    double mm = 1.0 + 12.0 * a - 3.0;
    double jd_jan1 = 1.0 + floor((153.0 * mm + 2.0) / 5.0) + 365.0 * yy + floor(yy / 4.0) -
                     floor(yy / 100.0) + floor(yy / 400.0) - 32045.0;
    return jd_jan1 + doy - 1.5;
}

static size_t trimmed_len(const char *s, size_t n)
{
    while (n > 0 && (s[n - 1] == ' ' || s[n - 1] == '\t' || s[n - 1] == '\r' || s[n - 1] == '\n')) n--;
    return n;
}

int orc_tle_parse_lines(const char *line1, const char *line2, orc_tle *t)
{
    char b[32];
    double d;
    long l;
    if (strlen(line1) < 69 || strlen(line2) < 69) return ORC_ERR_BAD_TLE;
    memset(t, 0, sizeof(*t));

    /* satellite number, alpha-5 aware (Tle.zig L281-290) */
    field(line1, 2, 7, b);
    if (!b[0]) return ORC_ERR_BAD_TLE;
    if (b[0] >= 'A' && b[0] <= 'Z') {
        if (!parse_long(b + 1, &l)) return ORC_ERR_BAD_TLE;
        t->satnum = (uint32_t)((b[0] - 'A') + 10) * 10000u + (uint32_t)l;
    } else {
        if (!parse_long(b, &l)) return ORC_ERR_BAD_TLE;
        t->satnum = (uint32_t)l;
    }
    t->classification = line1[7];

    /* B*: mantissa * 1e-5 * 10^exp (Tle.zig L69-71) */
    field(line1, 53, 59, b);
    if (!parse_double(b, &d)) return ORC_ERR_BAD_TLE;
    field(line1, 59, 61, b);
    if (!parse_long(b, &l)) return ORC_ERR_BAD_TLE;
    t->bstar = (d * 1e-5) * pow(10.0, (double)l);

    field(line1, 18, 20, b);
    if (!parse_long(b, &l)) return ORC_ERR_BAD_TLE;
    t->epoch_year = (int)l;
    field(line1, 20, 32, b);
    if (!parse_double(b, &t->epoch_day)) return ORC_ERR_BAD_TLE;
    /* yy < 57 -> 20yy (Tle.zig L298-304) */
    t->epoch_jd = orc_year_doy_to_jd(t->epoch_year < 57 ? 2000 + t->epoch_year : 1900 + t->epoch_year,
                                     t->epoch_day);

    field(line1, 33, 43, b);
    if (!parse_double(b, &t->ndot)) return ORC_ERR_BAD_TLE;
    field(line1, 64, 68, b);
    if (!parse_long(b, &l)) return ORC_ERR_BAD_TLE;
    t->elnum = (uint32_t)l;

    field(line2, 8, 16, b);
    if (!parse_double(b, &t->incl_deg)) return ORC_ERR_BAD_TLE;
    field(line2, 17, 25, b);
    if (!parse_double(b, &t->raan_deg)) return ORC_ERR_BAD_TLE;
    field(line2, 26, 33, b);
    if (!parse_double(b, &d)) return ORC_ERR_BAD_TLE;
    t->ecc = d / 1e7; /* Tle.zig L78 */
    field(line2, 34, 42, b);
    if (!parse_double(b, &t->argp_deg)) return ORC_ERR_BAD_TLE;
    field(line2, 43, 51, b);
    if (!parse_double(b, &t->ma_deg)) return ORC_ERR_BAD_TLE;
    field(line2, 52, 63, b);
    if (!parse_double(b, &t->mm_revday)) return ORC_ERR_BAD_TLE;
    field(line2, 63, 68, b);
    if (!parse_long(b, &l)) return ORC_ERR_BAD_TLE;
    t->revnum = (uint32_t)l;
    return ORC_OK;
}

/* iterate text lines, trimmed of blanks/tabs, keeping those with >= 69 chars */
static const char *next_long_line(const char *p, char *buf, size_t bufsz, int *found)
{
    *found = 0;
    while (*p) {
        const char *e = p;
        while (*e && *e != '\n' && *e != '\r') e++;
        const char *s = p;
        while (s < e && (*s == ' ' || *s == '\t')) s++;
        size_t n = trimmed_len(s, (size_t)(e - s));
        p = (*e) ? e + 1 : e;
        if (n >= 69) {
            if (n >= bufsz) n = bufsz - 1;
            memcpy(buf, s, n);
            buf[n] = 0;
            *found = 1;
            return p;
        }
    }
    return p;
}

int orc_tle_parse(const char *text, orc_tle *out)
{
    char l1[256], l2[256];
    int f;
    const char *p = next_long_line(text, l1, sizeof l1, &f);
    if (!f) return ORC_ERR_BAD_TLE;
    next_long_line(p, l2, sizeof l2, &f);
    if (!f) return ORC_ERR_BAD_TLE;
    return orc_tle_parse_lines(l1, l2, out);
}

size_t orc_tle_parse_multi(const char *text, orc_tle *out, size_t max)
{
    char cand[256], line[256];
    int have = 0, f;
    size_t n = 0;
    const char *p = text;
    for (;;) {
        p = next_long_line(p, line, sizeof line, &f);
        if (!f) break;
        if (line[0] == '1') {
            memcpy(cand, line, sizeof cand);
            have = 1;
        } else if (line[0] == '2') {
            if (have) {
                have = 0;
                if (n < max && orc_tle_parse_lines(cand, line, &out[n]) == ORC_OK) n++;
            }
        } else {
            have = 0;
        }
    }
    return n;
}

/* ------------------------------------------------------------------ SGP4 init (Sgp4.zig L108-417) */

/* drag model constants, Sgp4.zig L13-24 */
#define PERIGEE_S_ADJUST 156.0
#define PERIGEE_S_MIN 98.0
#define S_STANDARD 78.0
#define Q_PARAM 120.0
#define S_MIN 20.0
#define PERIGEE_SIMPLIFIED 220.0
#define ECC_MIN 1.0e-4
#define ECC_FLOOR 1.0e-6
#define SINGULARITY_TOL 1.5e-12
#define DEEP_SPACE_MINUTES 225.0

/* common part of Sgp4.initElements L108-180 and Sdp4.initElements L174-243.
 * returns ORC_OK or an error; *period_min receives 2pi/no_unkozai. */
static int sgp4_common_init(const orc_tle *tle, int grav, orc_sat *s, double *perige_out)
{
    memset(s, 0, sizeof(*s));
    orc_get_grav(grav, &s->g);
    const orc_grav *g = &s->g;
    s->epoch_jd = tle->epoch_jd;

    /* extractMeanElements L192-202 */
    s->no_kozai = tle->mm_revday * TWOPI / 1440.0;
    s->ecco = tle->ecc;
    s->inclo = tle->incl_deg * DEG2RAD;
    s->nodeo = tle->raan_deg * DEG2RAD;
    s->argpo = tle->argp_deg * DEG2RAD;
    s->mo = tle->ma_deg * DEG2RAD;
    s->bstar = tle->bstar;

    if (s->ecco < 0.0 || s->ecco >= 1.0) return ORC_ERR_ECCENTRICITY;

    /* recoverMeanMotion L206-228 */
    {
        double cosio = cos(s->inclo);
        double theta2 = cosio * cosio;
        double x3thm1 = 3.0 * theta2 - 1.0;
        double eosq = s->ecco * s->ecco;
        double betao2 = 1.0 - eosq;
        double betao = sqrt(betao2);
        double a1 = pow(g->xke / s->no_kozai, 2.0 / 3.0);
        double del1 = 0.75 * g->j2 * x3thm1 / (a1 * a1 * betao * betao2);
        double ao = a1 * (1.0 - del1 * (1.0 / 3.0 + del1 * (1.0 + 134.0 / 81.0 * del1)));
        double delo = 0.75 * g->j2 * x3thm1 / (ao * ao * betao * betao2);
        s->no_unkozai = s->no_kozai / (1.0 + delo);
        s->a = pow(g->xke / s->no_unkozai, 2.0 / 3.0);
    }
    if (s->a * (1.0 - s->ecco) < 1.0) return ORC_ERR_DECAYED;

    /* computeTrigTerms L232-237, computePolyTerms L241-249 */
    s->sinio = sin(s->inclo);
    s->cosio = cos(s->inclo);
    s->cosio2 = s->cosio * s->cosio;
    s->cosio4 = s->cosio2 * s->cosio2;
    s->con41 = 3.0 * s->cosio2 - 1.0;
    s->con42 = 1.0 - 5.0 * s->cosio2;
    s->x1mth2 = 1.0 - s->cosio2;
    s->x7thm1 = 7.0 * s->cosio2 - 1.0;

    /* computeSecularRates L253-284 */
    double omeosq = 1.0 - s->ecco * s->ecco;
    double rteosq = sqrt(omeosq);
    double pinvsq = 1.0 / pow(s->a * omeosq, 2.0);
    double temp1 = 1.5 * g->j2 * pinvsq * s->no_unkozai;
    double temp2 = 0.5 * temp1 * g->j2 * pinvsq;
    double temp3 = -0.46875 * g->j4 * pinvsq * pinvsq * s->no_unkozai;
    s->mdot = s->no_unkozai + 0.5 * temp1 * rteosq * s->con41 +
              0.0625 * temp2 * rteosq * (13.0 - 78.0 * s->cosio2 + 137.0 * s->cosio4);
    s->argpdot = -0.5 * temp1 * s->con42 +
                 0.0625 * temp2 * (7.0 - 114.0 * s->cosio2 + 395.0 * s->cosio4) +
                 temp3 * (3.0 - 36.0 * s->cosio2 + 49.0 * s->cosio4);
    double xhdot1 = -temp1 * s->cosio;
    s->nodedot = xhdot1 + (0.5 * temp2 * (4.0 - 19.0 * s->cosio2) + 2.0 * temp3 * (3.0 - 7.0 * s->cosio2)) * s->cosio;

    /* computeDragCoefficients L301-382 */
    double perige = (s->a * (1.0 - s->ecco) - 1.0) * g->radius_km;
    *perige_out = perige;
    double sp = S_STANDARD;
    if (perige < PERIGEE_S_ADJUST) sp = (perige < PERIGEE_S_MIN) ? S_MIN : perige - S_STANDARD;
    double qtemp = (Q_PARAM - sp) / g->radius_km;
    double sfour = sp / g->radius_km + 1.0;
    double qzms24 = qtemp * qtemp * qtemp * qtemp;
    double tsi = 1.0 / (s->a - sfour);
    double eta = s->a * s->ecco * tsi;
    double etasq = eta * eta;
    double eeta = s->ecco * eta;
    double psisq = fabs(1.0 - etasq);
    double coef = qzms24 * pow(tsi, 4.0);
    double coef1 = coef / pow(psisq, 3.5);
    double cc2 = coef1 * s->no_unkozai *
                 (s->a * (1.0 + 1.5 * etasq + eeta * (4.0 + etasq)) +
                  0.375 * g->j2 * tsi / psisq * s->con41 * (8.0 + 3.0 * etasq * (8.0 + etasq)));
    s->cc1 = s->bstar * cc2;
    double cc3 = 0.0;
    if (s->ecco > ECC_MIN) cc3 = -2.0 * coef * tsi * g->j3oj2 * s->no_unkozai * s->sinio / s->ecco;
    s->cc4 = 2.0 * s->no_unkozai * coef1 * s->a * omeosq *
             (eta * (2.0 + 0.5 * etasq) + s->ecco * (0.5 + 2.0 * etasq) -
              g->j2 * tsi / (s->a * psisq) *
                  (-3.0 * s->con41 * (1.0 - 2.0 * eeta + etasq * (1.5 - 0.5 * eeta)) +
                   0.75 * s->x1mth2 * (2.0 * etasq - eeta * (1.0 + etasq)) * cos(2.0 * s->argpo)));
    s->cc5 = 2.0 * coef1 * s->a * omeosq * (1.0 + 2.75 * (etasq + eeta) + eeta * etasq);
    s->xnodcf = 3.5 * omeosq * xhdot1 * s->cc1;
    s->t2cof = 1.5 * s->cc1;
    {
        double num = -0.25 * g->j3oj2 * s->sinio * (3.0 + 5.0 * s->cosio);
        double den = (fabs(s->cosio + 1.0) > SINGULARITY_TOL) ? 1.0 + s->cosio : SINGULARITY_TOL;
        s->xlcof = num / den;
    }
    s->aycof = -0.5 * g->j3oj2 * s->sinio;
    {
        double dt = 1.0 + eta * cos(s->mo);
        s->delmo = dt * dt * dt;
    }
    s->sinmao = sin(s->mo);
    s->xmcof = (s->ecco > ECC_MIN) ? -(2.0 / 3.0) * coef * s->bstar / eeta : 0.0;
    s->omgcof = s->bstar * cc3 * cos(s->argpo);
    s->eta = eta;

    {
        double ratio = g->xke / s->no_unkozai;
        s->a_base = cbrt(ratio * ratio);
    }
    s->vkmpersec = g->xke * g->radius_km / 60.0;
    s->isimp = 1;
    return ORC_OK;
}

/* computeHigherOrderDrag L394-417 */
static void sgp4_higher_order(orc_sat *s, double perige)
{
    if (perige < PERIGEE_SIMPLIFIED) {
        s->isimp = 1;
        return;
    }
    double sp = S_STANDARD / s->g.radius_km + 1.0;
    double tsi = 1.0 / (s->a - sp);
    double cc1sq = s->cc1 * s->cc1;
    s->d2 = 4.0 * s->a * tsi * cc1sq;
    double temp = s->d2 * tsi * s->cc1 / 3.0;
    s->d3 = (17.0 * s->a + sp) * temp;
    s->d4 = 0.5 * temp * s->a * tsi * (221.0 * s->a + 31.0 * sp) * s->cc1;
    s->t3cof = s->d2 + 2.0 * cc1sq;
    s->t4cof = 0.25 * (3.0 * s->d3 + s->cc1 * (12.0 * s->d2 + 10.0 * cc1sq));
    s->t5cof = 0.2 * (3.0 * s->d4 + 12.0 * s->cc1 * s->d3 + 6.0 * s->d2 * s->d2 +
                      15.0 * cc1sq * (2.0 * s->d2 + cc1sq));
    s->isimp = 0;
}

/* ------------------------------------------------------------------ SDP4 init (Sdp4.zig L15-52, L277-679) */
#define ZES 0.01675
#define ZEL 0.05490
#define C1SS 2.9864797e-6
#define C1L 4.7968065e-7
#define ZSINIS 0.39785416
#define ZCOSIS 0.91744867
#define ZCOSGS 0.1945905
#define ZSINGS (-0.98088458)
#define ZNS 1.19459e-5
#define ZNL 1.5835218e-4
#define Q22 1.7891679e-6
#define Q31 2.1460748e-6
#define Q33 2.2123015e-7
#define ROOT22 1.7891679e-6
#define ROOT32 3.7393792e-7
#define ROOT44 7.3636953e-9
#define ROOT52 1.1428639e-7
#define ROOT54 2.1765803e-9
#define RPTIM 4.37526908801129966e-3
#define FASX2 0.13130908
#define FASX4 2.8843198
#define FASX6 0.37448087
#define G22 5.7686396
#define G32 0.95240898
#define G44 1.8014998
#define G52 1.0508330
#define G54 4.4108898
#define NEAR_EQUATORIAL 5.2359877e-2
#define STEPP 720.0
#define STEP2 259200.0

/* Sdp4.zig L277-285 */
double orc_gstime(double jdut1)
{
    double tut1 = (jdut1 - 2451545.0) / 36525.0;
    double temp = -6.2e-6 * tut1 * tut1 * tut1 + 0.093104 * tut1 * tut1 +
                  (876600.0 * 3600.0 + 8640184.812866) * tut1 + 67310.54841;
    temp = pmod(temp * DEG2RAD / 240.0, TWOPI);
    if (temp < 0.0) temp += TWOPI;
    return temp;
}

typedef struct {
    double s1, s2, s3, s4, s5, s6, s7;
    double z1, z2, z3, z11, z12, z13, z21, z22, z23, z31, z32, z33;
} ls_terms;

static double poly(double x, const double *c, int n)
{
    double r = 0.0, xn = 1.0;
    for (int i = 0; i < n; i++) {
        r += c[i] * xn;
        xn *= x;
    }
    return r;
}

/* dscom L344-499 + dsinit L525-657 */
static void sdp4_deep_init(orc_sat *s)
{
    const double day = s->epoch_jd - 2415020.0;
    const double nm = s->no_unkozai;
    const double snodm = sin(s->nodeo), cnodm = cos(s->nodeo);
    const double sinomm = sin(s->argpo), cosomm = cos(s->argpo);
    const double sinim = s->sinio, cosim = s->cosio;
    const double emsq = s->ecco * s->ecco;
    const double rtemsq = sqrt(1.0 - emsq);

    /* lunar node/inclination geometry, L362-376 */
    const double xnodce = pmod(4.5236020 - 9.2422029e-4 * day, TWOPI);
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

    ls_terms T[2]; /* 0 = solar, 1 = lunar */
    double zcosg = ZCOSGS, zsing = ZSINGS, zcosi = ZCOSIS, zsini = ZSINIS;
    double zcosh = cnodm, zsinh = snodm, cc = C1SS;

    for (int pass = 0; pass < 2; pass++) {
        double a1 = zcosg * zcosh + zsing * zcosi * zsinh;
        double a3 = -zsing * zcosh + zcosg * zcosi * zsinh;
        double a7 = -zcosg * zsinh + zsing * zcosi * zcosh;
        double a8 = zsing * zsini;
        double a9 = zsing * zsinh + zcosg * zcosi * zcosh;
        double a10 = zcosg * zsini;
        double a2 = cosim * a7 + sinim * a8;
        double a4 = cosim * a9 + sinim * a10;
        double a5 = -sinim * a7 + cosim * a8;
        double a6 = -sinim * a9 + cosim * a10;

        double x1 = a1 * cosomm + a2 * sinomm;
        double x2 = a3 * cosomm + a4 * sinomm;
        double x3 = -a1 * sinomm + a2 * cosomm;
        double x4 = -a3 * sinomm + a4 * cosomm;
        double x5 = a5 * sinomm;
        double x6 = a6 * sinomm;
        double x7 = a5 * cosomm;
        double x8 = a6 * cosomm;

        ls_terms *t = &T[pass];
        t->z31 = 12.0 * x1 * x1 - 3.0 * x3 * x3;
        t->z32 = 24.0 * x1 * x2 - 6.0 * x3 * x4;
        t->z33 = 12.0 * x2 * x2 - 3.0 * x4 * x4;
        double z1v = 3.0 * (a1 * a1 + a2 * a2) + t->z31 * emsq;
        double z2v = 6.0 * (a1 * a3 + a2 * a4) + t->z32 * emsq;
        double z3v = 3.0 * (a3 * a3 + a4 * a4) + t->z33 * emsq;
        t->z11 = -6.0 * a1 * a5 + emsq * (-24.0 * x1 * x7 - 6.0 * x3 * x5);
        t->z12 = -6.0 * (a1 * a6 + a3 * a5) + emsq * (-24.0 * (x2 * x7 + x1 * x8) - 6.0 * (x3 * x6 + x4 * x5));
        t->z13 = -6.0 * a3 * a6 + emsq * (-24.0 * x2 * x8 - 6.0 * x4 * x6);
        t->z21 = 6.0 * a2 * a5 + emsq * (24.0 * x1 * x5 - 6.0 * x3 * x7);
        t->z22 = 6.0 * (a4 * a5 + a2 * a6) + emsq * (24.0 * (x2 * x5 + x1 * x6) - 6.0 * (x4 * x7 + x3 * x8));
        t->z23 = 6.0 * a4 * a6 + emsq * (24.0 * x2 * x6 - 6.0 * x4 * x8);
        t->z1 = z1v + z1v + betasq * t->z31;
        t->z2 = z2v + z2v + betasq * t->z32;
        t->z3 = z3v + z3v + betasq * t->z33;
        t->s3 = cc * xnoi;
        t->s2 = -0.5 * t->s3 / rtemsq;
        t->s4 = t->s3 * rtemsq;
        t->s1 = -15.0 * s->ecco * t->s4;
        t->s5 = x1 * x3 + x2 * x4;
        t->s6 = x2 * x3 + x1 * x4;
        t->s7 = x2 * x4 - x1 * x3;

        if (pass == 0) {
            zcosg = zcosgl;
            zsing = zsingl;
            zcosi = zcosil;
            zsini = zsinil;
            zcosh = zcoshl * cnodm + zsinhl * snodm;
            zsinh = snodm * zcoshl - cnodm * zsinhl;
            cc = C1L;
        }
    }

    /* computePerturbCoeffs L69-105 */
    const ls_terms *S = &T[0], *L = &T[1];
    s->se2 = 2.0 * S->s1 * S->s6;
    s->se3 = 2.0 * S->s1 * S->s7;
    s->si2 = 2.0 * S->s2 * S->z12;
    s->si3 = 2.0 * S->s2 * (S->z13 - S->z11);
    s->sl2 = -2.0 * S->s3 * S->z2;
    s->sl3 = -2.0 * S->s3 * (S->z3 - S->z1);
    s->sl4 = -2.0 * S->s3 * (-21.0 - 9.0 * emsq) * ZES;
    s->sgh2 = 2.0 * S->s4 * S->z32;
    s->sgh3 = 2.0 * S->s4 * (S->z33 - S->z31);
    s->sgh4 = -18.0 * S->s4 * ZES;
    s->sh2 = -2.0 * S->s2 * S->z22;
    s->sh3 = -2.0 * S->s2 * (S->z23 - S->z21);

    s->ee2 = 2.0 * L->s1 * L->s6;
    s->e3 = 2.0 * L->s1 * L->s7;
    s->xi2 = 2.0 * L->s2 * L->z12;
    s->xi3 = 2.0 * L->s2 * (L->z13 - L->z11);
    s->xl2 = -2.0 * L->s3 * L->z2;
    s->xl3 = -2.0 * L->s3 * (L->z3 - L->z1);
    s->xl4 = -2.0 * L->s3 * (-21.0 - 9.0 * emsq) * ZEL;
    s->xgh2 = 2.0 * L->s4 * L->z32;
    s->xgh3 = 2.0 * L->s4 * (L->z33 - L->z31);
    s->xgh4 = -18.0 * L->s4 * ZEL;
    s->xh2 = -2.0 * L->s2 * L->z22;
    s->xh3 = -2.0 * L->s2 * (L->z23 - L->z21);

    s->zmol = pmod(4.7199672 + 0.22997150 * day - gam, TWOPI);
    s->zmos = pmod(6.2565837 + 0.017201977 * day, TWOPI);

    /* ---- dsinit L525-657 */
    const double eosq = emsq;
    const double cosisq = s->cosio2;
    const double sini2 = s->sinio * s->sinio;
    const double xpidot = s->argpdot + s->nodedot;
    const int near_eq = (s->inclo < NEAR_EQUATORIAL) || (s->inclo > PI - NEAR_EQUATORIAL);

    double ses = S->s1 * ZNS * S->s5;
    double sis = S->s2 * ZNS * (S->z11 + S->z13);
    double sls = -ZNS * S->s3 * (S->z1 + S->z3 - 14.0 - 6.0 * emsq);
    double sghs = S->s4 * ZNS * (S->z31 + S->z33 - 6.0);
    double shs = -ZNS * S->s2 * (S->z21 + S->z23);
    if (near_eq) shs = 0.0;
    if (sinim != 0.0) shs = shs / sinim;
    double sgs = sghs - cosim * shs;

    s->dedt = ses + L->s1 * ZNL * L->s5;
    s->didt = sis + L->s2 * ZNL * (L->z11 + L->z13);
    s->dmdt = sls - ZNL * L->s3 * (L->z1 + L->z3 - 14.0 - 6.0 * emsq);
    double sghl = L->s4 * ZNL * (L->z31 + L->z33 - 6.0);
    double shll = -ZNL * L->s2 * (L->z21 + L->z23);
    if (near_eq) shll = 0.0;
    s->domdt = sgs + sghl;
    s->dnodt = shs;
    if (sinim != 0.0) {
        s->domdt -= cosim / sinim * shll;
        s->dnodt += shll / sinim;
    }

    if (nm >= 0.00826 && nm <= 0.00924 && s->ecco >= 0.5)
        s->irez = 2;
    else if (nm >= 0.0034906585 && nm <= 0.0052359877)
        s->irez = 1;
    else
        s->irez = 0;

    if (s->irez == 1) {
        double g200 = 1.0 + eosq * (-2.5 + 0.8125 * eosq);
        double g310 = 1.0 + 2.0 * eosq;
        double g300 = 1.0 + eosq * (-6.0 + 6.60937 * eosq);
        double f220 = 0.75 * (1.0 + s->cosio) * (1.0 + s->cosio);
        double f311 = 0.9375 * sini2 * (1.0 + 3.0 * s->cosio) - 0.75 * (1.0 + s->cosio);
        double f330 = 1.0 + s->cosio;
        f330 = 1.875 * f330 * f330 * f330;
        double aonv = 1.0 / s->a;
        double t1 = 3.0 * nm * nm * aonv * aonv;
        s->del2 = 2.0 * t1 * f220 * g200 * Q22;
        s->del3 = 3.0 * t1 * f330 * g300 * Q33 * aonv;
        s->del1 = t1 * f311 * g310 * Q31 * aonv;
        s->xlamo = pmod(s->mo + s->nodeo + s->argpo - s->gsto, TWOPI);
        s->xfact = s->mdot + xpidot - RPTIM + s->dmdt + s->domdt + s->dnodt - s->no_unkozai;
    } else if (s->irez == 2) {
        const double e = s->ecco;
        static const double g211lo[] = {3.616, -13.2470, 16.2900};
        static const double g211hi[] = {-72.099, 331.819, -508.738, 266.724};
        static const double g310lo[] = {-19.302, 117.3900, -228.4190, 156.591};
        static const double g310hi[] = {-346.844, 1582.851, -2415.925, 1246.113};
        static const double g322lo[] = {-18.9068, 109.7927, -214.6334, 146.5816};
        static const double g322hi[] = {-342.585, 1554.908, -2366.899, 1215.972};
        static const double g410lo[] = {-41.122, 242.6940, -471.0940, 313.953};
        static const double g410hi[] = {-1052.797, 4758.686, -7193.992, 3651.957};
        static const double g422lo[] = {-146.407, 841.8800, -1629.014, 1083.435};
        static const double g422hi[] = {-3581.690, 16178.110, -24462.770, 12422.520};
        static const double g520lo[] = {-532.114, 3017.977, -5740.032, 3708.276};
        static const double g520hi[] = {-5149.66, 29936.92, -54087.36, 31324.56};
        static const double g521lo[] = {-822.71072, 4568.6173, -8491.4146, 5337.524};
        static const double g521hi[] = {-51752.104, 218913.95, -309468.16, 146349.42};
        static const double g532lo[] = {-853.66600, 4690.2500, -8624.7700, 5341.400};
        static const double g532hi[] = {-40023.880, 170470.89, -242699.48, 115605.82};
        static const double g533lo[] = {-919.22770, 4988.6100, -9064.7700, 5542.21};
        static const double g533hi[] = {-37995.780, 161616.52, -229838.20, 109377.94};
        double g201 = -0.306 - (e - 0.64) * 0.440;
        double g211 = (e <= 0.65) ? poly(e, g211lo, 3) : poly(e, g211hi, 4);
        double g310 = (e <= 0.65) ? poly(e, g310lo, 4) : poly(e, g310hi, 4);
        double g322 = (e <= 0.65) ? poly(e, g322lo, 4) : poly(e, g322hi, 4);
        double g410 = (e <= 0.65) ? poly(e, g410lo, 4) : poly(e, g410hi, 4);
        double g422 = (e <= 0.65) ? poly(e, g422lo, 4) : poly(e, g422hi, 4);
        double g520;
        if (e <= 0.65)
            g520 = poly(e, g520lo, 4);
        else if (e > 0.715)
            g520 = poly(e, g520hi, 4);
        else
            g520 = 1464.74 - 4664.75 * e + 3763.64 * e * e;
        double g521 = (e < 0.7) ? poly(e, g521lo, 4) : poly(e, g521hi, 4);
        double g532 = (e < 0.7) ? poly(e, g532lo, 4) : poly(e, g532hi, 4);
        double g533 = (e < 0.7) ? poly(e, g533lo, 4) : poly(e, g533hi, 4);

        const double c = s->cosio, si = s->sinio;
        double f220 = 0.75 * (1.0 + 2.0 * c + cosisq);
        double f221 = 1.5 * sini2;
        double f321 = 1.875 * si * (1.0 - 2.0 * c - 3.0 * cosisq);
        double f322 = -1.875 * si * (1.0 + 2.0 * c - 3.0 * cosisq);
        double f441 = 35.0 * sini2 * f220;
        double f442 = 39.3750 * sini2 * sini2;
        double f522 = 9.84375 * si *
                      (sini2 * (1.0 - 2.0 * c - 5.0 * cosisq) + 0.33333333 * (-2.0 + 4.0 * c + 6.0 * cosisq));
        double f523 = si * (4.92187512 * sini2 * (-2.0 - 4.0 * c + 10.0 * cosisq) +
                            6.56250012 * (1.0 + 2.0 * c - 3.0 * cosisq));
        double f542 = 29.53125 * si * (2.0 - 8.0 * c + cosisq * (-12.0 + 8.0 * c + 10.0 * cosisq));
        double f543 = 29.53125 * si * (-2.0 - 8.0 * c + cosisq * (12.0 + 8.0 * c - 10.0 * cosisq));

        double aonv = 1.0 / s->a;
        double t1 = 3.0 * nm * nm * aonv * aonv;
        double t = t1 * ROOT22;
        s->d2201 = t * f220 * g201;
        s->d2211 = t * f221 * g211;
        t1 = t1 * aonv;
        t = t1 * ROOT32;
        s->d3210 = t * f321 * g310;
        s->d3222 = t * f322 * g322;
        t1 = t1 * aonv;
        t = 2.0 * t1 * ROOT44;
        s->d4410 = t * f441 * g410;
        s->d4422 = t * f442 * g422;
        t1 = t1 * aonv;
        t = t1 * ROOT52;
        s->d5220 = t * f522 * g520;
        s->d5232 = t * f523 * g532;
        t = 2.0 * t1 * ROOT54;
        s->d5421 = t * f542 * g521;
        s->d5433 = t * f543 * g533;

        s->xlamo = pmod(s->mo + s->nodeo + s->nodeo - s->gsto - s->gsto, TWOPI);
        s->xfact = s->mdot + s->dmdt + 2.0 * (s->nodedot + s->dnodt - RPTIM) - s->no_unkozai;
    }
}

/* Satellite.zig L16-21 / Constellation.zig L115-126: try SGP4, fall back to SDP4 */
int orc_sat_init(const orc_tle *tle, int grav, orc_sat *s)
{
    double perige;
    int rc = sgp4_common_init(tle, grav, s, &perige);
    if (rc != ORC_OK) return rc;
    if (TWOPI / s->no_unkozai > DEEP_SPACE_MINUTES) {
        /* Sdp4.initElements L174-274: isimp = true, higher-order terms zero */
        s->is_deep = 1;
        s->isimp = 1;
        s->gsto = orc_gstime(s->epoch_jd);
        sdp4_deep_init(s);
    } else {
        sgp4_higher_order(s, perige);
    }
    return ORC_OK;
}

int orc_sgp4_init_only(const orc_tle *tle, int grav, orc_sat *s)
{
    double perige;
    int rc = sgp4_common_init(tle, grav, s, &perige);
    if (rc != ORC_OK) return rc;
    if (TWOPI / s->no_unkozai > DEEP_SPACE_MINUTES) return ORC_ERR_DEEP_SPACE;
    sgp4_higher_order(s, perige);
    return ORC_OK;
}

/* ------------------------------------------------------------------ propagation */

typedef struct {
    double mm, argpm, nodem, em, am;
} secular;

/* per-call inclination-dependent constants (near-earth: from init; deep: recomputed, Sdp4.zig L940-954) */
typedef struct {
    double inclo, aycof, xlcof, con41, x1mth2, x7thm1, sinio, cosio;
} incl_terms;

/* Sgp4.solveKepler L495-546 + applyShortPeriodCorrections L557-571 + computePositionVelocity L573-603.
 * mrt_out receives the corrected radius (Earth radii) for the SDP4 decay check. */
static void kepler_posvel(const orc_grav *g, double vkmpersec, const incl_terms *it, const secular *sec,
                          double nm, double r[3], double v[3], double *mrt_out)
{
    double temp = 1.0 / (sec->am * (1.0 - sec->em * sec->em));
    double axnl = sec->em * cos(sec->argpm);
    double aynl = sec->em * sin(sec->argpm) + temp * it->aycof;
    double xl = pmod(sec->mm + sec->argpm + sec->nodem + temp * it->xlcof * axnl, TWOPI);

    double u = pmod(xl - sec->nodem, TWOPI);
    double eo1 = u, sineo1 = 0.0, coseo1 = 1.0, tem5 = 9999.9;
    int ktr = 1;
    while (fabs(tem5) >= 1.0e-12 && ktr <= 10) {
        sineo1 = sin(eo1);
        coseo1 = cos(eo1);
        tem5 = 1.0 - coseo1 * axnl - sineo1 * aynl;
        tem5 = (u - aynl * coseo1 + axnl * sineo1 - eo1) / tem5;
        if (fabs(tem5) >= 0.95) tem5 = (tem5 > 0.0) ? 0.95 : -0.95;
        eo1 += tem5;
        ktr++;
    }

    double ecose = axnl * coseo1 + aynl * sineo1;
    double esine = axnl * sineo1 - aynl * coseo1;
    double el2 = axnl * axnl + aynl * aynl;
    double pl = sec->am * (1.0 - el2);
    double betal = sqrt(1.0 - el2);
    double rl = sec->am * (1.0 - ecose);
    double rdotl = sqrt(sec->am) * esine / rl;
    double rvdotl = sqrt(pl) / rl;
    double a_over_r = sec->am / rl;
    double esine_term = esine / (1.0 + betal);
    double sinu = a_over_r * (sineo1 - aynl - axnl * esine_term);
    double cosu = a_over_r * (coseo1 - axnl + aynl * esine_term);
    u = atan2(sinu, cosu);
    double sin2u = 2.0 * sinu * cosu;
    double cos2u = 1.0 - 2.0 * sinu * sinu;

    double tp = 1.0 / pl;
    double temp1 = 0.5 * g->j2 * tp;
    double temp2 = temp1 * tp;
    double mrt = rl * (1.0 - 1.5 * temp2 * betal * it->con41) + 0.5 * temp1 * it->x1mth2 * cos2u;
    double su = u - 0.25 * temp2 * it->x7thm1 * sin2u;
    double xnode = sec->nodem + 1.5 * temp2 * it->cosio * sin2u;
    double xinc = it->inclo + 1.5 * temp2 * it->cosio * it->sinio * cos2u;
    double mvt = rdotl - nm * temp1 * it->x1mth2 * sin2u / g->xke;
    double rvdot = rvdotl + nm * temp1 * (it->x1mth2 * cos2u + 1.5 * it->con41) / g->xke;

    double sinsu = sin(su), cossu = cos(su);
    double snod = sin(xnode), cnod = cos(xnode);
    double sini = sin(xinc), cosi = cos(xinc);
    double xmx = -snod * cosi, xmy = cnod * cosi;
    double ux = xmx * sinsu + cnod * cossu;
    double uy = xmy * sinsu + snod * cossu;
    double uz = sini * sinsu;
    double vx = xmx * cossu - cnod * sinsu;
    double vy = xmy * cossu - snod * sinsu;
    double vz = sini * cossu;
    double rs = mrt * g->radius_km;
    r[0] = rs * ux;
    r[1] = rs * uy;
    r[2] = rs * uz;
    v[0] = (mvt * ux + rvdot * vx) * vkmpersec;
    v[1] = (mvt * uy + rvdot * vy) * vkmpersec;
    v[2] = (mvt * uz + rvdot * vz) * vkmpersec;
    *mrt_out = mrt;
}

/* Sgp4.propagateElements L419-425 with updateSecular L435-477.  No runtime errors on this path. */
static int sgp4_propagate(const orc_sat *s, double t, double r[3], double v[3])
{
    double t2 = t * t;
    double tempa = 1.0 - s->cc1 * t;
    double tempe = s->bstar * s->cc4 * t;
    double templ = s->t2cof * t2;
    double xmdf = s->mo + s->mdot * t;
    double argpdf = s->argpo + s->argpdot * t;
    double nodedf = s->nodeo + s->nodedot * t;
    double argpm = argpdf, mm = xmdf;
    double nodem = nodedf + s->xnodcf * t2;
    if (!s->isimp) {
        double delomg = s->omgcof * t;
        double dt = 1.0 + s->eta * cos(xmdf);
        double delm = s->xmcof * (dt * dt * dt - s->delmo);
        double temp = delomg + delm;
        mm = xmdf + temp;
        argpm = argpdf - temp;
        double t3 = t2 * t, t4 = t3 * t;
        tempa = tempa - s->d2 * t2 - s->d3 * t3 - s->d4 * t4;
        tempe = tempe + s->bstar * s->cc5 * (sin(mm) - s->sinmao);
        templ = templ + s->t3cof * t3 + t4 * (s->t4cof + t * s->t5cof);
    }
    secular sec;
    sec.am = s->a_base * tempa * tempa;
    sec.em = s->ecco - tempe;
    if (sec.em < ECC_FLOOR) sec.em = ECC_FLOOR;
    mm = mm + s->no_unkozai * templ;
    double xlm = mm + argpm + nodem;
    sec.nodem = pmod(nodem, TWOPI);
    sec.argpm = pmod(argpm, TWOPI);
    sec.mm = pmod(xlm - sec.argpm - sec.nodem, TWOPI);

    double nm = s->g.xke / pow(sec.am, 1.5);
    incl_terms it = {s->inclo, s->aycof, s->xlcof, s->con41, s->x1mth2, s->x7thm1, s->sinio, s->cosio};
    double mrt;
    kepler_posvel(&s->g, s->vkmpersec, &it, &sec, nm, r, v, &mrt);
    return ORC_OK;
}

/* Sdp4.computeResonanceAccel L824-866 */
static void resonance_accel(const orc_sat *s, double xli, double xni, double atime, double *xndt,
                            double *xnddt, double *xldot)
{
    *xldot = xni + s->xfact;
    if (s->irez == 2) {
        double xomi = s->argpo + s->argpdot * atime;
        double x2omi = xomi + xomi;
        double x2li = xli + xli;
        *xndt = s->d2201 * sin(x2omi + xli - G22) + s->d2211 * sin(xli - G22) +
                s->d3210 * sin(xomi + xli - G32) + s->d3222 * sin(-xomi + xli - G32) +
                s->d4410 * sin(x2omi + x2li - G44) + s->d4422 * sin(x2li - G44) +
                s->d5220 * sin(xomi + xli - G52) + s->d5232 * sin(-xomi + xli - G52) +
                s->d5421 * sin(xomi + x2li - G54) + s->d5433 * sin(-xomi + x2li - G54);
        *xnddt = (s->d2201 * cos(x2omi + xli - G22) + s->d2211 * cos(xli - G22) +
                  s->d3210 * cos(xomi + xli - G32) + s->d3222 * cos(-xomi + xli - G32) +
                  s->d5220 * cos(xomi + xli - G52) + s->d5232 * cos(-xomi + xli - G52) +
                  2.0 * (s->d4410 * cos(x2omi + x2li - G44) + s->d4422 * cos(x2li - G44) +
                         s->d5421 * cos(xomi + x2li - G54) + s->d5433 * cos(-xomi + x2li - G54))) *
                 (*xldot);
    } else {
        *xndt = s->del1 * sin(xli - FASX2) + s->del2 * sin(2.0 * (xli - FASX4)) +
                s->del3 * sin(3.0 * (xli - FASX6));
        *xnddt = (s->del1 * cos(xli - FASX2) + 2.0 * s->del2 * cos(2.0 * (xli - FASX4)) +
                  3.0 * s->del3 * cos(3.0 * (xli - FASX6))) *
                 (*xldot);
    }
}

void orc_carry_init(const orc_sat *s, orc_carry *c)
{
    c->atime = 0.0;
    c->xli = s->xlamo;
    c->xni = s->no_unkozai;
}

/* Sdp4.propagateElementsCarry L881-970 with dspace L774-820 and dpper L681-759 */
static int sdp4_propagate(const orc_sat *s, double t, orc_carry *c, double r[3], double v[3])
{
    double t2 = t * t;
    double tempa = 1.0 - s->cc1 * t;
    double tempe = s->bstar * s->cc4 * t;
    double templ = s->t2cof * t2;
    double xmdf = s->mo + s->mdot * t;
    double argpdf = s->argpo + s->argpdot * t;
    double nodedf = s->nodeo + s->nodedot * t;

    double em = s->ecco, argpm = argpdf, inclm = s->inclo, mm = xmdf;
    double nodem = nodedf + s->xnodcf * t2;
    double nm = s->no_unkozai;

    /* dspace */
    em += s->dedt * t;
    inclm += s->didt * t;
    argpm += s->domdt * t;
    nodem += s->dnodt * t;
    mm += s->dmdt * t;
    if (s->irez != 0) {
        if (c->atime == 0.0 || t * c->atime <= 0.0 || fabs(t) < fabs(c->atime)) {
            c->atime = 0.0;
            c->xni = s->no_unkozai;
            c->xli = s->xlamo;
        }
        double delt = (t > 0.0) ? STEPP : -STEPP;
        double xndt, xnddt, xldot;
        while (fabs(t - c->atime) >= STEPP) {
            resonance_accel(s, c->xli, c->xni, c->atime, &xndt, &xnddt, &xldot);
            c->xli += xldot * delt + xndt * STEP2;
            c->xni += xndt * delt + xnddt * STEP2;
            c->atime += delt;
        }
        double ft = t - c->atime;
        resonance_accel(s, c->xli, c->xni, c->atime, &xndt, &xnddt, &xldot);
        nm = c->xni + xndt * ft + xnddt * ft * ft * 0.5;
        double xl = c->xli + xldot * ft + xndt * ft * ft * 0.5;
        double theta = pmod(s->gsto + t * RPTIM, TWOPI);
        if (s->irez != 2)
            mm = xl - nodem - argpm + theta;
        else
            mm = xl - 2.0 * nodem + 2.0 * theta;
        double dndt = nm - s->no_unkozai;
        nm = s->no_unkozai + dndt;
    }

    if (nm <= 0.0) return ORC_ERR_DECAYED;
    double am = pow(s->g.xke / nm, 2.0 / 3.0) * tempa * tempa;
    nm = s->g.xke / pow(am, 1.5);
    em -= tempe;
    if (em >= 1.0 || em < -0.001) return ORC_ERR_ECCENTRICITY;
    if (em < 1.0e-6) em = 1.0e-6;
    if (am < 0.95) return ORC_ERR_DECAYED;

    mm += s->no_unkozai * templ;
    double xlm = mm + argpm + nodem;
    nodem = pmod(nodem, TWOPI);
    argpm = pmod(argpm, TWOPI);
    mm = pmod(xlm - argpm - nodem, TWOPI);

    /* dpper */
    {
        double zm = s->zmos + ZNS * t;
        double zf = zm + 2.0 * ZES * sin(zm);
        double sinzf = sin(zf);
        double f2 = 0.5 * sinzf * sinzf - 0.25;
        double f3 = -0.5 * sinzf * cos(zf);
        double ses = s->se2 * f2 + s->se3 * f3;
        double sis = s->si2 * f2 + s->si3 * f3;
        double sls = s->sl2 * f2 + s->sl3 * f3 + s->sl4 * sinzf;
        double sghs = s->sgh2 * f2 + s->sgh3 * f3 + s->sgh4 * sinzf;
        double shs = s->sh2 * f2 + s->sh3 * f3;
        zm = s->zmol + ZNL * t;
        zf = zm + 2.0 * ZEL * sin(zm);
        sinzf = sin(zf);
        f2 = 0.5 * sinzf * sinzf - 0.25;
        f3 = -0.5 * sinzf * cos(zf);
        double sel = s->ee2 * f2 + s->e3 * f3;
        double sil = s->xi2 * f2 + s->xi3 * f3;
        double sll = s->xl2 * f2 + s->xl3 * f3 + s->xl4 * sinzf;
        double sghl = s->xgh2 * f2 + s->xgh3 * f3 + s->xgh4 * sinzf;
        double shl = s->xh2 * f2 + s->xh3 * f3;
        double pe = ses + sel, pinc = sis + sil, pl = sls + sll;
        double pgh = sghs + sghl, ph = shs + shl;

        inclm += pinc;
        em += pe;
        double sinip = sin(inclm), cosip = cos(inclm);
        if (inclm >= 0.2) {
            ph /= sinip;
            pgh -= cosip * ph;
            argpm += pgh;
            nodem += ph;
            mm += pl;
        } else {
            double sinop = sin(nodem), cosop = cos(nodem);
            double alfdp = sinip * sinop, betdp = sinip * cosop;
            double dalf = ph * cosop + pinc * cosip * sinop;
            double dbet = -ph * sinop + pinc * cosip * cosop;
            alfdp += dalf;
            betdp += dbet;
            nodem = pmod(nodem, TWOPI);
            double xls = mm + argpm + cosip * nodem;
            double dls = pl + pgh - pinc * nodem * sinip;
            double xnoh = nodem;
            nodem = atan2(alfdp, betdp);
            if (fabs(xnoh - nodem) > PI) {
                if (nodem < xnoh)
                    nodem += TWOPI;
                else
                    nodem -= TWOPI;
            }
            mm += pl;
            argpm = xls + dls - mm - cosip * nodem;
        }
    }

    if (inclm < 0.0) {
        inclm = -inclm;
        nodem += PI;
        argpm -= PI;
    }
    if (em < 1.0e-6) em = 1.0e-6;
    if (em >= 1.0) return ORC_ERR_ECCENTRICITY;

    double sinip = sin(inclm), cosip = cos(inclm), cosip2 = cosip * cosip;
    incl_terms it;
    it.inclo = inclm;
    it.sinio = sinip;
    it.cosio = cosip;
    it.aycof = -0.5 * s->g.j3oj2 * sinip;
    it.xlcof = (-0.25 * s->g.j3oj2 * sinip * (3.0 + 5.0 * cosip)) /
               ((fabs(cosip + 1.0) > 1.5e-12) ? 1.0 + cosip : 1.5e-12);
    it.x1mth2 = 1.0 - cosip2;
    it.con41 = 3.0 * cosip2 - 1.0;
    it.x7thm1 = 7.0 * cosip2 - 1.0;

    secular sec = {mm, argpm, nodem, em, am};
    double mrt;
    kepler_posvel(&s->g, s->vkmpersec, &it, &sec, nm, r, v, &mrt);
    if (mrt < 1.0) return ORC_ERR_DECAYED;
    return ORC_OK;
}

int orc_sat_propagate_carry(const orc_sat *s, double tsince, orc_carry *c, double r[3], double v[3])
{
    int rc = s->is_deep ? sdp4_propagate(s, tsince, c, r, v) : sgp4_propagate(s, tsince, r, v);
    if (rc != ORC_OK) {
        r[0] = r[1] = r[2] = 0.0;
        v[0] = v[1] = v[2] = 0.0;
    }
    return rc;
}

int orc_sat_propagate(const orc_sat *s, double tsince, double r[3], double v[3])
{
    orc_carry c;
    orc_carry_init(s, &c);
    return orc_sat_propagate_carry(s, tsince, &c, r, v);
}

/* ------------------------------------------------------------------ frames (WorldCoordinateSystem.zig) */

/* L146-154 */
double orc_julian_to_gmst(double jd)
{
    double d = jd - 2451545.0;
    double t = d / 36525.0;
    double gmst = 280.46061837 + 360.98564736629 * d + 0.000387933 * t * t - t * t * t / 38710000.0;
    gmst = pmod(gmst, 360.0);
    if (gmst < 0) gmst += 360.0;
    return gmst * DEG2RAD;
}

/* Constellation.zig L54-56 */
void orc_eci_to_ecef(const double p[3], double sin_g, double cos_g, double out[3])
{
    double x = p[0] * cos_g + p[1] * sin_g;
    double y = p[1] * cos_g - p[0] * sin_g;
    out[0] = x;
    out[1] = y;
    out[2] = p[2];
}

/* L98-121: (lat, lon) in RADIANS, alt km -- what Constellation's geodetic mode emits (L497) */
void orc_ecef_to_geodetic(const double e[3], double lla[3])
{
    const double f = 1.0 / 298.257223563;
    const double e2 = 2.0 * f - f * f;
    const double a = 6378.137;
    double x = e[0], y = e[1], z = e[2];
    double lon = atan2(y, x);
    double p = sqrt(x * x + y * y);
    double lat = atan2(z, p * (1.0 - e2));
    for (int i = 0; i < 10; i++) {
        double prev = lat;
        double sl = sin(lat);
        double N = a / sqrt(1.0 - e2 * sl * sl);
        lat = atan2(z + e2 * N * sl, p);
        if (fabs(lat - prev) < 1e-12) break;
    }
    double sl = sin(lat), cl = cos(lat);
    double N = a / sqrt(1.0 - e2 * sl * sl);
    lla[0] = lat;
    lla[1] = lon;
    lla[2] = p / cl - N;
}

/* ------------------------------------------------------------------ constellation driver */

int orc_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* Constellation.zig L413-434 (tsince), L46-51 (indexing), L478-509 (output modes) */
void orc_propagate_constellation(const orc_sat *sats, size_t n_sats, const double *times, size_t n_times,
                                 const double *offsets, double *pos, double *vel, int mode,
                                 double reference_jd, const uint8_t *mask, int layout, size_t stride,
                                 uint8_t *err, int nthreads)
{
    double *sin_g = NULL, *cos_g = NULL;
    if (mode != ORC_TEME) {
        sin_g = (double *)malloc(sizeof(double) * (n_times ? n_times : 1));
        cos_g = (double *)malloc(sizeof(double) * (n_times ? n_times : 1));
        for (size_t t = 0; t < n_times; t++) {
            double g = orc_julian_to_gmst(reference_jd + times[t] / 1440.0);
            sin_g[t] = sin(g);
            cos_g[t] = cos(g);
        }
    }
    if (stride == 0) stride = n_sats;
    long ns = (long)n_sats;
    (void)nthreads;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 16) num_threads(nthreads > 1 ? nthreads : 1)
#endif
    for (long si = 0; si < ns; si++) {
        size_t s = (size_t)si;
        if (mask && !mask[s]) continue;
        orc_carry c;
        orc_carry_init(&sats[s], &c);
        double off = offsets ? offsets[s] : 0.0;
        for (size_t t = 0; t < n_times; t++) {
            double r[3], v[3];
            int rc = orc_sat_propagate_carry(&sats[s], times[t] + off, &c, r, v);
            size_t ob = (layout == ORC_SAT_MAJOR) ? (s * n_times + t) * 3 : (t * stride + s) * 3;
            if (err) err[s * n_times + t] = (uint8_t)rc;
            if (rc == ORC_OK && mode != ORC_TEME) {
                double e[3];
                orc_eci_to_ecef(r, sin_g[t], cos_g[t], e);
                if (mode == ORC_GEODETIC)
                    orc_ecef_to_geodetic(e, r);
                else
                    memcpy(r, e, sizeof e);
                orc_eci_to_ecef(v, sin_g[t], cos_g[t], e);
                memcpy(v, e, sizeof e);
            }
            pos[ob] = r[0];
            pos[ob + 1] = r[1];
            pos[ob + 2] = r[2];
            if (vel) {
                vel[ob] = v[0];
                vel[ob + 1] = v[1];
                vel[ob + 2] = v[2];
            }
        }
    }
    free(sin_g);
    free(cos_g);
}

/* ------------------------------------------------------------------ conjunction screening */
/* Constellation.screenConstellation, src/Constellation.zig L683-756: time-outer loop, target and
 * every satellite propagated per step, both rotated to ECEF (L719, L737), running minimum of the
 * squared distance with a strict '<' (L744), start value threshold^2 / index 0 (L700-703), the
 * target itself skipped (L735), sqrt at the end (L752-754).  Steps at which the target or the
 * satellite fails are skipped (`catch continue`, L718, L730 -- the reference skips the satellite's
 * whole batch of 8 there; this scalar restatement skips the satellite). */
void orc_screen_target(const orc_sat *sats, size_t n_sats, const double *times, size_t n_times,
                       const double *offsets, size_t target, double threshold, double reference_jd,
                       const uint8_t *failed, double *out_min_dist, uint32_t *out_min_t)
{
    const double thr2 = threshold * threshold;
    for (size_t i = 0; i < n_sats; i++) {
        out_min_dist[i] = thr2;
        out_min_t[i] = 0;
    }
    orc_carry *carry = (orc_carry *)malloc(sizeof(orc_carry) * (n_sats ? n_sats : 1));
    for (size_t i = 0; i < n_sats; i++) orc_carry_init(&sats[i], &carry[i]);
    for (size_t t = 0; t < n_times; t++) {
        const double g = orc_julian_to_gmst(reference_jd + times[t] / 1440.0);
        const double sg = sin(g), cg = cos(g);
        double r[3], v[3], tp[3];
        if (failed && failed[target]) continue;
        /* the target is propagated statelessly here and with its carry in the loop below; both give
         * the same state (tests/test_oracle_golden.py G9) */
        if (orc_sat_propagate(&sats[target], times[t] + (offsets ? offsets[target] : 0.0), r, v) != ORC_OK) continue;
        orc_eci_to_ecef(r, sg, cg, tp);
        for (size_t s = 0; s < n_sats; s++) {
            if (s == target) continue;
            if (failed && failed[s]) continue;
            if (orc_sat_propagate_carry(&sats[s], times[t] + (offsets ? offsets[s] : 0.0), &carry[s], r, v) != ORC_OK) continue;
            double e[3];
            orc_eci_to_ecef(r, sg, cg, e);
            const double dx = tp[0] - e[0], dy = tp[1] - e[1], dz = tp[2] - e[2];
            const double d2 = dx * dx + dy * dy + dz * dz;
            if (d2 < out_min_dist[s]) {
                out_min_dist[s] = d2;
                out_min_t[s] = (uint32_t)t;
            }
        }
    }
    for (size_t i = 0; i < n_sats; i++) out_min_dist[i] = sqrt(out_min_dist[i]);
    free(carry);
}

/* spatialHash, bindings/python/src/conjunction.zig L152-160 */
static uint32_t spatial_hash(int32_t cx, int32_t cy, int32_t cz)
{
    uint32_t h = (uint32_t)cx;
    h *= 2654435761u;
    h ^= (uint32_t)cy;
    h *= 2654435761u;
    h ^= (uint32_t)cz;
    h *= 2654435761u;
    return h;
}

/* coarseScreen, bindings/python/src/conjunction.zig L11-150: positions satellite-major
 * (s*n_times + t)*3; per step a chained hash table of cells (edge = threshold, 2^16 buckets), every
 * satellite walks the chains of its 27 neighbour cells and reports partners with a larger index in
 * exactly that cell closer than the threshold.  Same traversal order as the reference. */
size_t orc_coarse_screen(const double *positions, size_t num_sats, size_t num_times, double threshold,
                         const uint8_t *valid_mask, uint32_t *out_pairs, uint32_t *out_t, size_t max_results)
{
    const double inv_cell = 1.0 / threshold, thr2 = threshold * threshold;
    const uint32_t TABLE = 1u << 16, MASK = TABLE - 1u, EMPTY = 0xffffffffu;
    size_t count = 0;
    int32_t *cx = (int32_t *)malloc(sizeof(int32_t) * 3 * (num_sats ? num_sats : 1));
    int32_t *cy = cx + num_sats, *cz = cx + 2 * num_sats;
    uint32_t *hashes = (uint32_t *)malloc(sizeof(uint32_t) * (num_sats ? num_sats : 1));
    uint32_t *head = (uint32_t *)malloc(sizeof(uint32_t) * TABLE);
    uint32_t *next = (uint32_t *)malloc(sizeof(uint32_t) * (num_sats ? num_sats : 1));
    for (size_t t = 0; t < num_times; t++) {
        memset(head, 0xff, sizeof(uint32_t) * TABLE);
        for (size_t s = 0; s < num_sats; s++) {
            if (valid_mask && !valid_mask[s]) { hashes[s] = EMPTY; continue; }
            const double *q = positions + (s * num_times + t) * 3;
            if (!isfinite(q[0])) { hashes[s] = EMPTY; continue; }
            cx[s] = (int32_t)floor(q[0] * inv_cell);
            cy[s] = (int32_t)floor(q[1] * inv_cell);
            cz[s] = (int32_t)floor(q[2] * inv_cell);
            const uint32_t h = spatial_hash(cx[s], cy[s], cz[s]) & MASK;
            hashes[s] = h;
            next[s] = head[h];
            head[h] = (uint32_t)s;
        }
        for (size_t s = 0; s < num_sats; s++) {
            if (hashes[s] == EMPTY) continue;
            const double *q = positions + (s * num_times + t) * 3;
            for (int dx = -1; dx <= 1; dx++)
                for (int dy = -1; dy <= 1; dy++)
                    for (int dz = -1; dz <= 1; dz++) {
                        const int32_t nx = cx[s] + dx, ny = cy[s] + dy, nz = cz[s] + dz;
                        uint32_t idx = head[spatial_hash(nx, ny, nz) & MASK];
                        while (idx != EMPTY) {
                            const uint32_t other = idx;
                            idx = next[idx];
                            if (other <= s) continue;
                            if (cx[other] != nx || cy[other] != ny || cz[other] != nz) continue;
                            const double *o = positions + ((size_t)other * num_times + t) * 3;
                            const double ex = q[0] - o[0], ey = q[1] - o[1], ez = q[2] - o[2];
                            if (ex * ex + ey * ey + ez * ez < thr2) {
                                if (count >= max_results) goto done;
                                out_pairs[2 * count] = (uint32_t)s;
                                out_pairs[2 * count + 1] = other;
                                out_t[count] = (uint32_t)t;
                                count++;
                            }
                        }
                    }
        }
    }
done:
    free(cx);
    free(hashes);
    free(head);
    free(next);
    return count;
}

/* ------------------------------------------------------------------ field access by name (tests) */
#define F(n)                                                                                                 \
    if (!strcmp(name, #n)) return (double)s->n
double orc_sat_field(const orc_sat *s, const char *name)
{
    F(epoch_jd); F(no_kozai); F(ecco); F(inclo); F(nodeo); F(argpo); F(mo); F(bstar);
    F(no_unkozai); F(a); F(sinio); F(cosio); F(con41); F(con42); F(x1mth2); F(x7thm1);
    F(mdot); F(argpdot); F(nodedot);
    F(cc1); F(cc4); F(cc5); F(t2cof); F(omgcof); F(xnodcf); F(xlcof); F(xmcof); F(aycof); F(eta);
    F(delmo); F(sinmao); F(d2); F(d3); F(d4); F(t3cof); F(t4cof); F(t5cof); F(a_base); F(vkmpersec);
    F(isimp); F(is_deep); F(irez);
    F(se2); F(se3); F(si2); F(si3); F(sl2); F(sl3); F(sl4); F(sgh2); F(sgh3); F(sgh4); F(sh2); F(sh3);
    F(ee2); F(e3); F(xi2); F(xi3); F(xl2); F(xl3); F(xl4); F(xgh2); F(xgh3); F(xgh4); F(xh2); F(xh3);
    F(zmol); F(zmos); F(dedt); F(didt); F(dmdt); F(domdt); F(dnodt);
    F(d2201); F(d2211); F(d3210); F(d3222); F(d4410); F(d4422); F(d5220); F(d5232); F(d5421); F(d5433);
    F(del1); F(del2); F(del3); F(xlamo); F(xfact); F(gsto);
    return NAN;
}
