/*
 * astroz_oracle.h -- CPU ORACLE for the batched SGP4/SDP4 hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load this library.  The product path
 * (astroz_amd/, include/, libastroz_hip.so) never links, imports or calls it.
 *
 * It is a plain-C, scalar, fp64/libm restatement of the algorithm of the reference's
 * *scalar* path (ATTron/astroz v0.12.0):
 *   src/Sgp4.zig   L108-417 (init), L419-603 (propagate)
 *   src/Sdp4.zig   L15-52 (constants), L174-679 (init), L681-970 (dpper/dspace/propagate)
 *   src/Tle.zig    L49-101, L277-304 (fixed-column parse, epoch -> JD)
 *   src/Datetime.zig L222-231 (year+doy -> JD)
 *   src/WorldCoordinateSystem.zig L87-154 (GMST, ECI->ECEF, ECEF->geodetic)
 *   src/Constellation.zig L46-51, L413-434, L478-509 (indexing, tsince, output modes)
 *
 * Parity pinning: the reference cannot be built here (Zig toolchain absent), so the oracle
 * is pinned by the golden vectors the reference's own tests hold,
 * committed under tests/golden/ and checked by tests/test_oracle_golden.py.
 */
#ifndef ASTROZ_ORACLE_H
#define ASTROZ_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* error codes follow bindings/python/src/shared.zig L40-47 (python-sgp4 numbering) */
enum {
    ORC_OK = 0,
    ORC_ERR_ECCENTRICITY = 1,
    ORC_ERR_DEEP_SPACE = 3,
    ORC_ERR_DECAYED = 6,
    ORC_ERR_BAD_TLE = -1
};

enum { ORC_WGS84 = 0, ORC_WGS72 = 1 }; /* shared.zig L21-27 */
enum { ORC_TEME = 0, ORC_ECEF = 1, ORC_GEODETIC = 2 }; /* Constellation.zig L30-34 */
enum { ORC_SAT_MAJOR = 0, ORC_TIME_MAJOR = 1 };        /* Constellation.zig L37-42 */

typedef struct {
    uint32_t satnum;
    char classification;
    int epoch_year; /* two digits */
    double epoch_day;
    double epoch_jd;
    double ndot;  /* as in the TLE: rev/day^2 / 2 */
    double bstar;
    double incl_deg, raan_deg, ecc, argp_deg, ma_deg, mm_revday;
    uint32_t elnum, revnum;
} orc_tle;

typedef struct {
    double radius_km, mu, j2, j3, j4, xke, tumin, j3oj2;
} orc_grav;

/* Sgp4.Elements (Sgp4.zig L33-94) + Sdp4.Elements (Sdp4.zig L109-148) */
typedef struct {
    orc_grav g;
    double epoch_jd;
    double no_kozai, ecco, inclo, nodeo, argpo, mo, bstar;
    double no_unkozai, a;
    double sinio, cosio, cosio2, cosio4;
    double con41, con42, x1mth2, x7thm1;
    double mdot, argpdot, nodedot;
    double cc1, cc4, cc5, t2cof, omgcof, xnodcf, xlcof, xmcof, aycof, eta, delmo, sinmao;
    double d2, d3, d4, t3cof, t4cof, t5cof;
    double a_base, vkmpersec;
    int isimp;
    /* deep space */
    int is_deep;
    int irez;
    double se2, se3, si2, si3, sl2, sl3, sl4, sgh2, sgh3, sgh4, sh2, sh3; /* solar */
    double ee2, e3, xi2, xi3, xl2, xl3, xl4, xgh2, xgh3, xgh4, xh2, xh3;  /* lunar */
    double zmol, zmos, dedt, didt, dmdt, domdt, dnodt;
    double d2201, d2211, d3210, d3222, d4410, d4422, d5220, d5232, d5421, d5433;
    double del1, del2, del3, xlamo, xfact, gsto;
} orc_sat;

typedef struct {
    double atime, xli, xni;
} orc_carry;

size_t orc_sizeof_tle(void);
size_t orc_sizeof_sat(void);

void orc_get_grav(int which, orc_grav *out);

int orc_tle_parse_lines(const char *line1, const char *line2, orc_tle *out);
/* first two lines of >= 69 chars (Tle.zig L32-47) */
int orc_tle_parse(const char *text, orc_tle *out);
/* count / extract '1'...'2' pairs (Tle.zig MultiIterator L103-132); returns number found (<= max) */
size_t orc_tle_parse_multi(const char *text, orc_tle *out, size_t max);

double orc_year_doy_to_jd(int full_year, double doy);

/* classify + init: near-earth or deep-space; returns ORC_OK / ORC_ERR_* */
int orc_sat_init(const orc_tle *tle, int grav, orc_sat *out);
/* Sgp4.init semantics: deep-space orbit -> ORC_ERR_DEEP_SPACE (Sgp4.zig L120-123) */
int orc_sgp4_init_only(const orc_tle *tle, int grav, orc_sat *out);

double orc_sat_field(const orc_sat *s, const char *name);

void orc_carry_init(const orc_sat *s, orc_carry *c);
int orc_sat_propagate(const orc_sat *s, double tsince, double r[3], double v[3]);
int orc_sat_propagate_carry(const orc_sat *s, double tsince, orc_carry *c, double r[3], double v[3]);

double orc_gstime(double jdut1);
double orc_julian_to_gmst(double jd);
void orc_eci_to_ecef(const double eci[3], double sin_g, double cos_g, double out[3]);
void orc_ecef_to_geodetic(const double ecef[3], double lla[3]);

/*
 * Whole-constellation driver.  tsince(s,t) = times_min[t] + offsets_min[s].
 * Output index: sat-major  (s*n_times + t)*3 ; time-major (t*stride + s)*3.
 * err (optional) is (n_sats, n_times) row-major regardless of layout.
 * mask (optional): sats with mask[s]==0 are not written.
 * nthreads<=1 -> serial; otherwise OpenMP threads over satellites.
 */
void orc_propagate_constellation(const orc_sat *sats, size_t n_sats,
                                 const double *times_min, size_t n_times,
                                 const double *offsets_min,
                                 double *pos, double *vel,
                                 int output_mode, double reference_jd,
                                 const uint8_t *mask, int layout, size_t stride,
                                 uint8_t *err, int nthreads);

int orc_max_threads(void);

/* Conjunction screening: single-target minimum distances (Constellation.zig L683-756)
 * and the all-vs-all cell-list screen over satellite-major positions (conjunction.zig L11-150). */
void orc_screen_target(const orc_sat *sats, size_t n_sats, const double *times_min, size_t n_times,
                       const double *offsets_min, size_t target, double threshold_km, double reference_jd,
                       const uint8_t *failed /* per satellite: init failed, never propagated; may be NULL */,
                       double *out_min_dist, uint32_t *out_min_t);
size_t orc_coarse_screen(const double *positions, size_t num_sats, size_t num_times, double threshold_km,
                         const uint8_t *valid_mask, uint32_t *out_pairs, uint32_t *out_t, size_t max_results);

/* astroz_batch8.c: CPU baseline in the reference's multithreaded SIMD design (8 satellites per
 * vector, polynomial sincos/atan2; NOT the parity oracle: ~1e-7 rad atan2 like the reference's). */
size_t orc_batch8_propagate(const orc_sat *sats, size_t n_sats, const double *times_min, size_t n_times,
                            const double *offsets_min, double *pos, double *vel, int layout, size_t stride,
                            int nthreads);

#ifdef __cplusplus
}
#endif
#endif
