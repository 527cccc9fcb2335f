// fields.h -- device-resident satellite element table (structure-of-arrays in HBM).
//
// One row per field, one column per satellite: elem[field * n_pad + sat].  Lane = satellite, so a
// wave's load of one field is 64 consecutive doubles (512 B, fully coalesced).  The rows are the
// per-satellite state the reference keeps in Sgp4Batch.BatchElements (src/Sgp4Batch.zig L15-75)
// and Sdp4Batch.Sdp4BatchElements (src/Sdp4Batch.zig L16-125), minus the splatted gravity
// constants (kernel arguments here) and plus two folded products (bstar*cc4, bstar*cc5).
#pragma once

// X(name)
#define AZ_SGP4_FIELDS(X)                                                                          \
    X(epoch_jd) X(no_kozai) X(ecco) X(inclo) X(nodeo) X(argpo) X(mo) X(bstar)                      \
    X(no_unkozai) X(a) X(sinio) X(cosio) X(con41) X(x1mth2) X(x7thm1)                              \
    X(mdot) X(argpdot) X(nodedot)                                                                  \
    X(cc1) X(bc4) X(bc5) X(t2cof) X(omgcof) X(xnodcf) X(xlcof) X(xmcof) X(aycof) X(eta)            \
    X(delmo) X(sinmao) X(d2) X(d3) X(d4) X(t3cof) X(t4cof) X(t5cof) X(a_base) X(sqrt_a_base)

#define AZ_DEEP_FIELDS(X)                                                                          \
    X(se2) X(se3) X(si2) X(si3) X(sl2) X(sl3) X(sl4) X(sgh2) X(sgh3) X(sgh4) X(sh2) X(sh3)         \
    X(ee2) X(e3) X(xi2) X(xi3) X(xl2) X(xl3) X(xl4) X(xgh2) X(xgh3) X(xgh4) X(xh2) X(xh3)          \
    X(zmol) X(zmos) X(dedt) X(didt) X(dmdt) X(domdt) X(dnodt)                                      \
    X(d2201) X(d2211) X(d3210) X(d3222) X(d4410) X(d4422) X(d5220) X(d5232) X(d5421) X(d5433)      \
    X(del1) X(del2) X(del3) X(xlamo) X(xfact) X(gsto)

enum AzField {
#define X(n) F_##n,
    AZ_SGP4_FIELDS(X) AZ_DEEP_FIELDS(X)
#undef X
    AZ_NUM_FIELDS
};

// per-satellite status word written by the init kernel
//  bits 0-7  init error code (0 ok, 1 eccentricity, 6 decayed)   [shared.zig L40-47 numbering]
//  bit  8    deep-space (period > 225 min, handled by the SDP4 kernel)
//  bit  9    isimp (perigee < 220 km: simplified drag)
//  bits 10-11 irez (0 none, 1 synchronous, 2 half-day)
#define AZ_FLAG_ERR(f) ((f) & 0xff)
#define AZ_FLAG_DEEP (1u << 8)
#define AZ_FLAG_ISIMP (1u << 9)
#define AZ_FLAG_IREZ(f) (((f) >> 10) & 3u)
//  bits 12-13 eccentricity class of near-earth satellites (0: e < 0.0025 [el2 < 1.6e-5: the near-circular
//             Kepler path], 1: e < 0.0075, 2: e < 0.1, 3: rest);
//             classes 1-2 need more Kepler-Newton trips / wider rotation tiers; the host groups
//             them inside each workgroup so they do not drag whole waves through the slow path
#define AZ_FLAG_ECLASS(f) (((f) >> 12) & 3u)

// raw TLE-unit inputs to the init kernel: in[k * n_pad + sat]
enum AzRawField { R_epoch_jd, R_mm_revday, R_ecc, R_incl_deg, R_raan_deg, R_argp_deg, R_ma_deg, R_bstar, AZ_NUM_RAW };

struct AzGrav {
    double radius_km, j2, j4, xke, j3oj2, vkmpersec, half_j2;
};
