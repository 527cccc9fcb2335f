/*
 * A plain-C host of libastroz_hip.so: what a C (or Zig, via @cImport / extern) maintainer would write against
 * include/astroz_hip.h.  Compiled by the tests with `gcc -std=c99 -pedantic -Wall -Werror` -- the header has to be
 * valid C, not just C++ -- and linked against the shared library directly (no dlopen, no Python in between).
 *
 *   client parse  <line1> <line2>          text -> the 16 TLE fields                       (no GPU needed)
 *   client c_api  <line1> <line2> <t0> <dt> <n>    tle_parse / sgp4_init / sgp4_propagate_batch, the reference's
 *                                          c_api call sequence (src/c_api/root.zig L13-81)    (GPU)
 *   client batch  <tle-file> <t0> <dt> <n> azh_constellation_from_tle_text + azh_propagate_host, satellite-major
 *                                          TEME, positions and velocities                      (GPU)
 * Output: one number per token, %.17g, so that the calling test can compare with the oracle exactly.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "astroz_hip.h"

static int fail(const char *what, int rc)
{
    fprintf(stderr, "%s failed: rc=%d (%s)\n", what, rc, azh_last_error());
    return 3;
}

static int cmd_parse(const char *l1, const char *l2)
{
    double f[16];
    int i;
    const int rc = azh_parse_tle_lines(l1, l2, f);
    if (rc != AZ_OK) return fail("azh_parse_tle_lines", rc);
    for (i = 0; i < 16; ++i) printf("%.17g\n", f[i]);
    return 0;
}

static int cmd_c_api(const char *l1, const char *l2, double t0, double dt, unsigned n)
{
    char *text;
    void *tle = NULL, *sat = NULL;
    double *times, *res, pos[3], vel[3];
    unsigned i;
    int rc;

    astroz_init();
    text = (char *)malloc(strlen(l1) + strlen(l2) + 2);
    sprintf(text, "%s\n%s", l1, l2);
    rc = tle_parse(text, &tle);
    free(text);
    if (rc != AZ_OK) return fail("tle_parse", rc);
    printf("%u\n%.17g\n%.17g\n%.17g\n%.17g\n", tle_get_satellite_number(tle), tle_get_epoch(tle), tle_get_inclination(tle),
           tle_get_eccentricity(tle), tle_get_mean_motion(tle));
    rc = sgp4_init(tle, AZ_WGS72, &sat);
    if (rc != AZ_OK) return fail("sgp4_init", rc);

    rc = sgp4_propagate(sat, t0, pos, vel);
    if (rc != AZ_OK) return fail("sgp4_propagate", rc);
    printf("%.17g %.17g %.17g %.17g %.17g %.17g\n", pos[0], pos[1], pos[2], vel[0], vel[1], vel[2]);

    times = (double *)malloc(sizeof(double) * n);
    res = (double *)malloc(sizeof(double) * 6 * n);
    for (i = 0; i < n; ++i) times[i] = t0 + dt * i;
    rc = sgp4_propagate_batch(sat, times, res, n);
    if (rc != AZ_OK) return fail("sgp4_propagate_batch", rc);
    for (i = 0; i < n; ++i)
        printf("%.17g %.17g %.17g %.17g %.17g %.17g\n", res[6 * i], res[6 * i + 1], res[6 * i + 2], res[6 * i + 3],
               res[6 * i + 4], res[6 * i + 5]);
    free(times);
    free(res);
    sgp4_free(sat);
    tle_free(tle);
    astroz_deinit();
    return 0;
}

static int cmd_batch(const char *path, double t0, double dt, size_t n_times)
{
    FILE *f = fopen(path, "rb");
    long len;
    char *text;
    azh_constellation *c = NULL;
    size_t n_sats, s, t;
    double *times, *pos, *vel;
    uint8_t *err;
    int rc;

    if (!f) { perror(path); return 2; }
    fseek(f, 0, SEEK_END);
    len = ftell(f);
    fseek(f, 0, SEEK_SET);
    text = (char *)malloc((size_t)len + 1);
    if (fread(text, 1, (size_t)len, f) != (size_t)len) { fclose(f); return 2; }
    fclose(f);

    rc = azh_constellation_from_tle_text(text, (size_t)len, AZ_WGS72, 0, &c);
    free(text);
    if (rc != AZ_OK) return fail("azh_constellation_from_tle_text", rc);
    n_sats = azh_num_satellites(c);
    printf("%lu %lu %lu\n", (unsigned long)n_sats, (unsigned long)azh_num_sgp4(c), (unsigned long)azh_num_sdp4(c));

    times = (double *)malloc(sizeof(double) * n_times);
    pos = (double *)malloc(sizeof(double) * 3 * n_sats * n_times);
    vel = (double *)malloc(sizeof(double) * 3 * n_sats * n_times);
    err = (uint8_t *)malloc(n_sats * n_times);
    for (t = 0; t < n_times; ++t) times[t] = t0 + dt * (double)t;
    rc = azh_propagate_host(c, times, n_times, NULL, pos, vel, AZ_OUT_TEME, 0.0, NULL, AZ_LAYOUT_SAT_MAJOR, 0, err);
    if (rc != AZ_OK) return fail("azh_propagate_host", rc);
    for (s = 0; s < n_sats; ++s)
        for (t = 0; t < n_times; ++t) {
            const double *p = pos + (s * n_times + t) * 3, *v = vel + (s * n_times + t) * 3;
            printf("%u %.17g %.17g %.17g %.17g %.17g %.17g\n", (unsigned)err[s * n_times + t], p[0], p[1], p[2], v[0],
                   v[1], v[2]);
        }
    free(times); free(pos); free(vel); free(err);
    azh_constellation_free(c);
    return 0;
}

int main(int argc, char **argv)
{
    if (argc == 4 && !strcmp(argv[1], "parse")) return cmd_parse(argv[2], argv[3]);
    if (argc == 7 && !strcmp(argv[1], "c_api"))
        return cmd_c_api(argv[2], argv[3], atof(argv[4]), atof(argv[5]), (unsigned)atoi(argv[6]));
    if (argc == 6 && !strcmp(argv[1], "batch"))
        return cmd_batch(argv[2], atof(argv[3]), atof(argv[4]), (size_t)atol(argv[5]));
    fprintf(stderr, "usage: client parse|c_api|batch ... (see the header comment)\n");
    return 2;
}
