/*
 * astroz_hip.h -- C ABI of libastroz_hip.so, the MI355X-native drop-in for astroz's batched
 * SGP4/SDP4 constellation-propagation path.
 *
 * Plain C: pointers, sizes and scalars only.  Every entry point names the reference interface
 * it replaces (ATTron/astroz v0.12.0, file:line).  All floating-point work happens in HIP
 * kernels on the GPU; there is NO CPU fallback -- without a usable device every compute entry
 * point returns AZ_ERR_HIP.
 *
 * Two layers:
 *  (A) the reference's own `c_api` surface (src/c_api/root.zig L13-81), same names, argument
 *      meaning and error codes, so an existing dlopen/FFI client of libastroz_c.so relinks
 *      unchanged;
 *  (B) the coarse-grained constellation boundary (one call = all satellites x all times) that
 *      mirrors Constellation.zig's propagateConstellation / propagateSdp4Constellation /
 *      Constellation.propagate and what bindings/python/src/{satrec,sgp4}.zig hand to them.
 *      The reference's fine-grained plugin ABI (src/simdKernels.zig L9-29: 8 satellites x 1
 *      time per call through a function pointer) is a CPU-SIMD artefact that cannot drive a GPU
 *      usefully; the boundary therefore sits one level up, at its only caller
 *      (src/Constellation.zig L413-476).
 */
#ifndef ASTROZ_HIP_H
#define ASTROZ_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- error codes: src/c_api/error.zig L3-19 (+ AZ_ERR_HIP for device/runtime failures) ---- */
enum {
    AZ_OK = 0,
    AZ_ERR_BAD_TLE_LENGTH = -1,
    AZ_ERR_BAD_CHECKSUM = -2,
    AZ_ERR_DEEP_SPACE_NOT_SUPPORTED = -10,
    AZ_ERR_INVALID_ECCENTRICITY = -11,
    AZ_ERR_SATELLITE_DECAYED = -12,
    AZ_ERR_VALUE = -20,
    AZ_ERR_ALLOC_FAILED = -100,
    AZ_ERR_NULL_POINTER = -101,
    AZ_ERR_NOT_INITIALIZED = -102,
    AZ_ERR_HIP = -200,
    AZ_ERR_UNKNOWN = -999
};

/* gravity model ids: bindings/python/src/shared.zig L21-27, src/c_api/sgp4.zig L17-20 */
enum { AZ_WGS84 = 0, AZ_WGS72 = 1 };
/* src/Constellation.zig L30-34 */
enum { AZ_OUT_TEME = 0, AZ_OUT_ECEF = 1, AZ_OUT_GEODETIC = 2 };
/* src/Constellation.zig L37-42 */
enum { AZ_LAYOUT_SAT_MAJOR = 0, AZ_LAYOUT_TIME_MAJOR = 1 };

/* =====================================================================================
 * (A) reference c_api surface -- src/c_api/root.zig
 * ===================================================================================== */

/* root.zig L13-15: (major<<16)|(minor<<8)|patch of the C API (0.3.0) */
uint32_t astroz_version(void);
/* root.zig L17-23: allocator hooks, no-ops in the reference; here astroz_init() may be used to
 * fail early when no GPU is present (it never aborts; compute calls report AZ_ERR_HIP). */
void astroz_init(void);
void astroz_deinit(void);

/* root.zig L25-45 / src/c_api/tle.zig L12-54.  `str` holds the two 69-column lines. */
int32_t tle_parse(const char *str, void **out_handle);
void tle_free(void *handle);
uint32_t tle_get_satellite_number(void *handle);
double tle_get_epoch(void *handle); /* seconds since J2000 */
double tle_get_inclination(void *handle);   /* degrees  */
double tle_get_eccentricity(void *handle);
double tle_get_mean_motion(void *handle);   /* rev/day  */

/* root.zig L47-58 / src/c_api/sgp4.zig L18-89.  grav: 0 WGS84, 1 WGS72.  Deep-space element
 * sets are rejected with AZ_ERR_DEEP_SPACE_NOT_SUPPORTED exactly as the reference does. */
int32_t sgp4_init(void *tle_handle, int32_t grav, void **out_handle);
void sgp4_free(void *handle);
int32_t sgp4_propagate(void *handle, double tsince_min, double pos[3], double vel[3]);
/* results: count x [x,y,z,vx,vy,vz].  One kernel launch, lane = time. */
int32_t sgp4_propagate_batch(void *handle, const double *times_min, double *results, uint32_t count);

/* root.zig L73-81 / src/c_api/coordinates.zig: the output-mode math of this path (WorldCoordinateSystem.zig
 * L87-154) as scalar calls.  gmst in radians; lla = (lat deg, lon deg, alt km), degrees like the reference's
 * ecefToGeodeticDeg.  Pure host functions, as in the reference (no device needed); azh_selftest_coords evaluates the
 * same quantities through the kernels' own frame code on the device. */
double coords_julian_to_gmst(double jd);
void coords_eci_to_ecef(const double eci[3], double gmst, double ecef[3]);
void coords_ecef_to_geodetic(const double ecef[3], double lla[3]);

/* root.zig L60-71 / src/c_api/orbital_mechanics.zig: four closed-form scalars (src/calculations.zig L83-125) that are not on
 * the propagation path but belong to the reference's C surface, so that a client of libastroz_c.so links unchanged.  Same
 * argument checks and return conventions: orbital_hohmann -> AZ_ERR_VALUE for r <= 0 or |r1 - r2| < 1000; the scalar
 * functions return -1.0 for an invalid radius / semi-major axis; orbital_velocity(sma = 0) is the circular speed.
 * Pure host functions like coords_*. */
typedef struct azh_hohmann_result {
    double semi_major_axis, delta_v1, delta_v2, total_delta_v, transfer_time, transfer_time_days;
} azh_hohmann_result; /* = HohmannResult, orbital_mechanics.zig L9-16 */
int32_t orbital_hohmann(double mu, double r1, double r2, azh_hohmann_result *out);
double orbital_velocity(double mu, double radius, double sma);
double orbital_period(double mu, double sma);
double orbital_escape_velocity(double mu, double radius);

/* =====================================================================================
 * (B) constellation boundary
 * ===================================================================================== */

typedef struct azh_constellation azh_constellation;

/* Every numeric field of one TLE (src/Tle.zig L49-101), for hosts that want the Satrec getters
 * of bindings/python/src/satrec.zig L390-430 without a second parser.  out16 receives:
 * [0] satnum [1] epoch year (2 digits) [2] epoch day [3] epoch JD [4] ndot (TLE units)
 * [5] bstar [6] incl deg [7] raan deg [8] ecc [9] argp deg [10] mean anomaly deg
 * [11] mean motion rev/day [12] element set no [13] rev no [14] classification char [15] 0 */
int32_t azh_parse_tle_lines(const char *line1, const char *line2, double *out16);

/* the same 16 fields for every element set of a text: multi-TLE text (2- or 3-line, Tle.MultiIterator,
 * src/Tle.zig L103-132) or OMM JSON (object or array, src/Tle.zig L134-238).  Host-side text handling only --
 * no GPU needed.  *n_found = records in the text; at most max_records are written to out16 (16 doubles each). */
int32_t azh_parse_tle_text(const char *text, size_t len, double *out16, size_t max_records, size_t *n_found);
int32_t azh_parse_omm_json(const char *text, size_t len, double *out16, size_t max_records, size_t *n_found);
/* Text ingest at catalog scale (the reference's from_tle_text, bindings/python/src/sgp4.zig L293-350, reads
 * serially): text beyond a few hundred KB is cut at record boundaries and parsed by several host threads, results in
 * file order.  n = threads to use (0 = automatic: the host's cores, at most 16; 1 = serial).  Process-wide. */
void azh_set_parse_threads(int32_t n);

/* number of visible HIP devices (0 when there is no GPU / no driver) */
int azh_device_count(void);
/* last HIP error string of the calling thread's most recent failing call ("" if none) */
const char *azh_last_error(void);

/*
 * Build a device-resident constellation.  Replaces Constellation.init (src/Constellation.zig
 * L101-200), shared.buildBatches (bindings/python/src/shared.zig L62-97) and
 * Sgp4Constellation.from_tle_text (bindings/python/src/sgp4.zig L293-350): parses the text on
 * the host, uploads the raw elements, and runs the init kernel (SGP4 + deep-space init and the
 * near-earth / deep-space classification, src/Sgp4.zig L108-180, src/Sdp4.zig L174-274).
 * Satellites keep their catalog order; output rows are indexed by that order.
 * `device` is the HIP device ordinal (one process per GPU: pass LOCAL_RANK).
 */
int32_t azh_constellation_from_tle_text(const char *text, size_t len, int32_t grav, int32_t device,
                                        azh_constellation **out);
/* OMM JSON (one object or an array, CelesTrak's FORMAT=JSON): Tle.parseOmm / parseOmmArray
 * (src/Tle.zig L134-238).  Elements keep their full JSON precision.  AZ_ERR_BAD_TLE_LENGTH for an EPOCH
 * shorter than 19 characters or an empty array, AZ_ERR_VALUE for anything that is not OMM JSON. */
int32_t azh_constellation_from_omm_json(const char *text, size_t len, int32_t grav, int32_t device,
                                        azh_constellation **out);
/* n pairs of NUL-terminated lines */
int32_t azh_constellation_from_tle_lines(const char *const *line1, const char *const *line2, size_t n,
                                         int32_t grav, int32_t device, azh_constellation **out);
/* element arrays in TLE units (deg, rev/day), each of length n: for hosts that already hold
 * parsed elements (a Zig host's []Tle, OMM records, synthetic catalogs of 10^6 members) */
int32_t azh_constellation_from_elements(size_t n, const double *epoch_jd, const double *mean_motion_revday,
                                        const double *ecc, const double *incl_deg, const double *raan_deg,
                                        const double *argp_deg, const double *mean_anom_deg,
                                        const double *bstar, int32_t grav, int32_t device,
                                        azh_constellation **out);
/* a new constellation holding members indices[0..n) of an existing one, in that order, on `device`
 * (-1 = the same device): the shard of a multi-GPU job, or the reference's near-earth-first ordering
 * (bindings/python/astroz/__init__.py L374-393) without re-parsing any text */
int32_t azh_constellation_subset(const azh_constellation *c, const uint32_t *indices, size_t n, int32_t device,
                                 azh_constellation **out);
void azh_constellation_free(azh_constellation *c);

size_t azh_num_satellites(const azh_constellation *c);
size_t azh_num_sgp4(const azh_constellation *c);  /* Constellation.numSgp4, L82 */
size_t azh_num_sdp4(const azh_constellation *c);  /* Constellation.numSdp4, L89 */
/* per-satellite epoch JD (the `epochs` getter, bindings/python/src/sgp4.zig L283-291) */
int32_t azh_get_epochs(const azh_constellation *c, double *out_n);
/* per-satellite init status: python-sgp4 error code (0, 1 eccentricity, 6 decayed;
 * bindings/python/src/shared.zig L40-47), deep-space flag, resonance class irez (0,1,2) */
int32_t azh_get_status(const azh_constellation *c, uint8_t *err_n, uint8_t *is_deep_n, uint8_t *irez_n);
/* one row of the element table by name ("a", "no_unkozai", "mdot", "gsto", ... see fields.h):
 * backs the Satrec getters (bindings/python/src/satrec.zig L390-496) and the init parity tests */
int32_t azh_get_field(const azh_constellation *c, const char *name, double *out_n);

/*
 * Propagate every satellite to every time.  Replaces propagateConstellation
 * (src/Constellation.zig L541-605) AND propagateSdp4Constellation (L611-679): near-earth rows run
 * in the SGP4 kernel, deep-space rows in the SDP4 kernel (own stream, concurrently), both write
 * straight into the caller's layout at the satellite's catalog index.
 *   tsince(s,t) = times_min[t] + epoch_offsets_min[s]        (Constellation.zig L423-426)
 *   epoch_offsets_min == NULL  -> zeros (each satellite relative to its own epoch)
 *   vel == NULL                -> positions only
 *   output_mode                -> AZ_OUT_*; ECEF/geodetic use GMST(reference_jd + t/1440)
 *   sat_mask (n_sats bytes)    -> rows with mask==0 are left untouched
 *   layout / out_stride_sats   -> index (s*n_times+t)*3 or (t*stride+s)*3 (L46-51); stride 0 = n_sats
 *   err (n_sats x n_times, optional) -> per-(satellite,time) python-sgp4 codes; failing rows are
 *                                 zero-filled for that satellite only (scalar-path semantics)
 * The *_host form takes host pointers (copies in/out, synchronous, like the reference's call);
 * the *_device form takes device pointers for pos/vel/err and a hipStream_t (NULL = the
 * constellation's own stream), is asynchronous, and is what keeps results resident in HBM.
 */
int32_t azh_propagate_host(azh_constellation *c, const double *times_min, size_t n_times,
                           const double *epoch_offsets_min, double *pos, double *vel, int32_t output_mode,
                           double reference_jd, const uint8_t *sat_mask, int32_t layout,
                           size_t out_stride_sats, uint8_t *err);
int32_t azh_propagate_device(azh_constellation *c, const double *times_min /*host*/, size_t n_times,
                             const double *epoch_offsets_min /*host*/, double *d_pos, double *d_vel,
                             int32_t output_mode, double reference_jd, const uint8_t *sat_mask /*host*/,
                             int32_t layout, size_t out_stride_sats, uint8_t *d_err, void *stream);
/* same launch with times/offsets already uploaded by a previous call (no host work at all):
 * the steady-state form used when the same grid is propagated repeatedly */
int32_t azh_propagate_device_cached(azh_constellation *c, double *d_pos, double *d_vel, int32_t layout,
                                    size_t out_stride_sats, uint8_t *d_err, void *stream);
/* the cached launch restricted to the satellites row_lo <= index < row_hi (other rows are left untouched):
 * lets a multi-GPU host pipeline its shard in chunks -- chunk k+1 is computed while chunk k is in the
 * all-gather (astroz_amd/distributed.py).  No reference counterpart (single process). */
int32_t azh_propagate_device_window(azh_constellation *c, size_t row_lo, size_t row_hi, double *d_pos, double *d_vel,
                                    int32_t layout, size_t out_stride_sats, uint8_t *d_err, void *stream);
/* fp32 OUTPUT variants (BASELINE config 5: 1M satellites x 10,000 steps would be 480 GB in fp64); d_pos/d_vel
 * are float arrays of the same shapes.  No reference counterpart (astroz is fp64 only).  Arithmetic, azh_set_f32_mode:
 *   mode 0 (default)  mixed precision for near-circular members on uniform grids, satellite-major TEME
 *                     (astroz_amd/csrc/fast_step_f32.h, az_sgp4_fast_step_f32p): every O(1) quantity -- the along-track
 *                     phase pair and the rotation applied to it, node and inclination pairs, orientation, radius, speed --
 *                     in fp64 per grid point, everything small in packed fp32 for two grid points at once.  Within 0.6 m /
 *                     0.6 mm/s per component of the fp64 result (measured 0.35 m / 0.39 mm/s over 10,000-minute spans):
 *                     the level of fp32 storage itself (half an ulp is 0.25 m / 0.24 mm/s per component in LEO) and inside
 *                     the reference's own SIMD-vs-scalar bar of 1 mm/s (src/Sgp4Batch.zig L186-187).  Everything else
 *                     (eccentric and deep-space members, ECEF / geodetic, irregular grids) as mode 2.
 *   mode 1            opt-in: packed fp32 arithmetic with fp64 phase and radius chains for the same members -- 1.15x the
 *                     rate of mode 0, positions within 4 m and velocities within 6 mm/s (measured 2.4 m / 4.0 mm/s).
 *   mode 2            fp64 arithmetic throughout, every component rounded ONCE when it is stored: float32(fp64 result).
 * Returns AZ_ERR_VALUE for any other mode.
 * azh_set_f32_arithmetic keeps the BOOLEAN meaning it was introduced with (round 2): 0 = fp64 arithmetic rounded once at the
 * store (= mode 2), non-zero = packed fp32 arithmetic (= mode 1).  (Round 3 had overloaded it with the mode numbers above, so
 * that a caller passing 0 to pin float32(fp64 result) silently got the mixed step: the modes now have their own entry point.) */
#define AZH_F32_MIXED 0
#define AZH_F32_PACKED 1
#define AZH_F32_FP64_ROUNDED 2
int32_t azh_set_f32_mode(azh_constellation *c, int32_t mode);
int32_t azh_set_f32_arithmetic(azh_constellation *c, int32_t enabled);
int32_t azh_propagate_device_f32(azh_constellation *c, const double *times_min, size_t n_times,
                                 const double *epoch_offsets_min, float *d_pos, float *d_vel, int32_t output_mode,
                                 double reference_jd, const uint8_t *sat_mask, int32_t layout,
                                 size_t out_stride_sats, uint8_t *d_err, void *stream);
int32_t azh_propagate_device_cached_f32(azh_constellation *c, float *d_pos, float *d_vel, int32_t layout,
                                        size_t out_stride_sats, uint8_t *d_err, void *stream);

/* Fused single-target conjunction screen = Constellation.screenConstellation
 * (src/Constellation.zig L683-756; Python: Sgp4Constellation.screen_conjunction,
 * bindings/python/astroz/__init__.py L625-632).  For every satellite the minimum distance (km) to
 * satellite `target_index` over the grid and the grid index where it occurs; start value
 * `threshold_km` / index 0 (a satellite that never comes closer, the target itself and failed
 * members report exactly that); earliest index among equal minima.  Propagation and reduction
 * are one kernel: no positions are written.  `reference_jd` is accepted for signature parity (the
 * reference rotates to ECEF before differencing, which does not change a distance).  Deep-space
 * members are screened too (the reference's routine covers SGP4 batches only). */
int32_t azh_screen_target_host(azh_constellation *c, const double *times_min, size_t n_times,
                               const double *epoch_offsets_min, size_t target_index, double threshold_km,
                               double reference_jd, double *min_dist_km, uint32_t *min_t_index);
int32_t azh_screen_target_device(azh_constellation *c, const double *times_min, size_t n_times,
                                 const double *epoch_offsets_min, size_t target_index, double threshold_km,
                                 double reference_jd, double *d_min_dist_km, uint32_t *d_min_t_index, void *stream);

/* The same screen against an EXTERNAL track: d_track = n_times x 3 doubles (TEME km) on c's device -- the track of a satellite
 * that lives in another shard of a multi-GPU run (every rank screens ITS rows against the one target: SURVEY 8e's
 * gather-free consumer; astroz_amd.distributed.sharded_screen_target), or of an object that is in no catalog.  exclude_index:
 * a member of c that reports threshold / 0 like the target does above (SIZE_MAX: none).  Non-finite track points (a target
 * whose propagation failed there; azh_screen_target_* writes NaN for them too) never compare closer than the threshold. */
int32_t azh_screen_track_device(azh_constellation *c, const double *times_min, size_t n_times,
                                const double *epoch_offsets_min, const double *d_track, size_t exclude_index,
                                double threshold_km, double *d_min_dist_km, uint32_t *d_min_t_index, void *stream);

/* All-vs-all coarse screen = coarseScreen (bindings/python/src/conjunction.zig L11-150): every pair
 * (s < other) closer than threshold_km at grid index t, found with a per-step cell list (cell edge =
 * threshold).  Positions: fp64, either layout; rows with valid_mask[s] == 0 or a non-finite x are
 * skipped (L53-66).  Results are written to HOST arrays sorted by (t, s, other); at most
 * max_results (the first in that order), *n_found = number written.
 *   _device: positions already in HBM (e.g. from azh_propagate_device)
 *   _host  : positions in host memory (copied to `device` first)
 *   azh_screen_all_host: propagate to TEME on the device and screen, positions never leave HBM
 *                        (= screen(..., target=None), __init__.py L633-658) */
int32_t azh_coarse_screen_device(const double *d_pos, size_t n_sats, size_t n_times, int32_t layout,
                                 size_t stride_sats, double threshold_km, const uint8_t *valid_mask,
                                 uint32_t *out_pairs, uint32_t *out_t_index, size_t max_results, size_t *n_found,
                                 void *stream);
int32_t azh_coarse_screen_host(const double *pos, size_t n_sats, size_t n_times, int32_t layout, size_t stride_sats,
                               double threshold_km, const uint8_t *valid_mask, uint32_t *out_pairs,
                               uint32_t *out_t_index, size_t max_results, size_t *n_found, int32_t device);
int32_t azh_screen_all_host(azh_constellation *c, const double *times_min, size_t n_times,
                            const double *epoch_offsets_min, double threshold_km, uint32_t *out_pairs,
                            uint32_t *out_t_index, size_t max_results, size_t *n_found);

/* ---- several GPUs behind one handle (one process; a Zig / C host has no torch.distributed) -----------------
 * Block-cyclic satellite shards over `devices`: the catalog is cut into n_chunks super-blocks of
 * n_devices * rows consecutive satellites (rows a multiple of 64), device d owns rows [d*rows, (d+1)*rows) of each
 * (the plan of astroz_amd/distributed.py).  No reference counterpart (src/Constellation.zig L327-385 is
 * one process on one host).  Satellite-major outputs in CATALOG order:
 *   azh_group_propagate_host       every device propagates its shard and copies its blocks straight into the caller's
 *                                  (n_sats, n_times, 3) host arrays over its own PCIe link: no collective;
 *   azh_group_propagate_allgather  the full TEME arrays resident on EVERY device: d_pos[i] / d_vel[i] are device
 *                                  buffers on devices[i] of azh_group_padded_rows() x n_times x 3 doubles (rows beyond
 *                                  n_sats are padding and arrive as zeros); RCCL all-gathers over xGMI (librccl is
 *                                  loaded on first use), chunk k in flight while chunk k+1 is propagated.  Synchronous.
 * epoch_offsets_min (both calls): NULL, or n_offsets doubles indexed by catalog row (the length of the reference's
 * epochOffsets slice, src/Constellation.zig L541-552); n_offsets < azh_group_num_satellites() is AZ_ERR_VALUE. */
typedef struct azh_group azh_group;
int32_t azh_group_create_from_tle_text(const char *text, size_t len, int32_t grav, const int32_t *devices,
                                       int32_t n_devices, int32_t n_chunks, azh_group **out);
int32_t azh_group_create_from_omm_json(const char *text, size_t len, int32_t grav, const int32_t *devices,
                                       int32_t n_devices, int32_t n_chunks, azh_group **out);
void azh_group_free(azh_group *g);
size_t azh_group_num_satellites(const azh_group *g);
int32_t azh_group_num_devices(const azh_group *g);
size_t azh_group_padded_rows(const azh_group *g);
int32_t azh_group_get_epochs(const azh_group *g, double *out_n);
int32_t azh_group_propagate_host(azh_group *g, const double *times_min, size_t n_times, const double *epoch_offsets_min,
                                 size_t n_offsets, double *pos, double *vel, int32_t output_mode, double reference_jd,
                                 uint8_t *err);
int32_t azh_group_propagate_allgather(azh_group *g, const double *times_min, size_t n_times,
                                      const double *epoch_offsets_min, size_t n_offsets, double *const *d_pos,
                                      double *const *d_vel);

/* The fused single-target screen over a group (Constellation.screenConstellation, src/Constellation.zig L683-756, one
 * process, N devices): every device screens the rows it owns against the target's track -- computed once on the device that
 * owns `target_index` (a catalog row) and handed to the others, n_times x 24 bytes -- with no collective: the one multi-GPU
 * workload on this path whose time falls with N (the gathered propagation's does not).
 *   _host   : min_dist_km / min_t_index are host arrays of azh_group_num_satellites() entries in CATALOG order; synchronous.
 *   _device : d_min_dist_km[i] / d_min_t_index[i] are device buffers on devices[i] of azh_group_shard_size(g, i) entries in
 *             that shard's local order (= ascending catalog rows, azh_group_shard_rows); asynchronous on every shard's own
 *             stream -- azh_group_synchronize waits for all of them.
 * Values as azh_screen_target_host: start value threshold_km / index 0; the target itself and failed members report that. */
int32_t azh_group_screen_target_host(azh_group *g, const double *times_min, size_t n_times, const double *epoch_offsets_min,
                                     size_t n_offsets, size_t target_index, double threshold_km, double reference_jd,
                                     double *min_dist_km, uint32_t *min_t_index);
int32_t azh_group_screen_target_device(azh_group *g, const double *times_min, size_t n_times, const double *epoch_offsets_min,
                                       size_t n_offsets, size_t target_index, double threshold_km, double reference_jd,
                                       double *const *d_min_dist_km, uint32_t *const *d_min_t_index);
size_t azh_group_shard_size(const azh_group *g, int32_t device_slot);
int32_t azh_group_shard_rows(const azh_group *g, int32_t device_slot, uint32_t *out_rows);
int32_t azh_group_synchronize(azh_group *g);

/* Constellation.propagate (src/Constellation.zig L245-308): absolute times jd[t]+fr[t]; the
 * reference epoch is that of the first near-earth member (the first entry of the reference's SGP4 batch list,
 * L139-140; the first member's if there is none). Host pointers. */
int32_t azh_propagate_jd_host(azh_constellation *c, const double *jd, const double *fr, size_t n_times,
                              double *pos, double *vel, int32_t output_mode, int32_t layout, uint8_t *err);
int32_t azh_synchronize(azh_constellation *c);

/* one satellite x many times (lane = time): Satrec.sgp4 / sgp4_array
 * (bindings/python/src/satrec.zig L169-201, L256-343) and dispatch.sgp4Times8 / sdp4Times8
 * (src/dispatch.zig L32-44).  tsince in minutes from that satellite's epoch, in any order and spacing.
 * pos/vel: n x 3 each (host); err: n bytes (optional).  Up to 16,384 points the kernel exchanges the data with a pinned
 * buffer of the handle itself (one launch, one synchronize: ~20 us a call); from 2^20 points of a near-earth member every
 * wave checks its own 1,024 times and runs the branch-free step where they are (quasi-)uniform (azh_last_one_stats). */
int32_t azh_propagate_one_host(azh_constellation *c, size_t sat_index, const double *tsince_min, size_t n,
                               double *pos, double *vel, uint8_t *err);

/* the same with device pointers (tsince, pos, vel, err all in HBM; asynchronous on `stream`, NULL = the
 * constellation's own): the kernel rate without the PCIe copies */
int32_t azh_propagate_one_device(azh_constellation *c, size_t sat_index, const double *d_tsince_min, size_t n,
                                 double *d_pos, double *d_vel, uint8_t *d_err, void *stream);
/* known-answer hook for the kernels' element math ON THE DEVICE (devmath.h: az_sincos, az_rcp, az_rsqrt,
 * az_rotate; the reference's KATs are src/simdMath.zig L214-286).  x: n host values; out6n (host):
 * [0,n) sin x, [n,2n) cos x, [2n,3n) x*rcp(x), [3n,4n) |x|*rsqrt(|x|)^2, [4n,6n) (sin,cos)(0.7321 + x) obtained
 * by rotating (sin,cos)(0.7321).  Test infrastructure; not part of the reference surface. */
int32_t azh_selftest_math(const double *x, size_t n, double *out6n, int32_t device);
/* the seven closed-form scalars of part (A) (coords_*, orbital_*) evaluated ON THE DEVICE through the kernels' own frame code
 * (known-answer tests; part (A)'s functions themselves are host closed forms).  op 0: GMST of in[0]; 1: ECI in[0..2] -> ECEF
 * at GMST in[3]; 2: ECEF -> (lat deg, lon deg, alt km); 3: (velocity, period, escape velocity) of (mu, radius, sma);
 * 4: Hohmann (sma, dv1, dv2, |dv1| + |dv2|, transfer time) of (mu, r1, r2).  Unused outputs are 0. */
int32_t azh_selftest_coords(int32_t op, const double in[4], double out[5]);
/* the host route's step (azh_set_host_points; astroz_amd/csrc/host_step.h) on a caller-supplied element table
 * el[field * n_pad + sat] (the 85 rows of astroz_amd/csrc/fields.h, in that order) with status word `flags`: n points, out6n = n x
 * (x, y, z, vx, vy, vz), err (n, optional).  Needs no device: lets the CPU test tier hold the shipped object to its reference checker.
 * Test infrastructure; not part of the reference surface. */
int32_t azh_selftest_host_step(const double *el, size_t n_pad, size_t sat, uint32_t flags, int32_t grav, const double *tsince_min,
                               size_t n, double *out6n, uint8_t *err);

/* tuning knobs (kernel time-tile length; 0 = automatic).  Not part of the reference surface. */
int32_t azh_set_time_tile(azh_constellation *c, uint32_t sgp4_tile, uint32_t sdp4_tile);
/* enable (default) / disable the branch-free uniform-grid step (astroz_amd/csrc/fast_step.h).  Results
 * of the two paths agree to rounding; the switch exists so that tests can compare them. */
int32_t azh_set_fast_path(azh_constellation *c, int32_t enabled);
/* time-major output on (quasi-)uniform grids: 1 (default) = the 16-satellite tile kernel (k_tiles_fast: lane = time
 * arithmetic, transposed through LDS); 2 = the branch-free lane = satellite kernel (k_cols_fast: one wave = 64 catalog rows,
 * 1,536-byte runs; measured slower than the tiles, kept as an option); 0 = neither (the generic lane = satellite kernel).
 * Results agree to rounding; the switch exists so that tests and benchmarks can compare them. */
int32_t azh_set_tile_kernel(azh_constellation *c, int32_t enabled);
/* hipGraph replay of the cached-input launch sets (azh_propagate_device_cached / _cached_f32 / _window): a launch set is three
 * to nine runtime calls (kernels on up to three streams, fork / join events, a memset); the second call with the same
 * outputs, layout, stride, row window and stream is captured and every later one is ONE hipGraphLaunch.  OFF by default
 * (environment ASTROZ_AMD_GRAPHS=1 switches it on for new handles): measured on MI355X / ROCm 7.0 it pays where one step is
 * several launch sets -- the chunked row-window pipeline of a sharded run: 0.40 -> 0.33 ms for four windows of config 2 --
 * and costs 1-4 % on a single launch set (profiles/r05_experiments.txt B).  The cache is dropped whenever new inputs are
 * staged or a switch of the handle changes.  Not used while azh_set_timing is on (its event pair sits inside the set). */
int32_t azh_set_graphs(azh_constellation *c, int32_t enabled);
/* host-returning calls (azh_propagate_host, azh_propagate_jd_host, azh_group_propagate_host): results travel device -> pinned
 * staging slots (kept in the handle) -> the caller's arrays, the second hop by n host threads while the next chunk is on the
 * link.  A direct copy into FRESH pageable arrays pays the runtime's first-time pinning of the range (config 2: 55 ms instead
 * of the link's 17).  -1 automatic (default: 6), 0 = direct pageable copies. */
void azh_set_host_copy_threads(int32_t n);
/* One-satellite host-pointer calls of at most n points (azh_propagate_one_host, sgp4_propagate, sgp4_propagate_batch, and
 * azh_propagate_host on a one-satellite handle; default 128 -- half of it for deep-space members --, environment ASTROZ_AMD_HOST_POINTS, 0 = never) do not launch a
 * kernel: the library evaluates its own per-point step (the source k_one_satellite compiles, astroz_amd/csrc/host_step.h) on
 * the calling thread, from the element column the DEVICE initialised -- what the reference's scalar call is
 * (bindings/python/src/satrec.zig L169-201: 0.4 us; a launch + a synchronize is 20 us).  Results agree with the kernels' to
 * rounding (tests/test_gpu_round6.py: 1e-9 km / 1e-12 km/s); azh_last_path reports AZH_PATH_HOST_STEP.  Process-wide. */
void azh_set_host_points(size_t n);
size_t azh_get_host_points(void);
/* Result arrays the DMA engines can write: pinned host memory from a pool inside the library.  A host-returning call
 * (azh_propagate_host, azh_propagate_jd_host, azh_group_propagate_host, azh_propagate_one_host) whose pos / vel arrays were
 * allocated here skips the staging hop: the device-to-host copy lands in them at the link rate (config 2's 932 MB: 16.4 ms
 * on PCIe Gen5 x16, against 20-25 ms through the staging slots into fresh pageable arrays).  azh_host_free returns a block
 * to the pool (it stays pinned for the next result of that size; at most ASTROZ_AMD_HOST_POOL_MB -- default 4,096 -- of free
 * blocks are kept); azh_host_pool_trim releases the free blocks.  The reference's Python layer allocates its results itself
 * per call (numpy.empty, bindings/python/astroz/api.py L304-314); astroz_amd.api does the same from this pool.
 * Any thread; blocks are 2-MiB multiples.  AZ_ERR_ALLOC_FAILED when the host cannot pin the memory. */
int32_t azh_host_alloc(size_t bytes, void **out);
void azh_host_free(void *p);
void azh_host_pool_stats(size_t *live_bytes, size_t *free_bytes);
void azh_host_pool_trim(void);
/* enable (default) / disable the hipEvent pair recorded around every propagate call; disabling it
 * removes two event records per call from tight replay loops (azh_last_kernel_ms then returns -1) */
int32_t azh_set_timing(azh_constellation *c, int32_t enabled);
/* elapsed GPU milliseconds of the most recent propagate call's kernels (hipEvent on the
 * launch stream; valid after azh_synchronize) */
double azh_last_kernel_ms(azh_constellation *c);
/* which kernel families the most recent propagate / screen call of this handle launched (bit mask): lets a caller -- and the
 * tests -- see that a grid took the branch-free kernels.  A grid counts as uniform when times[i] = times[0] + i*step within
 * 4e-6 min: exactly uniform grids, and the quasi-uniform ones the reference's own (jd, fr) arithmetic produces
 * (times = ((jd + fr) - reference_jd) * 1440 is quantised to 2^-31 day = 6.7e-7 min; api.py L300-302, Constellation.zig
 * L266-269), which run the same kernels with a first-order correction of every point to its actual time. */
#define AZH_PATH_ROWS_FAST 1u     /* k_rows_fast / k_rows_fast32: lane = time, branch-free, satellite-major rows (or the screen) */
#define AZH_PATH_TILES_FAST 2u    /* k_tiles_fast: lane = time, branch-free, 16-row time-major tiles */
#define AZH_PATH_ROWS_GENERIC 4u  /* k_rows over the whole near-earth list: lane = time, any grid */
#define AZH_PATH_LANE_SAT 8u      /* k_propagate: lane = satellite (short grids; time-major with masks / fp32 / irregular grids) */
#define AZH_PATH_DEEP_ROWS 16u    /* k_rows_deep: lane = time deep-space rows */
#define AZH_PATH_COLS_FAST 64u     /* k_cols_fast: lane = satellite, branch-free, time-major runs of 64 catalog rows */
#define AZH_PATH_QUASI_UNIFORM 32u /* the staged grid is quasi-uniform: the fast kernels ran in their DELTA form */
#define AZH_PATH_HOST_STEP 128u   /* no kernel: a one-satellite call of at most azh_get_host_points() points, evaluated on the calling thread */
uint32_t azh_last_path(const azh_constellation *c);
/* the most recent one-satellite call of this handle (azh_propagate_one_host / _device; sgp4_propagate / _batch through their
 * handles): the number of 1,024-point segments the branch-free kernel was launched on (0: the series was short, interleaved,
 * deep-space or failed -- one generic step per point throughout) and how many of them it handed over to the generic kernel
 * (irregular times, a window outside the fast step's bounds).  Waits for that call to finish.  Returns an astroz error code. */
int32_t azh_last_one_stats(azh_constellation *c, uint32_t *n_segments, uint32_t *n_handed_over);

#ifdef __cplusplus
}
#endif
#endif
