//! extern declarations of libastroz_hip.so (include/astroz_hip.h) for a Zig host.
//! Not compiled in the build image (no Zig toolchain); kept in sync with the C header by hand.
pub const Handle = opaque {};

pub const Err = enum(i32) {
    ok = 0,
    bad_tle_length = -1,
    bad_checksum = -2,
    deep_space_not_supported = -10,
    invalid_eccentricity = -11,
    satellite_decayed = -12,
    value_error = -20,
    alloc_failed = -100,
    null_pointer = -101,
    not_initialized = -102,
    hip = -200,
    unknown = -999,
};

pub extern "c" fn azh_device_count() c_int;
pub extern "c" fn azh_last_error() [*:0]const u8;
pub extern "c" fn azh_parse_tle_lines(line1: [*:0]const u8, line2: [*:0]const u8, out16: [*]f64) i32;
pub extern "c" fn azh_constellation_from_tle_text(text: [*]const u8, len: usize, grav: i32, device: i32, out: *?*Handle) i32;
pub extern "c" fn azh_constellation_from_tle_lines(line1: [*]const [*:0]const u8, line2: [*]const [*:0]const u8, n: usize, grav: i32, device: i32, out: *?*Handle) i32;
pub extern "c" fn azh_constellation_from_elements(n: usize, epoch_jd: [*]const f64, mm_revday: [*]const f64, ecc: [*]const f64, incl_deg: [*]const f64, raan_deg: [*]const f64, argp_deg: [*]const f64, ma_deg: [*]const f64, bstar: [*]const f64, grav: i32, device: i32, out: *?*Handle) i32;
pub extern "c" fn azh_constellation_free(h: ?*Handle) void;
pub extern "c" fn azh_num_satellites(h: ?*const Handle) usize;
pub extern "c" fn azh_num_sgp4(h: ?*const Handle) usize;
pub extern "c" fn azh_num_sdp4(h: ?*const Handle) usize;
pub extern "c" fn azh_get_epochs(h: ?*const Handle, out: [*]f64) i32;
pub extern "c" fn azh_get_status(h: ?*const Handle, err: ?[*]u8, is_deep: ?[*]u8, irez: ?[*]u8) i32;
pub extern "c" fn azh_get_field(h: ?*const Handle, name: [*:0]const u8, out: [*]f64) i32;
pub extern "c" fn azh_propagate_host(h: ?*Handle, times_min: [*]const f64, n_times: usize, epoch_offsets_min: ?[*]const f64, pos: [*]f64, vel: ?[*]f64, output_mode: i32, reference_jd: f64, sat_mask: ?[*]const u8, layout: i32, out_stride_sats: usize, err: ?[*]u8) i32;
pub extern "c" fn azh_propagate_device(h: ?*Handle, times_min: [*]const f64, n_times: usize, epoch_offsets_min: ?[*]const f64, d_pos: [*]f64, d_vel: ?[*]f64, output_mode: i32, reference_jd: f64, sat_mask: ?[*]const u8, layout: i32, out_stride_sats: usize, d_err: ?[*]u8, stream: ?*anyopaque) i32;
pub extern "c" fn azh_propagate_device_cached(h: ?*Handle, d_pos: [*]f64, d_vel: ?[*]f64, layout: i32, out_stride_sats: usize, d_err: ?[*]u8, stream: ?*anyopaque) i32;
pub extern "c" fn azh_propagate_jd_host(h: ?*Handle, jd: [*]const f64, fr: [*]const f64, n_times: usize, pos: [*]f64, vel: ?[*]f64, output_mode: i32, layout: i32, err: ?[*]u8) i32;
pub extern "c" fn azh_propagate_one_host(h: ?*Handle, sat_index: usize, tsince_min: [*]const f64, n: usize, pos: [*]f64, vel: ?[*]f64, err: ?[*]u8) i32;
pub extern "c" fn azh_synchronize(h: ?*Handle) i32;
pub extern "c" fn azh_set_time_tile(h: ?*Handle, sgp4_tile: u32, sdp4_tile: u32) i32;
pub extern "c" fn azh_set_timing(h: ?*Handle, enabled: i32) i32;
pub extern "c" fn azh_last_kernel_ms(h: ?*Handle) f64;

// fp32 outputs (config 5)
pub extern "c" fn azh_propagate_device_f32(h: ?*Handle, times_min: [*]const f64, n_times: usize, epoch_offsets_min: ?[*]const f64,
    d_pos: [*]f32, d_vel: ?[*]f32, output_mode: i32, reference_jd: f64, sat_mask: ?[*]const u8, layout: i32,
    out_stride_sats: usize, d_err: ?[*]u8, stream: ?*anyopaque) i32;
pub extern "c" fn azh_propagate_device_cached_f32(h: ?*Handle, d_pos: [*]f32, d_vel: ?[*]f32, layout: i32,
    out_stride_sats: usize, d_err: ?[*]u8, stream: ?*anyopaque) i32;

// conjunction screening: Constellation.screenConstellation (src/Constellation.zig L683-756) and
// coarseScreen (bindings/python/src/conjunction.zig L11-150)
pub extern "c" fn azh_screen_target_host(h: ?*Handle, times_min: [*]const f64, n_times: usize, epoch_offsets_min: ?[*]const f64,
    target_index: usize, threshold_km: f64, reference_jd: f64, min_dist_km: [*]f64, min_t_index: [*]u32) i32;
pub extern "c" fn azh_screen_target_device(h: ?*Handle, times_min: [*]const f64, n_times: usize, epoch_offsets_min: ?[*]const f64,
    target_index: usize, threshold_km: f64, reference_jd: f64, d_min_dist_km: [*]f64, d_min_t_index: [*]u32, stream: ?*anyopaque) i32;
pub extern "c" fn azh_coarse_screen_device(d_pos: [*]const f64, n_sats: usize, n_times: usize, layout: i32, stride_sats: usize,
    threshold_km: f64, valid_mask: ?[*]const u8, out_pairs: [*]u32, out_t_index: [*]u32, max_results: usize, n_found: *usize,
    stream: ?*anyopaque) i32;
pub extern "c" fn azh_coarse_screen_host(pos: [*]const f64, n_sats: usize, n_times: usize, layout: i32, stride_sats: usize,
    threshold_km: f64, valid_mask: ?[*]const u8, out_pairs: [*]u32, out_t_index: [*]u32, max_results: usize, n_found: *usize,
    device: i32) i32;
pub extern "c" fn azh_screen_all_host(h: ?*Handle, times_min: [*]const f64, n_times: usize, epoch_offsets_min: ?[*]const f64,
    threshold_km: f64, out_pairs: [*]u32, out_t_index: [*]u32, max_results: usize, n_found: *usize) i32;
