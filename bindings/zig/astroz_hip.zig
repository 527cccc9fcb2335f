//! astroz_hip.zig -- the FFI shim a maintainer of ATTron/astroz adds as src/hip.zig: every entry point of
//! include/astroz_hip.h as a Zig `extern "c"` declaration, plus the AZ_ERR_* -> Zig error mapping.
//! Link with `-lastroz_hip`.  Zig is not installed in this repository's build image, so this file is not compiled
//! here; tests/test_host_cpu.py::test_zig_shim_matches_header checks it against the header (every exported name,
//! its parameter count), and the same call sequences are compiled and run from C (tests/c_client/client.c).
//! Reference interfaces replaced: src/Constellation.zig L245-308, L541-679; src/c_api/root.zig L13-81;
//! src/dispatch.zig L18-54.

/// include/astroz_hip.h AZ_ERR_* (= src/c_api/error.zig L3-19) -> the error sets the reference's callers expect
/// (Sgp4.Error / Sdp4.Error members where they exist, src/Sgp4.zig L26-31).
pub const HipError = error{
    BadTleLength,
    BadChecksum,
    DeepSpaceNotSupported,
    InvalidEccentricity,
    SatelliteDecayed,
    ValueError,
    OutOfMemory,
    NullPointer,
    NotInitialized,
    /// AZ_ERR_HIP: no device, a failed launch or copy (text: azh_last_error())
    DeviceFailure,
    Unknown,
};

pub fn check(rc: i32) HipError!void {
    return switch (rc) {
        0 => {},
        -1 => HipError.BadTleLength,
        -2 => HipError.BadChecksum,
        -10 => HipError.DeepSpaceNotSupported,
        -11 => HipError.InvalidEccentricity,
        -12 => HipError.SatelliteDecayed,
        -20 => HipError.ValueError,
        -100 => HipError.OutOfMemory,
        -101 => HipError.NullPointer,
        -102 => HipError.NotInitialized,
        -200 => HipError.DeviceFailure,
        else => HipError.Unknown,
    };
}

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
pub extern "c" fn azh_screen_track_device(h: ?*Handle, times_min: [*]const f64, n_times: usize, epoch_offsets_min: ?[*]const f64,
    d_track: [*]const f64, exclude_index: usize, threshold_km: f64, d_min_dist_km: [*]f64, d_min_t_index: [*]u32, stream: ?*anyopaque) i32; // the same against an external track (another shard's satellite)
pub extern "c" fn azh_coarse_screen_device(d_pos: [*]const f64, n_sats: usize, n_times: usize, layout: i32, stride_sats: usize,
    threshold_km: f64, valid_mask: ?[*]const u8, out_pairs: [*]u32, out_t_index: [*]u32, max_results: usize, n_found: *usize,
    stream: ?*anyopaque) i32;
pub extern "c" fn azh_coarse_screen_host(pos: [*]const f64, n_sats: usize, n_times: usize, layout: i32, stride_sats: usize,
    threshold_km: f64, valid_mask: ?[*]const u8, out_pairs: [*]u32, out_t_index: [*]u32, max_results: usize, n_found: *usize,
    device: i32) i32;
pub extern "c" fn azh_screen_all_host(h: ?*Handle, times_min: [*]const f64, n_times: usize, epoch_offsets_min: ?[*]const f64,
    threshold_km: f64, out_pairs: [*]u32, out_t_index: [*]u32, max_results: usize, n_found: *usize) i32;

// ---- round 2 ----
// text front ends (Tle.MultiIterator / parseOmm / parseOmmArray, src/Tle.zig L103-238)
pub extern "c" fn azh_parse_tle_text(text: [*]const u8, len: usize, out16: [*]f64, max_records: usize, n_found: *usize) i32;
pub extern "c" fn azh_parse_omm_json(text: [*]const u8, len: usize, out16: [*]f64, max_records: usize, n_found: *usize) i32;
pub extern "c" fn azh_set_parse_threads(n: i32) void;
pub extern "c" fn azh_constellation_from_omm_json(text: [*]const u8, len: usize, grav: i32, device: i32, out: *?*Handle) i32;
pub extern "c" fn azh_constellation_subset(h: ?*const Handle, indices: [*]const u32, n: usize, device: i32, out: *?*Handle) i32;

// row windows (chunked multi-GPU pipelines), arithmetic / path switches, device-pointer one-satellite call
pub extern "c" fn azh_propagate_device_window(h: ?*Handle, row_lo: usize, row_hi: usize, d_pos: [*]f64, d_vel: ?[*]f64, layout: i32,
    out_stride_sats: usize, d_err: ?[*]u8, stream: ?*anyopaque) i32;
pub extern "c" fn azh_set_f32_mode(h: ?*Handle, mode: i32) i32; // 0 mixed precision (default), 1 packed fp32, 2 fp64 rounded at the store
pub extern "c" fn azh_set_f32_arithmetic(h: ?*Handle, enabled: i32) i32; // boolean: 0 = fp64 rounded at the store, else packed fp32
pub extern "c" fn azh_set_fast_path(h: ?*Handle, enabled: i32) i32;
pub extern "c" fn azh_set_tile_kernel(h: ?*Handle, enabled: i32) i32;
pub extern "c" fn azh_set_graphs(h: ?*Handle, enabled: i32) i32; // hipGraph replay of repeated cached-input launch sets (default off; pays for multi-window pipelines)
pub extern "c" fn azh_set_host_points(n: usize) void; // one-satellite calls of at most n points run the library's step on the calling thread (default 128; deep-space members: half)
pub extern "c" fn azh_get_host_points() usize;
pub extern "c" fn azh_set_host_copy_threads(n: i32) void; // host threads behind the pinned staging of host-returning copies (-1 auto, 0 = direct pageable copies)
// result arrays the DMA engines write directly (pinned, pooled inside the library): a Zig host allocates `positions` / `velocities`
// here instead of from its allocator, and azh_propagate_host lands in them at the link rate (no staging hop)
pub extern "c" fn azh_host_alloc(bytes: usize, out: *?*anyopaque) i32;
pub extern "c" fn azh_host_free(p: ?*anyopaque) void;
pub extern "c" fn azh_host_pool_stats(live_bytes: ?*usize, free_bytes: ?*usize) void;
pub extern "c" fn azh_host_pool_trim() void;
pub extern "c" fn azh_last_path(h: ?*const Handle) u32; // AZH_PATH_* bits: which kernel families the last call launched
pub extern "c" fn azh_last_one_stats(h: ?*Handle, n_segments: ?*u32, n_handed_over: ?*u32) i32; // last one-satellite call: fast segments / handed over
pub extern "c" fn azh_propagate_one_device(h: ?*Handle, sat_index: usize, d_tsince_min: [*]const f64, n: usize, d_pos: [*]f64,
    d_vel: ?[*]f64, d_err: ?[*]u8, stream: ?*anyopaque) i32;
pub extern "c" fn azh_selftest_math(x: [*]const f64, n: usize, out6n: [*]f64, device: i32) i32;
pub extern "c" fn azh_selftest_host_step(el: [*]const f64, n_pad: usize, sat: usize, flags: u32, grav: i32, tsince_min: [*]const f64, n: usize,
    out6n: [*]f64, err: ?[*]u8) i32; // the host route's step on a caller-supplied element table (no device needed; KAT)
pub extern "c" fn azh_selftest_coords(op: i32, in: *const [4]f64, out: *[5]f64) i32; // part (A)'s closed forms through the kernels' frame code (KAT)

// src/c_api/coords.zig
pub extern "c" fn coords_julian_to_gmst(jd: f64) f64;
pub extern "c" fn coords_eci_to_ecef(eci: *const [3]f64, gmst: f64, ecef: *[3]f64) void;
pub extern "c" fn coords_ecef_to_geodetic(ecef: *const [3]f64, lla: *[3]f64) void;

// one process, several devices: replaces the std.Thread fan-out of Constellation.propagateConstellation
// (src/Constellation.zig L557-603)
pub const Group = opaque {};
pub extern "c" fn azh_group_create_from_tle_text(text: [*]const u8, len: usize, grav: i32, devices: [*]const i32, n_devices: i32,
    n_chunks: i32, out: *?*Group) i32;
pub extern "c" fn azh_group_create_from_omm_json(text: [*]const u8, len: usize, grav: i32, devices: [*]const i32, n_devices: i32,
    n_chunks: i32, out: *?*Group) i32;
pub extern "c" fn azh_group_free(g: ?*Group) void;
pub extern "c" fn azh_group_num_satellites(g: ?*const Group) usize;
pub extern "c" fn azh_group_num_devices(g: ?*const Group) i32;
pub extern "c" fn azh_group_padded_rows(g: ?*const Group) usize;
pub extern "c" fn azh_group_get_epochs(g: ?*const Group, out: [*]f64) i32;
pub extern "c" fn azh_group_propagate_host(g: ?*Group, times_min: [*]const f64, n_times: usize, epoch_offsets_min: ?[*]const f64,
    n_offsets: usize, pos: [*]f64, vel: ?[*]f64, output_mode: i32, reference_jd: f64, err: ?[*]u8) i32;
pub extern "c" fn azh_group_propagate_allgather(g: ?*Group, times_min: [*]const f64, n_times: usize,
    epoch_offsets_min: ?[*]const f64, n_offsets: usize, d_pos: [*]const [*]f64, d_vel: ?[*]const [*]f64) i32;
// the fused single-target screen over a group (Constellation.screenConstellation, src/Constellation.zig L683-756): every device
// screens its own rows against the target's track, no collective
pub extern "c" fn azh_group_screen_target_host(g: ?*Group, times_min: [*]const f64, n_times: usize, epoch_offsets_min: ?[*]const f64,
    n_offsets: usize, target_index: usize, threshold_km: f64, reference_jd: f64, min_dist_km: [*]f64, min_t_index: [*]u32) i32;
pub extern "c" fn azh_group_screen_target_device(g: ?*Group, times_min: [*]const f64, n_times: usize, epoch_offsets_min: ?[*]const f64,
    n_offsets: usize, target_index: usize, threshold_km: f64, reference_jd: f64, d_min_dist_km: [*]const [*]f64, d_min_t_index: [*]const [*]u32) i32;
pub extern "c" fn azh_group_shard_size(g: ?*const Group, device_slot: i32) usize;
pub extern "c" fn azh_group_shard_rows(g: ?*const Group, device_slot: i32, out_rows: [*]u32) i32;
pub extern "c" fn azh_group_synchronize(g: ?*Group) i32;
