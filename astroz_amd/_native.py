"""ctypes binding of libastroz_hip.so (include/astroz_hip.h).

This is the ONLY compute path of the package: there is no NumPy / CPU fallback.  If the shared
library is missing, or no MI355X is visible, the calls raise.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# ASTROZ_AMD_LIB selects an alternative build of the SAME library (kernel tuning sweeps)
LIB_PATH = os.environ.get("ASTROZ_AMD_LIB") or os.path.join(_HERE, "libastroz_hip.so")

WGS84, WGS72 = 0, 1
OUT_TEME, OUT_ECEF, OUT_GEODETIC = 0, 1, 2
SAT_MAJOR, TIME_MAJOR = 0, 1
OUTPUT_MODES = {"teme": OUT_TEME, "ecef": OUT_ECEF, "geodetic": OUT_GEODETIC}

AZ_ERR_HIP = -200
# azh_last_path bits (include/astroz_hip.h)
PATH_ROWS_FAST, PATH_TILES_FAST, PATH_ROWS_GENERIC, PATH_LANE_SAT, PATH_DEEP_ROWS, PATH_QUASI_UNIFORM, PATH_COLS_FAST, PATH_HOST_STEP = 1, 2, 4, 8, 16, 32, 64, 128

# every symbol include/astroz_hip.h declares (tests check the library exports all of them)
EXPORTS = [
    "astroz_version", "astroz_init", "astroz_deinit", "tle_parse", "tle_free",
    "tle_get_satellite_number", "tle_get_epoch", "tle_get_inclination", "tle_get_eccentricity",
    "tle_get_mean_motion", "sgp4_init", "sgp4_free", "sgp4_propagate", "sgp4_propagate_batch",
    "azh_device_count", "azh_last_error", "azh_parse_tle_lines", "azh_constellation_from_tle_text",
    "azh_constellation_from_tle_lines", "azh_constellation_from_elements", "azh_constellation_subset", "azh_constellation_free",
    "azh_num_satellites", "azh_num_sgp4", "azh_num_sdp4", "azh_get_epochs", "azh_get_status",
    "azh_get_field", "azh_propagate_host", "azh_propagate_device", "azh_propagate_device_cached", "azh_propagate_device_window",
    "azh_propagate_jd_host", "azh_synchronize", "azh_propagate_one_host", "azh_set_time_tile", "azh_set_timing", "azh_set_fast_path", "azh_set_tile_kernel", "azh_set_graphs", "azh_set_f32_arithmetic", "azh_set_f32_mode",
    "azh_last_kernel_ms", "azh_last_path", "azh_last_one_stats", "azh_set_host_copy_threads", "azh_set_host_points", "azh_get_host_points", "azh_selftest_coords", "azh_selftest_host_step", "azh_host_alloc", "azh_host_free", "azh_host_pool_stats", "azh_host_pool_trim", "azh_propagate_device_f32", "azh_propagate_device_cached_f32",
    "azh_screen_target_host", "azh_screen_target_device", "azh_coarse_screen_device", "azh_coarse_screen_host",
    "azh_screen_all_host", "azh_constellation_from_omm_json", "azh_propagate_one_device", "azh_selftest_math",
    "azh_parse_tle_text", "azh_parse_omm_json", "azh_set_parse_threads", "coords_julian_to_gmst",
    "azh_group_create_from_tle_text", "azh_group_create_from_omm_json", "azh_group_free", "azh_group_num_satellites",
    "azh_group_num_devices", "azh_group_padded_rows", "azh_group_get_epochs", "azh_group_propagate_host",
    "azh_group_propagate_allgather", "azh_group_screen_target_host", "azh_group_screen_target_device", "azh_group_shard_size",
    "azh_group_shard_rows", "azh_group_synchronize", "azh_screen_track_device", "coords_eci_to_ecef", "coords_ecef_to_geodetic",
    "orbital_hohmann", "orbital_velocity", "orbital_period", "orbital_escape_velocity",
]


class NativeError(RuntimeError):
    def __init__(self, code, what):
        self.code = code
        super().__init__("%s failed with code %d%s" % (what, code, _detail(code)))


_lib = None


def _detail(code):
    if _lib is not None and code == AZ_ERR_HIP:
        msg = _lib.azh_last_error().decode(errors="replace")
        return " (%s)" % msg if msg else " (HIP runtime error: is an MI355X visible?)"
    return ""


def _share_hip_runtime_with_torch():
    """One HIP runtime per process.  PyTorch-ROCm wheels ship their own libamdhip64 / libhsa-runtime64
    and ask for them by unversioned file name, so a process that loads /opt/rocm's runtime first (through
    this library) and imports torch afterwards ends up with two ROCr instances, and the second one
    finds no device.  When torch is installed, bind this library to torch's copy (same SONAME,
    libamdhip64.so.7) -- device pointers and streams are then shared with torch tensors in either
    import order.  ASTROZ_AMD_SYSTEM_HIP=1 keeps /opt/rocm's runtime (processes that never import torch)."""
    if os.environ.get("ASTROZ_AMD_SYSTEM_HIP"):
        return
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(cand):
        C.CDLL(cand, mode=C.RTLD_GLOBAL)


def lib():
    """Load libastroz_hip.so.  Fails loudly if the HIP extension has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "astroz_amd: %s not found -- build it with `python -c 'import __graft_entry__ as g; "
            "g.build()'` (hipcc --offload-arch=gfx950). There is no CPU fallback." % LIB_PATH)
    _share_hip_runtime_with_torch()
    L = C.CDLL(LIB_PATH)
    vp, sz, dbl, i32, u32 = C.c_void_p, C.c_size_t, C.c_double, C.c_int32, C.c_uint32
    L.astroz_version.restype = u32
    L.tle_parse.argtypes = [C.c_char_p, C.POINTER(vp)]
    L.tle_parse.restype = i32
    L.tle_free.argtypes = [vp]
    L.tle_get_satellite_number.argtypes = [vp]
    L.tle_get_satellite_number.restype = u32
    for f in ("tle_get_epoch", "tle_get_inclination", "tle_get_eccentricity", "tle_get_mean_motion"):
        getattr(L, f).argtypes = [vp]
        getattr(L, f).restype = dbl
    L.sgp4_init.argtypes = [vp, i32, C.POINTER(vp)]
    L.sgp4_init.restype = i32
    L.sgp4_free.argtypes = [vp]
    L.sgp4_propagate.argtypes = [vp, dbl, vp, vp]
    L.sgp4_propagate.restype = i32
    L.sgp4_propagate_batch.argtypes = [vp, vp, vp, u32]
    L.sgp4_propagate_batch.restype = i32
    L.azh_device_count.restype = C.c_int
    L.azh_last_error.restype = C.c_char_p
    L.azh_parse_tle_lines.argtypes = [C.c_char_p, C.c_char_p, vp]
    L.azh_parse_tle_lines.restype = i32
    L.azh_constellation_from_tle_text.argtypes = [C.c_char_p, sz, i32, i32, C.POINTER(vp)]
    L.azh_constellation_from_tle_text.restype = i32
    for f in ("azh_parse_tle_text", "azh_parse_omm_json"):
        getattr(L, f).argtypes = [C.c_char_p, sz, vp, sz, C.POINTER(sz)]
        getattr(L, f).restype = i32
    L.azh_set_parse_threads.argtypes = [i32]
    L.azh_set_parse_threads.restype = None
    for f in ("azh_group_create_from_tle_text", "azh_group_create_from_omm_json"):
        getattr(L, f).argtypes = [C.c_char_p, sz, i32, vp, i32, i32, C.POINTER(vp)]
        getattr(L, f).restype = i32
    L.azh_group_free.argtypes = [vp]
    L.azh_group_num_satellites.argtypes = [vp]
    L.azh_group_num_satellites.restype = sz
    L.azh_group_num_devices.argtypes = [vp]
    L.azh_group_num_devices.restype = i32
    L.azh_group_padded_rows.argtypes = [vp]
    L.azh_group_padded_rows.restype = sz
    L.azh_group_get_epochs.argtypes = [vp, vp]
    L.azh_group_get_epochs.restype = i32
    L.azh_group_propagate_host.argtypes = [vp, vp, sz, vp, sz, vp, vp, i32, dbl, vp]
    L.azh_group_propagate_host.restype = i32
    L.azh_group_propagate_allgather.argtypes = [vp, vp, sz, vp, sz, vp, vp]
    L.azh_group_propagate_allgather.restype = i32
    L.azh_group_screen_target_host.argtypes = [vp, vp, sz, vp, sz, sz, dbl, dbl, vp, vp]
    L.azh_group_screen_target_host.restype = i32
    L.azh_group_screen_target_device.argtypes = [vp, vp, sz, vp, sz, sz, dbl, dbl, vp, vp]
    L.azh_group_screen_target_device.restype = i32
    L.azh_group_shard_size.argtypes = [vp, i32]
    L.azh_group_shard_size.restype = sz
    L.azh_group_shard_rows.argtypes = [vp, i32, vp]
    L.azh_group_shard_rows.restype = i32
    L.azh_group_synchronize.argtypes = [vp]
    L.azh_group_synchronize.restype = i32
    L.azh_constellation_from_omm_json.argtypes = [C.c_char_p, sz, i32, i32, C.POINTER(vp)]
    L.azh_constellation_from_omm_json.restype = i32
    L.azh_propagate_one_device.argtypes = [vp, sz, vp, sz, vp, vp, vp, vp]
    L.azh_propagate_one_device.restype = i32
    L.azh_selftest_math.argtypes = [vp, sz, vp, i32]
    L.azh_selftest_math.restype = i32
    L.coords_julian_to_gmst.argtypes = [dbl]
    L.coords_julian_to_gmst.restype = dbl
    L.coords_eci_to_ecef.argtypes = [vp, dbl, vp]
    L.coords_eci_to_ecef.restype = None
    L.coords_ecef_to_geodetic.argtypes = [vp, vp]
    L.coords_ecef_to_geodetic.restype = None
    L.azh_constellation_from_tle_lines.argtypes = [vp, vp, sz, i32, i32, C.POINTER(vp)]
    L.azh_constellation_from_tle_lines.restype = i32
    L.azh_constellation_from_elements.argtypes = [sz] + [vp] * 8 + [i32, i32, C.POINTER(vp)]
    L.azh_constellation_from_elements.restype = i32
    L.azh_constellation_subset.argtypes = [vp, vp, sz, i32, C.POINTER(vp)]
    L.azh_constellation_subset.restype = i32
    L.azh_constellation_free.argtypes = [vp]
    for f in ("azh_num_satellites", "azh_num_sgp4", "azh_num_sdp4"):
        getattr(L, f).argtypes = [vp]
        getattr(L, f).restype = sz
    L.azh_get_epochs.argtypes = [vp, vp]
    L.azh_get_epochs.restype = i32
    L.azh_get_status.argtypes = [vp, vp, vp, vp]
    L.azh_get_status.restype = i32
    L.azh_get_field.argtypes = [vp, C.c_char_p, vp]
    L.azh_get_field.restype = i32
    L.azh_propagate_host.argtypes = [vp, vp, sz, vp, vp, vp, i32, dbl, vp, i32, sz, vp]
    L.azh_propagate_host.restype = i32
    L.azh_propagate_device.argtypes = [vp, vp, sz, vp, vp, vp, i32, dbl, vp, i32, sz, vp, vp]
    L.azh_propagate_device.restype = i32
    L.azh_propagate_device_cached.argtypes = [vp, vp, vp, i32, sz, vp, vp]
    L.azh_propagate_device_cached.restype = i32
    L.azh_propagate_device_window.argtypes = [vp, sz, sz, vp, vp, i32, sz, vp, vp]
    L.azh_propagate_device_window.restype = i32
    L.azh_propagate_jd_host.argtypes = [vp, vp, vp, sz, vp, vp, i32, i32, vp]
    L.azh_propagate_jd_host.restype = i32
    L.azh_synchronize.argtypes = [vp]
    L.azh_synchronize.restype = i32
    L.azh_propagate_one_host.argtypes = [vp, sz, vp, sz, vp, vp, vp]
    L.azh_propagate_one_host.restype = i32
    L.azh_set_time_tile.argtypes = [vp, u32, u32]
    L.azh_set_time_tile.restype = i32
    L.azh_set_timing.argtypes = [vp, i32]
    L.azh_set_timing.restype = i32
    L.azh_set_fast_path.argtypes = [vp, i32]
    L.azh_set_fast_path.restype = i32
    L.azh_set_tile_kernel.argtypes = [vp, i32]
    L.azh_set_tile_kernel.restype = i32
    L.azh_set_graphs.argtypes = [vp, i32]
    L.azh_set_graphs.restype = i32
    L.azh_set_f32_arithmetic.argtypes = [vp, i32]
    L.azh_set_f32_arithmetic.restype = i32
    L.azh_set_f32_mode.argtypes = [vp, i32]
    L.azh_set_f32_mode.restype = i32
    L.azh_last_kernel_ms.argtypes = [vp]
    L.azh_last_kernel_ms.restype = dbl
    L.azh_set_host_copy_threads.argtypes = [i32]
    L.azh_set_host_copy_threads.restype = None
    L.azh_set_host_points.argtypes = [sz]
    L.azh_set_host_points.restype = None
    L.azh_get_host_points.argtypes = []
    L.azh_get_host_points.restype = sz
    L.azh_host_alloc.argtypes = [C.c_size_t, C.POINTER(C.c_void_p)]
    L.azh_host_alloc.restype = i32
    L.azh_host_free.argtypes = [C.c_void_p]
    L.azh_host_free.restype = None
    L.azh_host_pool_stats.argtypes = [C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
    L.azh_host_pool_stats.restype = None
    L.azh_host_pool_trim.argtypes = []
    L.azh_host_pool_trim.restype = None
    L.azh_last_path.argtypes = [vp]
    L.azh_last_path.restype = u32
    L.azh_last_one_stats.argtypes = [vp, vp, vp]
    L.azh_last_one_stats.restype = C.c_int32
    L.azh_propagate_device_f32.argtypes = L.azh_propagate_device.argtypes
    L.azh_propagate_device_f32.restype = i32
    L.azh_propagate_device_cached_f32.argtypes = L.azh_propagate_device_cached.argtypes
    L.azh_propagate_device_cached_f32.restype = i32
    L.azh_screen_target_host.argtypes = [vp, vp, sz, vp, sz, dbl, dbl, vp, vp]
    L.azh_screen_target_host.restype = i32
    L.azh_screen_target_device.argtypes = [vp, vp, sz, vp, sz, dbl, dbl, vp, vp, vp]
    L.azh_screen_target_device.restype = i32
    L.azh_screen_track_device.argtypes = [vp, vp, sz, vp, vp, sz, dbl, vp, vp, vp]
    L.azh_screen_track_device.restype = i32
    L.azh_coarse_screen_device.argtypes = [vp, sz, sz, i32, sz, dbl, vp, vp, vp, sz, C.POINTER(sz), vp]
    L.azh_coarse_screen_device.restype = i32
    L.azh_coarse_screen_host.argtypes = [vp, sz, sz, i32, sz, dbl, vp, vp, vp, sz, C.POINTER(sz), i32]
    L.azh_coarse_screen_host.restype = i32
    L.azh_screen_all_host.argtypes = [vp, vp, sz, vp, dbl, vp, vp, sz, C.POINTER(sz)]
    L.azh_screen_all_host.restype = i32
    L.orbital_hohmann.argtypes = [dbl, dbl, dbl, vp]
    L.orbital_hohmann.restype = i32
    for f, n in (("orbital_velocity", 3), ("orbital_period", 2), ("orbital_escape_velocity", 2)):
        getattr(L, f).argtypes = [dbl] * n
        getattr(L, f).restype = dbl
    _lib = L
    return L


def check(code, what):
    if code != 0:
        raise NativeError(code, what)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _ptr(a):
    return None if a is None else a.ctypes.data


class DeviceConstellation:
    """Owner of one azh_constellation handle (device-resident element table)."""

    def __init__(self, handle):
        self._h = handle
        L = lib()
        self.n = L.azh_num_satellites(handle)
        self.n_sgp4 = L.azh_num_sgp4(handle)
        self.n_sdp4 = L.azh_num_sdp4(handle)
        self._epochs = None
        self._status = None

    # -- constructors ------------------------------------------------------------------
    @classmethod
    def from_tle_text(cls, text, grav=WGS72, device=0):
        b = text.encode() if isinstance(text, str) else bytes(text)
        h = C.c_void_p()
        check(lib().azh_constellation_from_tle_text(b, len(b), grav, device, C.byref(h)),
              "azh_constellation_from_tle_text")
        return cls(h)

    @classmethod
    def from_omm_json(cls, text, grav=WGS72, device=0):
        b = text.encode() if isinstance(text, str) else bytes(text)
        h = C.c_void_p()
        check(lib().azh_constellation_from_omm_json(b, len(b), grav, device, C.byref(h)),
              "azh_constellation_from_omm_json")
        return cls(h)

    @classmethod
    def from_tle_lines(cls, pairs, grav=WGS72, device=0):
        n = len(pairs)
        a1 = (C.c_char_p * n)(*[p[0].encode() for p in pairs])
        a2 = (C.c_char_p * n)(*[p[1].encode() for p in pairs])
        h = C.c_void_p()
        check(lib().azh_constellation_from_tle_lines(a1, a2, n, grav, device, C.byref(h)),
              "azh_constellation_from_tle_lines")
        return cls(h)

    @classmethod
    def from_elements(cls, epoch_jd, mm_revday, ecc, incl_deg, raan_deg, argp_deg, ma_deg, bstar,
                      grav=WGS72, device=0):
        cols = [_f64(x) for x in (epoch_jd, mm_revday, ecc, incl_deg, raan_deg, argp_deg, ma_deg, bstar)]
        n = len(cols[0])
        assert all(len(c) == n for c in cols)
        h = C.c_void_p()
        check(lib().azh_constellation_from_elements(n, *[c.ctypes.data for c in cols], grav, device,
                                                    C.byref(h)), "azh_constellation_from_elements")
        return cls(h)

    def subset(self, indices, device=-1):
        """New constellation with members `indices` (in that order); device -1 = same device."""
        idx = np.ascontiguousarray(indices, dtype=np.uint32)
        h = C.c_void_p()
        check(lib().azh_constellation_subset(self._h, idx.ctypes.data, len(idx), device, C.byref(h)),
              "azh_constellation_subset")
        return type(self)(h)

    def handle_address(self):
        """The azh_constellation pointer as an integer (valid while this object lives)."""
        return int(self._h.value)

    def close(self):
        if getattr(self, "_h", None):
            lib().azh_constellation_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- queries -----------------------------------------------------------------------
    @property
    def epochs(self):
        if self._epochs is None:
            e = np.empty(self.n, dtype=np.float64)
            check(lib().azh_get_epochs(self._h, e.ctypes.data), "azh_get_epochs")
            self._epochs = e
        return self._epochs

    @property
    def status(self):
        """(init error code, is_deep, irez) per satellite."""
        if self._status is None:
            e = np.empty(self.n, dtype=np.uint8)
            d = np.empty(self.n, dtype=np.uint8)
            r = np.empty(self.n, dtype=np.uint8)
            check(lib().azh_get_status(self._h, e.ctypes.data, d.ctypes.data, r.ctypes.data), "azh_get_status")
            self._status = (e, d.astype(bool), r)
        return self._status

    def field(self, name):
        out = np.empty(self.n, dtype=np.float64)
        check(lib().azh_get_field(self._h, name.encode(), out.ctypes.data), "azh_get_field(%s)" % name)
        return out

    def set_time_tile(self, sgp4_tile=0, sdp4_tile=0):
        check(lib().azh_set_time_tile(self._h, sgp4_tile, sdp4_tile), "azh_set_time_tile")

    # -- propagation -------------------------------------------------------------------
    def propagate_host(self, times_min, offsets_min=None, *, pos, vel=None, mode=OUT_TEME, reference_jd=0.0,
                       mask=None, layout=TIME_MAJOR, stride=0, err=None):
        times = _f64(times_min)
        off = None if offsets_min is None else _f64(offsets_min)
        if off is not None and len(off) < self.n:
            raise ValueError("epoch_offsets must have at least num_satellites elements")
        m = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8)
        if m is not None and len(m) < self.n:
            raise ValueError("satellite_mask must have at least num_satellites elements")
        rows = (stride or self.n) if layout == TIME_MAJOR else self.n
        need = rows * len(times) * 3 * 8
        for name, arr in (("positions", pos), ("velocities", vel)):
            if arr is None:
                continue
            if arr.dtype != np.float64 or not arr.flags.c_contiguous or not arr.flags.writeable:
                raise ValueError("%s must be a writable C-contiguous float64 array" % name)
            if arr.nbytes < need:
                raise ValueError("%s array too small" % name)
        if err is not None and (err.dtype != np.uint8 or err.nbytes < self.n * len(times)):
            raise ValueError("err array too small")
        check(lib().azh_propagate_host(self._h, times.ctypes.data, len(times), _ptr(off), pos.ctypes.data,
                                       _ptr(vel), mode, float(reference_jd), _ptr(m), layout, stride,
                                       _ptr(err)), "azh_propagate_host")

    def propagate_device(self, times_min, offsets_min, d_pos, d_vel=None, *, mode=OUT_TEME, reference_jd=0.0,
                         mask=None, layout=TIME_MAJOR, stride=0, d_err=None, stream=None, f32=False):
        """d_pos/d_vel/d_err are raw device pointers (e.g. torch.Tensor.data_ptr()); asynchronous.
        f32=True: d_pos/d_vel are float32 arrays; the arithmetic behind them is set_f32_arithmetic's mode (default:
        mixed precision, within 0.6 m / 0.6 mm/s of the fp64 path).  Time-major: `stride` (satellites per time row,
        0 = n) a multiple of 16 puts every tile run on whole cache lines (about 8 % faster)."""
        times = _f64(times_min)
        off = None if offsets_min is None else _f64(offsets_min)
        m = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8)
        fn = lib().azh_propagate_device_f32 if f32 else lib().azh_propagate_device
        check(fn(self._h, times.ctypes.data, len(times), _ptr(off), d_pos, d_vel, mode,
                 float(reference_jd), _ptr(m), layout, stride, d_err, stream), "azh_propagate_device")

    def propagate_device_cached(self, d_pos, d_vel=None, *, layout=TIME_MAJOR, stride=0, d_err=None, stream=None,
                                f32=False):
        fn = lib().azh_propagate_device_cached_f32 if f32 else lib().azh_propagate_device_cached
        check(fn(self._h, d_pos, d_vel, layout, stride, d_err, stream), "azh_propagate_device_cached")

    def propagate_device_window(self, row_lo, row_hi, d_pos, d_vel=None, *, layout=SAT_MAJOR, stride=0, d_err=None,
                                stream=None):
        """Cached launch restricted to satellites [row_lo, row_hi) (chunked multi-GPU pipelines)."""
        check(lib().azh_propagate_device_window(self._h, int(row_lo), int(row_hi), d_pos, d_vel, layout, stride, d_err,
                                                stream), "azh_propagate_device_window")

    # -- conjunction screening ------------------------------------------------------------
    def screen_target(self, times_min, target, threshold=10.0, offsets_min=None, reference_jd=0.0):
        """Fused propagate+screen against satellite `target`: (min_dist km (n,), min_t_index (n,) u32)."""
        times = _f64(times_min)
        off = None if offsets_min is None else _f64(offsets_min)
        if off is not None and len(off) < self.n:
            raise ValueError("epoch_offsets must have at least num_satellites elements")
        if not 0 <= int(target) < self.n:
            raise ValueError("target index out of range")
        d = np.empty(self.n, dtype=np.float64)
        ti = np.empty(self.n, dtype=np.uint32)
        check(lib().azh_screen_target_host(self._h, times.ctypes.data, len(times), _ptr(off), int(target),
                                           float(threshold), float(reference_jd), d.ctypes.data, ti.ctypes.data),
              "azh_screen_target_host")
        return d, ti

    def screen_track_device(self, times_min, d_track, threshold, d_min_dist, d_min_t, offsets_min=None, exclude=None, stream=None):
        """Fused screen of this constellation's rows against an EXTERNAL track (azh_screen_track_device): d_track = raw device
        pointer to (n_times, 3) float64 TEME km; results into device buffers of n entries; asynchronous.  exclude: a member that
        reports threshold / 0 (the target itself, when this shard owns it)."""
        times = _f64(times_min)
        off = None if offsets_min is None else _f64(offsets_min)
        if off is not None and len(off) < self.n:
            raise ValueError("epoch_offsets must have at least num_satellites elements")
        ex = C.c_size_t(-1).value if exclude is None else int(exclude)
        check(lib().azh_screen_track_device(self._h, times.ctypes.data, len(times), _ptr(off), d_track, ex, float(threshold),
                                            d_min_dist, d_min_t, stream), "azh_screen_track_device")

    def screen_all(self, times_min, threshold=10.0, offsets_min=None, max_results=10_000_000):
        """All-vs-all: propagate on the device and screen there: (pairs (k,2) u32, t_index (k,) u32),
        sorted by (t, s, other)."""
        times = _f64(times_min)
        off = None if offsets_min is None else _f64(offsets_min)
        pairs = np.empty((max_results, 2), dtype=np.uint32)
        tt = np.empty(max_results, dtype=np.uint32)
        k = C.c_size_t(0)
        check(lib().azh_screen_all_host(self._h, times.ctypes.data, len(times), _ptr(off), float(threshold),
                                        pairs.ctypes.data, tt.ctypes.data, max_results, C.byref(k)),
              "azh_screen_all_host")
        return pairs[:k.value].copy(), tt[:k.value].copy()

    def propagate_one(self, sat_index, tsince_min):
        t = _f64(np.atleast_1d(tsince_min))
        n = len(t)
        pos = result_empty((n, 3))      # (long series: pinned blocks of the library's pool, the DMA writes them directly)
        vel = result_empty((n, 3))
        err = result_empty((n,), np.uint8)          # (pinned when the series is long: the library writes every byte)
        check(lib().azh_propagate_one_host(self._h, sat_index, t.ctypes.data, n, pos.ctypes.data, vel.ctypes.data,
                                           err.ctypes.data), "azh_propagate_one_host")
        return err, pos, vel

    def propagate_one_device(self, sat_index, d_tsince, n, d_pos, d_vel=None, d_err=None, stream=None):
        """One satellite x n times, all buffers in HBM (raw device pointers); asynchronous."""
        check(lib().azh_propagate_one_device(self._h, sat_index, d_tsince, n, d_pos, d_vel, d_err, stream),
              "azh_propagate_one_device")

    F32_MODES = {"mixed": 0, "packed": 1, "fp64": 2}

    def set_f32_arithmetic(self, mode):
        """Arithmetic behind fp32 outputs (azh_set_f32_mode).  "mixed" / 0 (default): the mixed-precision step where it
        applies (near-circular members on a uniform grid, satellite-major TEME: within 0.6 m / 0.6 mm/s of the fp64 oracle,
        the level of fp32 storage itself), fp64 rounded at the store elsewhere; "packed" / 1: packed fp32 arithmetic where it
        applies (opt-in: 4 m / 6 mm/s); "fp64" / 2: fp64 arithmetic rounded once at the store everywhere (0.5 m / 0.4 mm/s).
        A bool keeps the meaning this setter was introduced with: False = "fp64", True = "packed".  Which kernel runs
        depends on whether the staged grid is uniform, so fp32 results are not bit-stable across grids in the first two modes."""
        if isinstance(mode, (bool, np.bool_)):
            # the boolean this setter was introduced with: False = fp64 arithmetic rounded at the store, True = packed fp32
            check(lib().azh_set_f32_arithmetic(self._h, 1 if mode else 0), "azh_set_f32_arithmetic")
            return
        m = self.F32_MODES[mode] if isinstance(mode, str) else int(mode)
        check(lib().azh_set_f32_mode(self._h, m), "azh_set_f32_mode")

    def set_fast_path(self, enabled):
        check(lib().azh_set_fast_path(self._h, 1 if enabled else 0), "azh_set_fast_path")

    def set_tile_kernel(self, enabled):
        """Time-major output on (quasi-)uniform grids: True / 1 = the 16-row tile kernel (default), 2 = the lane = satellite
        kernel k_cols_fast (opt-in: parity-identical, measured slower), False / 0 = neither (k_propagate)."""
        check(lib().azh_set_tile_kernel(self._h, int(enabled) if not isinstance(enabled, bool) else (1 if enabled else 0)), "azh_set_tile_kernel")

    def set_graphs(self, enabled):
        """hipGraph replay of repeated cached-input launch sets (propagate_device_cached / _window); default off: it pays for
        multi-window pipelines (ShardedPropagator with several chunks), not for a single launch set."""
        check(lib().azh_set_graphs(self._h, 1 if enabled else 0), "azh_set_graphs")

    def set_timing(self, enabled):
        check(lib().azh_set_timing(self._h, 1 if enabled else 0), "azh_set_timing")

    def synchronize(self):
        check(lib().azh_synchronize(self._h), "azh_synchronize")

    def last_kernel_ms(self):
        return lib().azh_last_kernel_ms(self._h)

    def last_path(self):
        """AZH_PATH_* bits (PATH_* below): which kernel families the most recent call launched."""
        return int(lib().azh_last_path(self._h))

    def last_one_stats(self):
        """The most recent propagate_one / propagate_one_device call: (segments of 1,024 points the branch-free kernel was
        launched on, segments it handed over to the generic kernel); (0, 0) when the call took the generic kernel throughout.
        Waits for the call to finish."""
        a, b = C.c_uint32(0), C.c_uint32(0)
        check(lib().azh_last_one_stats(self._h, C.byref(a), C.byref(b)), "azh_last_one_stats")
        return int(a.value), int(b.value)


class DeviceGroup:
    """azh_group: one catalog over several GPUs of this process (the C-host route to N devices; Python hosts with
    one process per GPU use astroz_amd.distributed instead)."""

    def __init__(self, text, devices, grav=WGS72, n_chunks=4):
        b = text.encode() if isinstance(text, str) else bytes(text)
        dv = (C.c_int32 * len(devices))(*devices)
        h = C.c_void_p()
        fn = lib().azh_group_create_from_omm_json if b.lstrip()[:1] in (b"{", b"[") else lib().azh_group_create_from_tle_text
        check(fn(b, len(b), grav, dv, len(devices), n_chunks, C.byref(h)), "azh_group_create")
        self._h = h
        self.n = lib().azh_group_num_satellites(h)
        self.n_devices = lib().azh_group_num_devices(h)
        self.padded_rows = lib().azh_group_padded_rows(h)
        self.epochs = np.empty(self.n)
        check(lib().azh_group_get_epochs(h, self.epochs.ctypes.data), "azh_group_get_epochs")

    def close(self):
        if getattr(self, "_h", None):
            lib().azh_group_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _offsets(self, offsets_min):
        """The C entry points index `offsets` by catalog row up to num_satellites without a length argument."""
        if offsets_min is None:
            return None
        off = _f64(offsets_min)
        if off.ndim != 1 or len(off) < self.n:
            raise ValueError("epoch_offsets must have at least num_satellites elements")
        return off

    def propagate_host(self, times_min, offsets_min=None, *, velocities=True, mode=OUT_TEME, reference_jd=0.0, errors=False):
        """-> pos (n, n_times, 3)[, vel][, err (n, n_times) u8], catalog order, host arrays."""
        t = _f64(times_min)
        off = self._offsets(offsets_min)
        pos = result_empty((self.n, len(t), 3))
        vel = result_empty((self.n, len(t), 3)) if velocities else None
        big = self.n * len(t) >= PINNED_MIN_BYTES
        err = (result_empty((self.n, len(t)), np.uint8) if big else np.zeros((self.n, len(t)), dtype=np.uint8)) if errors else None
        check(lib().azh_group_propagate_host(self._h, t.ctypes.data, len(t), _ptr(off), 0 if off is None else len(off), pos.ctypes.data, _ptr(vel), mode,
                                             float(reference_jd), _ptr(err)), "azh_group_propagate_host")
        return pos, vel, err

    def screen_target(self, times_min, target, threshold_km, offsets_min=None, reference_jd=0.0):
        """Fused single-target screen over all devices of the group (azh_group_screen_target_host): every device screens
        its own rows, no collective.  -> (min_dist (n,) float64 km, min_t (n,) uint32), catalog order."""
        t = _f64(times_min)
        off = self._offsets(offsets_min)
        d = np.empty(self.n)
        ti = np.empty(self.n, dtype=np.uint32)
        check(lib().azh_group_screen_target_host(self._h, t.ctypes.data, len(t), _ptr(off), 0 if off is None else len(off), int(target),
                                                 float(threshold_km), float(reference_jd), d.ctypes.data, ti.ctypes.data),
              "azh_group_screen_target_host")
        return d, ti

    def screen_target_device(self, times_min, target, threshold_km, d_dist_ptrs, d_t_ptrs, offsets_min=None):
        """The same with per-device result buffers (raw device pointers on devices[i], shard_size(i) entries each, local order =
        shard_rows(i)); asynchronous -- synchronize() waits for every shard."""
        t = _f64(times_min)
        off = self._offsets(offsets_min)
        dd = (C.c_void_p * self.n_devices)(*d_dist_ptrs)
        tt = (C.c_void_p * self.n_devices)(*d_t_ptrs)
        check(lib().azh_group_screen_target_device(self._h, t.ctypes.data, len(t), _ptr(off), 0 if off is None else len(off), int(target),
                                                   float(threshold_km), 0.0, dd, tt), "azh_group_screen_target_device")

    def shard_size(self, slot):
        return int(lib().azh_group_shard_size(self._h, int(slot)))

    def shard_rows(self, slot):
        out = np.empty(self.shard_size(slot), dtype=np.uint32)
        check(lib().azh_group_shard_rows(self._h, int(slot), out.ctypes.data), "azh_group_shard_rows")
        return out

    def synchronize(self):
        check(lib().azh_group_synchronize(self._h), "azh_group_synchronize")

    def propagate_allgather(self, times_min, offsets_min, d_pos_ptrs, d_vel_ptrs=None):
        """Full TEME arrays on every device: d_pos_ptrs[i] = raw device pointer on devices[i] to
        padded_rows x n_times x 3 doubles."""
        t = _f64(times_min)
        off = self._offsets(offsets_min)
        pp = (C.c_void_p * self.n_devices)(*d_pos_ptrs)
        vv = (C.c_void_p * self.n_devices)(*d_vel_ptrs) if d_vel_ptrs is not None else None
        check(lib().azh_group_propagate_allgather(self._h, t.ctypes.data, len(t), _ptr(off), 0 if off is None else len(off), pp, vv),
              "azh_group_propagate_allgather")


class _PinnedBlock:
    """Owner of one azh_host_alloc block: returns it to the library's pool when the last array over it is gone."""
    __slots__ = ("ptr", "_free", "__weakref__")

    def __init__(self, nbytes):
        self.ptr = None
        q = C.c_void_p(0)
        check(lib().azh_host_alloc(int(nbytes), C.byref(q)), "azh_host_alloc")
        self.ptr, self._free = q.value, lib().azh_host_free

    def __del__(self):
        if self.ptr:
            self._free(self.ptr)
            self.ptr = None


_PINNED_RESULTS = True
PINNED_MIN_BYTES = 8 << 20   # below this a pageable array costs nothing measurable (the small-call paths do not copy at all)
# Budget of LIVE pinned result bytes (blocks the caller still holds): beyond it results are plain numpy.empty arrays again (filled
# through the staging slots), so a caller who accumulates results -- as the reference's numpy.empty arrays allow -- does not page-
# lock the host.  Default 8 GiB = four config-2 results (1.9 GB each); ASTROZ_AMD_PINNED_LIVE_MB / set_pinned_budget change it.
_PINNED_LIVE_BUDGET = int(os.environ.get("ASTROZ_AMD_PINNED_LIVE_MB", "8192")) << 20


def set_pinned_budget(nbytes):
    """Upper bound on pinned result bytes alive at once (see set_pinned_results); results beyond it are pageable arrays."""
    global _PINNED_LIVE_BUDGET
    _PINNED_LIVE_BUDGET = max(0, int(nbytes))


def set_pinned_results(enabled):
    """Result arrays of the host-returning Python calls (SatrecArray.sgp4, propagate, DeviceGroup.propagate_host) come from the
    library's pinned pool (default: True for results of 8 MiB and more): the device-to-host DMA lands in them directly.
    False: plain numpy.empty arrays, filled through the pinned staging slots (about 25 % slower, no pinned memory held by
    results the caller keeps)."""
    global _PINNED_RESULTS
    _PINNED_RESULTS = bool(enabled)


def result_empty(shape, dtype=np.float64, pinned=None):
    """numpy.empty for a RESULT of a host-returning call: a writable C-contiguous ndarray over a pinned block of the library's
    pool (returned to the pool when the array and all its views are gone), or a plain numpy array when pinned results are
    switched off, the array is small, or the host cannot pin the memory."""
    n = 1
    for d in shape:
        n *= int(d)
    if n * 8 < PINNED_MIN_BYTES and dtype is np.float64:
        return np.empty(shape)            # (the small-call paths: nothing here may cost a microsecond)
    dt = np.dtype(dtype)
    nbytes = n * dt.itemsize
    use = _PINNED_RESULTS if pinned is None else pinned
    if not use or nbytes < PINNED_MIN_BYTES:
        return np.empty(shape, dtype=dt)
    if host_pool_stats()[0] + nbytes > _PINNED_LIVE_BUDGET:
        return np.empty(shape, dtype=dt)  # (the caller keeps earlier results alive: stay inside the live-bytes budget)
    try:
        blk = _PinnedBlock(nbytes)
    except NativeError:
        return np.empty(shape, dtype=dt)
    buf = (C.c_char * nbytes).from_address(blk.ptr)
    buf._az_owner = blk          # the ctypes array is the ndarray's base: the block lives as long as any view does
    return np.frombuffer(buf, dtype=dt).reshape(shape)


_fast_mod = False


def fast_scalar():
    """The CPython shim of the scalar call (astroz_amd/csrc/pyfast.c, built by __graft_entry__.build()), bound to the loaded
    library's azh_propagate_one_host -- or None when it was not built (Satrec.sgp4 then makes the same call through ctypes)."""
    global _fast_mod
    if _fast_mod is False:
        try:
            from . import _azfast
            _azfast.bind(C.cast(lib().azh_propagate_one_host, C.c_void_p).value)
            _fast_mod = _azfast
        except Exception:
            _fast_mod = None
    return _fast_mod


def set_host_points(n):
    """One-satellite host-pointer calls of at most n points run the library's own step on the calling thread instead of
    launching a kernel (azh_set_host_points; default 128, deep-space members half of it; 0 = never)."""
    lib().azh_set_host_points(int(n))


def get_host_points():
    return int(lib().azh_get_host_points())


def host_pool_stats():
    """(bytes of pinned result blocks in use, bytes kept free in the pool)"""
    a, b = C.c_size_t(0), C.c_size_t(0)
    lib().azh_host_pool_stats(C.byref(a), C.byref(b))
    return int(a.value), int(b.value)


def host_pool_trim():
    lib().azh_host_pool_trim()


def set_host_copy_threads(n):
    """Host threads behind the pinned staging of host-returning copies (-1 automatic, 0 = direct pageable copies)."""
    lib().azh_set_host_copy_threads(int(n))


def parse_tle_lines(line1, line2):
    """All numeric fields of one TLE (host-side text parsing only; no GPU needed)."""
    out = np.zeros(16, dtype=np.float64)
    rc = lib().azh_parse_tle_lines(line1.encode(), line2.encode(), out.ctypes.data)
    if rc != 0:
        raise ValueError("Failed to parse TLE lines")
    return out


def parse_element_text(text):
    """(n, 16) array of the numeric fields of every element set in multi-TLE text or OMM JSON (object or
    array).  Host-side text handling only (several threads on catalog-scale TLE text, see set_parse_threads);
    raises ValueError on malformed OMM."""
    b = text.encode() if isinstance(text, str) else bytes(text)
    fn = lib().azh_parse_omm_json if b.lstrip()[:1] in (b"{", b"[") else lib().azh_parse_tle_text
    k = C.c_size_t(0)
    cap = len(b) // 138 + 1  # a TLE record is at least two 69-character lines: one pass is enough for TLE text
    out = np.zeros((cap, 16), dtype=np.float64)
    rc = fn(b, len(b), out.ctypes.data, cap, C.byref(k))
    if rc != 0:
        raise ValueError("malformed element text (code %d)" % rc)
    if k.value > cap:  # (OMM records can be shorter than 138 bytes)
        out = np.zeros((k.value, 16), dtype=np.float64)
        rc = fn(b, len(b), out.ctypes.data, k.value, C.byref(k))
        if rc != 0:
            raise ValueError("malformed element text (code %d)" % rc)
    return out[:k.value]


def set_parse_threads(n):
    """Host threads for catalog-scale TLE text (0 = automatic, 1 = serial); process-wide."""
    lib().azh_set_parse_threads(int(n))


def coarse_screen(positions, threshold, valid_mask=None, *, layout=SAT_MAJOR, max_results=10_000_000, device=0,
                  device_ptr=None, shape=None, stream=None):
    """All-vs-all cell-list screen of a position array on the GPU (coarseScreen, conjunction.zig).
    `positions`: float64 host array (n_sats, n_times, 3) [sat-major] or (n_times, n_sats, 3)
    [time-major]; or pass device_ptr= + shape= for positions already in HBM."""
    if device_ptr is None:
        pos = np.ascontiguousarray(positions, dtype=np.float64)
        shape = pos.shape
    if len(shape) != 3 or shape[2] != 3:
        raise ValueError("positions must have shape (n_sats, n_times, 3) or (n_times, n_sats, 3)")
    ns, nt = (shape[0], shape[1]) if layout == SAT_MAJOR else (shape[1], shape[0])
    m = None if valid_mask is None else np.ascontiguousarray(valid_mask, dtype=np.uint8)
    if m is not None and len(m) < ns:
        raise ValueError("valid_mask must have num_sats elements")
    pairs = np.empty((max_results, 2), dtype=np.uint32)
    tt = np.empty(max_results, dtype=np.uint32)
    k = C.c_size_t(0)
    if device_ptr is None:
        check(lib().azh_coarse_screen_host(pos.ctypes.data, ns, nt, layout, 0, float(threshold), _ptr(m),
                                           pairs.ctypes.data, tt.ctypes.data, max_results, C.byref(k), device),
              "azh_coarse_screen_host")
    else:
        check(lib().azh_coarse_screen_device(device_ptr, ns, nt, layout, 0, float(threshold), _ptr(m),
                                             pairs.ctypes.data, tt.ctypes.data, max_results, C.byref(k), stream),
              "azh_coarse_screen_device")
    return pairs[:k.value].copy(), tt[:k.value].copy()


def selftest_math(x, device=0):
    """Device-side known answers of the kernels' element math: dict of arrays (see azh_selftest_math)."""
    x = _f64(np.atleast_1d(x))
    n = len(x)
    out = np.empty(6 * n)
    check(lib().azh_selftest_math(x.ctypes.data, n, out.ctypes.data, device), "azh_selftest_math")
    return {"sin": out[:n], "cos": out[n:2 * n], "x_rcp": out[2 * n:3 * n], "x_rsqrt2": out[3 * n:4 * n],
            "rot_sin": out[4 * n:5 * n], "rot_cos": out[5 * n:]}


def julian_to_gmst(jd):
    return float(lib().coords_julian_to_gmst(float(jd)))


def eci_to_ecef(eci, gmst):
    a = _f64(eci)
    out = np.empty(3)
    lib().coords_eci_to_ecef(a.ctypes.data, float(gmst), out.ctypes.data)
    return out


def ecef_to_geodetic(ecef):
    a = _f64(ecef)
    out = np.empty(3)
    lib().coords_ecef_to_geodetic(a.ctypes.data, out.ctypes.data)
    return out


def device_count():
    return lib().azh_device_count()
