"""ctypes binding of the CPU ORACLE (oracle/astroz_oracle.c).

TEST INFRASTRUCTURE ONLY.  May be imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg -- never by anything under astroz_amd/.  See oracle/astroz_oracle.h.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libastroz_oracle.so")

WGS84, WGS72 = 0, 1
TEME, ECEF, GEODETIC = 0, 1, 2
SAT_MAJOR, TIME_MAJOR = 0, 1


def build(force=False):
    """Compile the oracle with gcc (oracle/Makefile)."""
    src = os.path.join(_HERE, "astroz_oracle.c")
    if (force or not os.path.exists(_LIB_PATH)
            or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src)):
        subprocess.check_call(["make", "-s", "-C", _HERE])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        L.orc_sizeof_tle.restype = C.c_size_t
        L.orc_sizeof_sat.restype = C.c_size_t
        L.orc_tle_parse_lines.argtypes = [C.c_char_p, C.c_char_p, C.c_void_p]
        L.orc_tle_parse.argtypes = [C.c_char_p, C.c_void_p]
        L.orc_tle_parse_multi.argtypes = [C.c_char_p, C.c_void_p, C.c_size_t]
        L.orc_tle_parse_multi.restype = C.c_size_t
        L.orc_year_doy_to_jd.argtypes = [C.c_int, C.c_double]
        L.orc_year_doy_to_jd.restype = C.c_double
        L.orc_sat_init.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.orc_sgp4_init_only.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.orc_sat_field.argtypes = [C.c_void_p, C.c_char_p]
        L.orc_sat_field.restype = C.c_double
        L.orc_sat_propagate.argtypes = [C.c_void_p, C.c_double, C.c_void_p, C.c_void_p]
        L.orc_gstime.argtypes = [C.c_double]
        L.orc_gstime.restype = C.c_double
        L.orc_julian_to_gmst.argtypes = [C.c_double]
        L.orc_julian_to_gmst.restype = C.c_double
        L.orc_ecef_to_geodetic.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_propagate_constellation.argtypes = [
            C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p,
            C.c_int, C.c_double, C.c_void_p, C.c_int, C.c_size_t, C.c_void_p, C.c_int]
        L.orc_propagate_constellation.restype = None
        L.orc_max_threads.restype = C.c_int
        L.orc_screen_target.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t,
                                        C.c_double, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_screen_target.restype = None
        L.orc_coarse_screen.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_double, C.c_void_p, C.c_void_p,
                                        C.c_void_p, C.c_size_t]
        L.orc_coarse_screen.restype = C.c_size_t
        _lib = L
    return _lib


class TleFields(C.Structure):
    _fields_ = [("satnum", C.c_uint32), ("classification", C.c_char), ("epoch_year", C.c_int),
                ("epoch_day", C.c_double), ("epoch_jd", C.c_double), ("ndot", C.c_double),
                ("bstar", C.c_double), ("incl_deg", C.c_double), ("raan_deg", C.c_double),
                ("ecc", C.c_double), ("argp_deg", C.c_double), ("ma_deg", C.c_double),
                ("mm_revday", C.c_double), ("elnum", C.c_uint32), ("revnum", C.c_uint32)]


def parse_lines(line1, line2):
    t = TleFields()
    assert C.sizeof(t) == lib().orc_sizeof_tle()
    rc = lib().orc_tle_parse_lines(line1.encode(), line2.encode(), C.byref(t))
    if rc != 0:
        raise ValueError("oracle: bad TLE (rc=%d)" % rc)
    return t


def parse_text(text):
    t = TleFields()
    rc = lib().orc_tle_parse(text.encode(), C.byref(t))
    if rc != 0:
        raise ValueError("oracle: bad TLE (rc=%d)" % rc)
    return t


def parse_multi(text, max_n=None):
    if max_n is None:
        max_n = text.count("\n") // 2 + 2
    arr = (TleFields * max_n)()
    n = lib().orc_tle_parse_multi(text.encode(), arr, max_n)
    return [arr[i] for i in range(n)]


class Catalog:
    """An array of initialised oracle satellites (opaque orc_sat records)."""

    def __init__(self, tles, grav=WGS72):
        L = lib()
        self.n = len(tles)
        self._sz = L.orc_sizeof_sat()
        self._buf = C.create_string_buffer(self._sz * max(self.n, 1))
        self.init_rc = np.zeros(self.n, dtype=np.int32)
        base = C.addressof(self._buf)
        for i, t in enumerate(tles):
            self.init_rc[i] = L.orc_sat_init(C.byref(t), grav, base + i * self._sz)
        self.epoch_jd = np.array([t.epoch_jd for t in tles], dtype=np.float64)

    @classmethod
    def from_pairs(cls, pairs, grav=WGS72):
        return cls([parse_lines(a, b) for a, b in pairs], grav)

    @classmethod
    def from_text(cls, text, grav=WGS72):
        return cls(parse_multi(text), grav)

    def _ptr(self, i):
        return C.addressof(self._buf) + i * self._sz

    def field(self, i, name):
        return lib().orc_sat_field(self._ptr(i), name.encode())

    def fields(self, name):
        return np.array([self.field(i, name) for i in range(self.n)])

    @property
    def is_deep(self):
        return self.fields("is_deep").astype(bool)

    def propagate_one(self, i, tsince):
        r = (C.c_double * 3)()
        v = (C.c_double * 3)()
        rc = lib().orc_sat_propagate(self._ptr(i), float(tsince), r, v)
        return rc, np.array(r[:]), np.array(v[:])

    def propagate(self, times_min, offsets_min=None, *, velocities=True, mode=TEME,
                  reference_jd=0.0, mask=None, layout=SAT_MAJOR, stride=0, threads=1, out=None):
        """Returns (err (n_sats,n_times) u8, pos, vel-or-None); pos/vel shaped per layout.
        `out` = (err, pos, vel) from a previous call re-uses those buffers (timing loops)."""
        times = np.ascontiguousarray(times_min, dtype=np.float64)
        nt = len(times)
        ns = self.n
        st = stride or ns
        shape = (ns, nt, 3) if layout == SAT_MAJOR else (nt, st, 3)
        if out is not None:
            err, pos, vel = out
            assert pos.shape == shape and err.shape == (ns, nt)
        else:
            pos = np.zeros(shape, dtype=np.float64)
            vel = np.zeros(shape, dtype=np.float64) if velocities else None
            err = np.zeros((ns, nt), dtype=np.uint8)
        off = None if offsets_min is None else np.ascontiguousarray(offsets_min, dtype=np.float64)
        if off is not None:
            assert len(off) >= ns
        m = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8)
        lib().orc_propagate_constellation(
            C.addressof(self._buf), ns, times.ctypes.data, nt,
            None if off is None else off.ctypes.data, pos.ctypes.data,
            None if vel is None else vel.ctypes.data, mode, float(reference_jd),
            None if m is None else m.ctypes.data, layout, st, err.ctypes.data, int(threads))
        return err, pos, vel


def _catalog_screen_target(self, times_min, target, threshold, offsets_min=None, reference_jd=0.0):
    """(min_dist (n,), min_t (n,) u32): Constellation.screenConstellation (Constellation.zig L683-756)."""
    times = np.ascontiguousarray(times_min, dtype=np.float64)
    off = None if offsets_min is None else np.ascontiguousarray(offsets_min, dtype=np.float64)
    failed = np.ascontiguousarray(self.init_rc != 0, dtype=np.uint8)
    d = np.zeros(self.n, dtype=np.float64)
    ti = np.zeros(self.n, dtype=np.uint32)
    lib().orc_screen_target(C.addressof(self._buf), self.n, times.ctypes.data, len(times),
                            None if off is None else off.ctypes.data, int(target), float(threshold),
                            float(reference_jd), failed.ctypes.data, d.ctypes.data, ti.ctypes.data)
    return d, ti


Catalog.screen_target = _catalog_screen_target


def coarse_screen(positions, threshold, valid_mask=None, max_results=10_000_000):
    """positions (n_sats, n_times, 3) float64 satellite-major -> (pairs (k,2) u32, t (k,) u32) in the
    reference's traversal order (conjunction.zig L11-150)."""
    pos = np.ascontiguousarray(positions, dtype=np.float64)
    ns, nt = pos.shape[0], pos.shape[1]
    pairs = np.zeros((max_results, 2), dtype=np.uint32)
    tt = np.zeros(max_results, dtype=np.uint32)
    m = None if valid_mask is None else np.ascontiguousarray(valid_mask, dtype=np.uint8)
    k = lib().orc_coarse_screen(pos.ctypes.data, ns, nt, float(threshold), None if m is None else m.ctypes.data,
                                pairs.ctypes.data, tt.ctypes.data, max_results)
    return pairs[:k].copy(), tt[:k].copy()


_b8 = None


def batch8_lib():
    """libastroz_batch8.<cpu tag>.so, compiled with -march=native on THIS host on first use."""
    global _b8
    if _b8 is None:
        import hashlib
        flags = ""
        try:
            for ln in open("/proc/cpuinfo"):
                if ln.startswith("flags"):
                    flags = ln
                    break
        except OSError:
            pass
        tag = hashlib.sha1(flags.encode()).hexdigest()[:10]
        path = os.path.join(_HERE, "libastroz_batch8.%s.so" % tag)
        src = os.path.join(_HERE, "astroz_batch8.c")
        if not os.path.exists(path) or os.path.getmtime(path) < os.path.getmtime(src):
            subprocess.check_call(["make", "-s", "-C", _HERE, "batch8", "BATCH8_TAG=" + tag])
        L = C.CDLL(path)
        L.orc_batch8_propagate.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p,
                                           C.c_void_p, C.c_int, C.c_size_t, C.c_int]
        L.orc_batch8_propagate.restype = C.c_size_t
        _b8 = L
    return _b8


def _catalog_propagate_batch8(self, times_min, offsets_min=None, *, velocities=True, layout=SAT_MAJOR, threads=1,
                              out=None):
    """CPU baseline in the reference's SIMD design (astroz_batch8.c); near-earth members only.
    Returns (failed_batch_steps, pos, vel)."""
    assert not self.is_deep.any() and not self.init_rc.any(), "batch8 baseline: near-earth, initialised members only"
    times = np.ascontiguousarray(times_min, dtype=np.float64)
    nt, ns = len(times), self.n
    shape = (ns, nt, 3) if layout == SAT_MAJOR else (nt, ns, 3)
    if out is not None:
        pos, vel = out
    else:
        pos = np.zeros(shape)
        vel = np.zeros(shape) if velocities else None
    off = None if offsets_min is None else np.ascontiguousarray(offsets_min, dtype=np.float64)
    failed = batch8_lib().orc_batch8_propagate(
        C.addressof(self._buf), ns, times.ctypes.data, nt, None if off is None else off.ctypes.data, pos.ctypes.data,
        None if vel is None else vel.ctypes.data, layout, 0, int(threads))
    return failed, pos, vel


Catalog.propagate_batch8 = _catalog_propagate_batch8


def gstime(jd):
    return lib().orc_gstime(float(jd))


def julian_to_gmst(jd):
    return lib().orc_julian_to_gmst(float(jd))


def max_threads():
    return lib().orc_max_threads()
