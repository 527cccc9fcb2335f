"""python-sgp4 compatible API, GPU-backed -- mirror of the reference's ``astroz.api``.

Same names and call signatures as bindings/python/astroz/api.py (L86-359): ``Satrec``,
``SatrecArray``, ``jday``, ``days2mdhms``, ``WGS72``, ``WGS84``, ``WGS72OLD``, ``accelerated``::

    from astroz_amd.api import Satrec, SatrecArray, jday, WGS72

Every propagation runs in HIP kernels on an MI355X through libastroz_hip.so; there is no CPU
path.  Differences from the reference that are deliberate (SURVEY.md 8a17 "quirks not to copy"):

* error codes are real: ``e`` holds python-sgp4 codes per (satellite, time) and only the failing
  satellite's row is zero-filled (the reference returns all-zero ``e`` and zero-fills 8 lanes);
* the deep-space rows of ``SatrecArray.sgp4`` come back satellite-major as documented (the
  reference's ``sdp4_batch_propagate_into`` writes them time-major into a satellite-major array);
* ``jday`` is python-sgp4's exact formula (the reference's passes seconds through an f16).
"""
import math
import threading
import weakref

import numpy as np

from . import _native
from ._native import WGS72, WGS84

WGS72OLD = WGS72
accelerated = True

_TWOPI = 2.0 * math.pi
_DEG2RAD = math.pi / 180.0


def jday(year, mon, day, hr, minute, sec):
    """Calendar date -> (jd, fr): Julian date at the preceding midnight and the day fraction.
    python-sgp4's formula; reference Datetime.jday, src/Datetime.zig L235-240."""
    jd = (367.0 * year - 7 * (year + ((mon + 9) // 12.0)) * 0.25 // 1.0
          + 275 * mon / 9.0 // 1.0 + day + 1721013.5)
    fr = (sec + minute * 60.0 + hr * 3600.0) / 86400.0
    return jd, fr


_CUM_DAYS = (0, 31, 59, 90, 120, 151, 181, 212, 243, 273, 304, 334, 365)


def days2mdhms(year, days):
    """Fractional day of year -> (month, day, hour, minute, second).
    reference Datetime.days2mdhms, src/Datetime.zig L244-253."""
    whole = int(math.floor(days))
    leap = (year % 4 == 0 and year % 100 != 0) or year % 400 == 0
    month = 1
    while month < 12:
        end = _CUM_DAYS[month] + (1 if leap and month >= 2 else 0)
        if whole <= end:
            break
        month += 1
    start = _CUM_DAYS[month - 1] + (1 if leap and month > 2 else 0)
    day = whole - start
    rem = (days - whole) * 24.0
    hour = int(math.floor(rem))
    rem = (rem - hour) * 60.0
    minute = int(math.floor(rem))
    second = (rem - minute) * 60.0
    return month, day, hour, minute, second


# Satrec objects whose device-side initialisation is still pending, by gravity model (weak references).  The first one that
# needs its handle takes ALL pending records of its gravity model along: one constellation handle, one init launch -- making a
# handle costs ~0.4 ms (a stream, events, allocations, a launch, a synchronize) and freeing one ~0.5 ms, which a python-sgp4
# style loop over a catalog (`[Satrec.twoline2rv(a, b) for ...]`, then `sat.sgp4(jd, fr)` for each) would pay 13,478 times.
_PENDING = {}
_PENDING_LOCK = threading.Lock()


class Satrec:
    """One satellite record.  Use :meth:`twoline2rv`.

    TLE text is parsed on the host at construction; the element initialisation (and every
    propagation of more than a few points) runs on the GPU and is performed lazily -- together with every other record that is
    waiting for it (one shared handle, `_idx` = the record's row) -- or in one batch when the record is placed in a
    :class:`SatrecArray`."""

    def __init__(self, line1, line2, whichconst, fields):
        self._line1 = line1
        self._line2 = line2
        self.whichconst = int(whichconst)
        self._f = fields
        epoch_jd = float(fields[3])
        # python-sgp4 style split (bindings/python/src/satrec.zig L129-131)
        self.jdsatepoch = math.floor(epoch_jd - 0.5) + 0.5
        self.jdsatepochF = epoch_jd - self.jdsatepoch
        self.error = 0
        self.t = 0.0
        self._dev = None      # device constellation holding this record (lazy; shared with the records initialised with it)
        self._idx = 0         # ... and the record's row in it
        self._status = None   # (err, is_deep, irez), filled by _ensure or by a SatrecArray
        self._a = None
        self._scalar = None   # (shim function, handle address, row, epoch) of the scalar call, or False without the shim
        with _PENDING_LOCK:
            lst = _PENDING.setdefault(self.whichconst, [])
            lst.append(weakref.ref(self))
            if (len(lst) & 4095) == 0:      # (records that were dropped, or initialised through somebody else's batch)
                lst[:] = [r for r in lst if r() is not None and r()._dev is None]

    @classmethod
    def twoline2rv(cls, line1, line2, whichconst=WGS72):
        """Create a Satrec from TLE lines (bindings/python/src/satrec.zig L83-166)."""
        fields = _native.parse_tle_lines(line1, line2)
        return cls(line1, line2, whichconst, fields)

    # -- TLE-derived attributes (satrec.zig L390-430) -----------------------------------
    satnum = property(lambda self: int(self._f[0]))
    epochyr = property(lambda self: int(self._f[1]))
    epochdays = property(lambda self: float(self._f[2]))
    ecco = property(lambda self: float(self._f[8]))
    inclo = property(lambda self: float(self._f[6]) * _DEG2RAD)
    nodeo = property(lambda self: float(self._f[7]) * _DEG2RAD)
    argpo = property(lambda self: float(self._f[9]) * _DEG2RAD)
    mo = property(lambda self: float(self._f[10]) * _DEG2RAD)
    no_kozai = property(lambda self: float(self._f[11]) * _TWOPI / 1440.0)
    bstar = property(lambda self: float(self._f[5]))
    ndot = property(lambda self: float(self._f[4]) * _TWOPI / (1440.0 * 1440.0))

    # -- device-derived attributes (satrec.zig L432-474) ---------------------------------
    def _ensure(self):
        if self._dev is None:
            with _PENDING_LOCK:
                refs = _PENDING.pop(self.whichconst, [])
            batch = [self]
            for ref in refs:
                s = ref()
                if s is not None and s is not self and s._dev is None:
                    batch.append(s)
            dev = _native.DeviceConstellation.from_tle_lines([(s._line1, s._line2) for s in batch], self.whichconst)
            e, d, r = dev.status
            for i, s in enumerate(batch):
                s._dev, s._idx = dev, i
                s._status = (int(e[i]), bool(d[i]), int(r[i]))
                if s._status[0]:
                    s.error = s._status[0]
        return self._dev

    @property
    def is_deep_space(self):
        if self._status is None:
            self._ensure()
        return self._status[1]

    @property
    def a(self):
        if self._a is None:
            self._a = float(self._ensure().field("a")[self._idx])
        return self._a

    @property
    def alta(self):
        return self.a * (1.0 + self.ecco) - 1.0

    @property
    def altp(self):
        return self.a * (1.0 - self.ecco) - 1.0

    # -- propagation ---------------------------------------------------------------------
    def sgp4(self, jd, fr):
        """-> (error, (x, y, z) km, (vx, vy, vz) km/s), TEME.  satrec.zig L169-201.

        One point does not launch a kernel: the library evaluates its per-point step on the calling thread from the elements
        the GPU initialised (azh_set_host_points; include/astroz_hip.h), reached through a CPython shim when it is built
        (astroz_amd/csrc/pyfast.c) and through ctypes otherwise."""
        sc = self._scalar
        if sc is None:
            sc = self._bind_scalar()
        if sc:
            t, rc, e, r, v = sc[0](sc[1], sc[2], jd, fr, sc[3])
            if rc:
                _native.check(rc, "azh_propagate_one_host")
            self.t = t
            self.error = e
            return e, r, v
        tsince = ((jd + fr) - (self.jdsatepoch + self.jdsatepochF)) * 1440.0
        self.t = tsince
        e, r, v = self._ensure().propagate_one(self._idx, tsince)
        self.error = int(e[0])
        return self.error, tuple(float(x) for x in r[0]), tuple(float(x) for x in v[0])

    def sgp4_array_into(self, jd, fr, positions, velocities):
        """The native single-satellite batch call (satrec.zig L296-345): like :meth:`sgp4_array`, written into the caller's
        writable float64 buffers of ``len(jd) * 3`` elements each."""
        jd = np.atleast_1d(np.asarray(jd, dtype=np.float64))
        fr = np.atleast_1d(np.asarray(fr, dtype=np.float64))
        tsince = ((jd + fr) - (self.jdsatepoch + self.jdsatepochF)) * 1440.0
        outs = []
        for name, a in (("positions", positions), ("velocities", velocities)):
            a = np.asarray(a)
            if a.dtype != np.float64 or not a.flags.c_contiguous or not a.flags.writeable or a.size < 3 * len(tsince):
                raise ValueError("%s must be a writable C-contiguous float64 array of len(jd) * 3 elements" % name)
            outs.append(a.reshape(-1)[:3 * len(tsince)])
        _, r, v = self._ensure().propagate_one(self._idx, tsince)
        outs[0][:] = r.reshape(-1)
        outs[1][:] = v.reshape(-1)

    def _bind_scalar(self):
        dev = self._ensure()
        mod = _native.fast_scalar()
        self._scalar = (mod.sgp4, dev.handle_address(), self._idx, self.jdsatepoch + self.jdsatepochF) if mod is not None else False
        return self._scalar

    def sgp4_array(self, jd, fr):
        """Many times, one satellite (lane = time on the GPU).  -> e (n,), r (n,3), v (n,3)."""
        jd = np.atleast_1d(np.asarray(jd, dtype=np.float64))
        fr = np.atleast_1d(np.asarray(fr, dtype=np.float64))
        tsince = ((jd + fr) - (self.jdsatepoch + self.jdsatepochF)) * 1440.0
        e, r, v = self._ensure().propagate_one(self._idx, tsince)
        return e, r, v


class SatrecArray:
    """Batch propagator: all satellites x all times in one GPU call.

    Near-earth and deep-space members are handled transparently (two kernels on two streams);
    the gravity model is the first satellite's, as in the reference (satrec.zig L879-880)."""

    def __init__(self, satrecs, device=0):
        satrecs = list(satrecs)
        if not satrecs:
            raise ValueError("Must provide at least one Satrec object")
        for s in satrecs:
            if not isinstance(s, Satrec):
                raise TypeError("All items must be Satrec objects")
        self._num_sats = len(satrecs)
        self._device = int(device)
        grav = satrecs[0].whichconst
        self._dev = _native.DeviceConstellation.from_tle_lines(
            [(s._line1, s._line2) for s in satrecs], grav, device)
        err, deep, irez = self._dev.status
        if err.any():
            # shared.buildBatches raises on any init failure (shared.zig L78-82)
            bad = int(np.flatnonzero(err)[0])
            raise ValueError("%s (satellite index %d)" % (
                "Invalid eccentricity" if err[bad] == 1 else "Satellite decayed", bad))
        for s, e, d, r in zip(satrecs, err, deep, irez):
            if s._status is None and s.whichconst == grav:
                s._status = (int(e), bool(d), int(r))
        self._sgp4_indices = np.flatnonzero(~deep)
        self._sdp4_indices = np.flatnonzero(deep)
        self._epochs = self._dev.epochs

    @property
    def num_satellites(self):
        return self._num_sats

    @property
    def epochs(self):
        """Epoch Julian date of every satellite (the native type's `epochs` getter, bindings/python/src/satrec.zig L807)."""
        return [float(x) for x in self._epochs]

    def propagate_into(self, times, positions, velocities=None, epoch_offsets=None):
        """The native batch call the reference's facade is built on (satrec.zig L895-990): TEME, TIME-major --
        ``positions`` / ``velocities`` are caller-owned writable float64 buffers of at least ``num_satellites * n_times * 3``
        elements, filled as ``(n_times, num_satellites, 3)``; ``times`` in minutes, ``epoch_offsets`` (minutes, per satellite;
        default: zeros = every satellite relative to its own epoch).  ValueError when a buffer is too small."""
        t = np.ascontiguousarray(times, dtype=np.float64)
        need = self._num_sats * len(t) * 3
        arrs = []
        for name, a in (("positions", positions), ("velocities", velocities)):
            if a is None:
                arrs.append(None)
                continue
            a = np.asarray(a)
            if a.dtype != np.float64 or not a.flags.c_contiguous or not a.flags.writeable:
                raise ValueError("%s must be a writable C-contiguous float64 array" % name)
            if a.size < need:
                raise ValueError("%s array too small" % name)
            arrs.append(a.reshape(-1)[:need])
        off = None if epoch_offsets is None else np.ascontiguousarray(epoch_offsets, dtype=np.float64)
        self._dev.propagate_host(t, off, pos=arrs[0], vel=arrs[1], layout=_native.TIME_MAJOR)

    def _grid(self, jd, fr):
        jd = np.atleast_1d(np.asarray(jd, dtype=np.float64))
        fr = np.atleast_1d(np.asarray(fr, dtype=np.float64))
        # api.py L300-302
        reference_jd = jd[0] + fr[0]
        epoch_offsets = (reference_jd - self._epochs) * 1440.0
        times = ((jd + fr) - reference_jd) * 1440.0
        return times, epoch_offsets

    def sgp4(self, jd, fr, *, velocities=True):
        """-> e (n_sats, n_times) uint8, r, v (n_sats, n_times, 3) float64 [km, km/s, TEME].

        Like the reference the arrays are physically time-major and returned as transposed
        views (api.py L307-320)."""
        times, offsets = self._grid(jd, fr)
        n_times, n_sats = len(times), self._num_sats
        # the callee allocates, as in the reference (api.py L304-314) -- from pinned memory the DMA engines write directly
        # (large results; _native.set_pinned_results(False) restores plain numpy.empty arrays)
        r_tm = _native.result_empty((n_times, n_sats, 3))
        v_tm = _native.result_empty((n_times, n_sats, 3)) if velocities else None
        # (the error matrix too when it is large: 19 MB for config 2 -- a pageable array of that size costs the copy a millisecond
        # of first-time pinning; the library writes every byte of it, so it needs no zero-fill)
        big = n_sats * n_times >= _native.PINNED_MIN_BYTES
        e = _native.result_empty((n_sats, n_times), np.uint8) if big else np.zeros((n_sats, n_times), dtype=np.uint8)
        self._dev.propagate_host(times, offsets, pos=r_tm, vel=v_tm, layout=_native.TIME_MAJOR, err=e)
        r = r_tm.transpose(1, 0, 2)
        v = v_tm.transpose(1, 0, 2) if velocities else np.zeros((n_sats, n_times, 3), dtype=np.float64)
        return e, r, v

    def sgp4_device(self, jd, fr, *, velocities=True, stream=None, padded=False, layout="time"):
        """Same computation, results left resident in HBM: returns torch tensors on the GPU, (e, r, v) with e (n_sats, n_times)
        uint8 and float64 r, v whose shape follows `layout`.  Asynchronous with respect to the host; ordered on the
        constellation's stream (or `stream`).

        layout="time" (default): r, v are (n_times, n_sats, 3), the reference's physical layout (api.py L304-314), dense and
        contiguous unless padded=True.
        layout="sat": r, v are dense (n_sats, n_times, 3) tensors -- the shapes `sgp4` returns, in the physical layout the GPU
        writes a third faster (the reference's time-major layout exists for its CPU threads).  padded=True has no meaning
        there and raises ValueError.

        padded=True: the time rows of the underlying arrays are padded to a multiple of 16 satellites (384 bytes = three whole
        128-byte lines: every run of the time-major tile kernel then starts on a line boundary and leaves as streaming stores,
        about 8 % faster); r_tm / v_tm are then NON-contiguous (n_times, n_sats, 3) views of
        (n_times, stride, 3) arrays whose padding columns are zero -- do not hand their data_ptr() to code that assumes a
        dense layout."""
        import torch

        if layout not in ("time", "sat"):
            raise ValueError("layout must be 'time' or 'sat'")
        if layout == "sat" and padded:
            raise ValueError("padded=True applies to layout='time' only (satellite-major rows are dense)")
        times, offsets = self._grid(jd, fr)
        n_times, n_sats = len(times), self._num_sats
        dev = torch.device("cuda", self._device)  # the device the element table lives on
        if layout == "sat":
            # satellite-major physical layout: (n_sats, n_times, 3) dense tensors, what `sgp4` returns as shapes -- and the
            # layout this GPU writes fastest (one wave per satellite row: config 2 in 0.20 ms against 0.30 ms time-major)
            r = torch.empty((n_sats, n_times, 3), dtype=torch.float64, device=dev)
            v = torch.empty((n_sats, n_times, 3), dtype=torch.float64, device=dev) if velocities else None
            e = torch.empty((n_sats, n_times), dtype=torch.uint8, device=dev)
            torch.cuda.current_stream(dev).synchronize()
            self._dev.propagate_device(times, offsets, r.data_ptr(), None if v is None else v.data_ptr(),
                                       layout=_native.SAT_MAJOR, d_err=e.data_ptr(), stream=stream)
            return e, r, v
        stride = (n_sats + 15) // 16 * 16 if padded else n_sats
        alloc = torch.zeros if stride != n_sats else torch.empty
        r_tm = alloc((n_times, stride, 3), dtype=torch.float64, device=dev)
        v_tm = alloc((n_times, stride, 3), dtype=torch.float64, device=dev) if velocities else None
        e = torch.empty((n_sats, n_times), dtype=torch.uint8, device=dev)
        torch.cuda.current_stream(dev).synchronize()  # allocations visible before a foreign stream writes
        self._dev.propagate_device(times, offsets, r_tm.data_ptr(), None if v_tm is None else v_tm.data_ptr(),
                                   layout=_native.TIME_MAJOR, stride=stride, d_err=e.data_ptr(), stream=stream)
        if stride == n_sats:
            return e, r_tm, v_tm
        return e, r_tm[:, :n_sats], (None if v_tm is None else v_tm[:, :n_sats])

    def synchronize(self):
        self._dev.synchronize()


__all__ = ["Satrec", "SatrecArray", "jday", "days2mdhms", "WGS72", "WGS84", "WGS72OLD", "accelerated"]
