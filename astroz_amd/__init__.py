"""astroz_amd -- MI355X-native batched SGP4/SDP4 constellation propagation.

Drop-in for the batched-propagation path of ATTron/astroz's Python package
(bindings/python/astroz/__init__.py L305-532): :class:`Constellation` and :func:`propagate`
with the same arguments, shapes and defaults; the python-sgp4 compatible layer is
:mod:`astroz_amd.api`.  All arithmetic runs in hand-written gfx950 HIP kernels behind
``libastroz_hip.so`` (C ABI in ``include/astroz_hip.h``); there is no CPU fallback.
"""
import json
import math
from datetime import datetime, timezone
from pathlib import Path

import numpy as np

from . import _native
from ._native import WGS72, WGS84  # noqa: F401

__version__ = "0.1.0"

_CELESTRAK_ALIASES = {"all": "active", "iss": "stations", "gps": "gps-ops", "glonass": "glo-ops"}
DAY_SECONDS = 86400.0


def _fetch_url(url):
    import urllib.request

    req = urllib.request.Request(url, headers={"User-Agent": "astroz"})
    return urllib.request.urlopen(req, timeout=60).read().decode("utf-8")


def _celestrak_url(group=None, norad_id=None, fmt="tle"):
    if norad_id is not None:
        ids = ",".join(str(i) for i in norad_id) if isinstance(norad_id, (list, tuple)) else str(norad_id)
        return "https://celestrak.org/NORAD/elements/gp.php?CATNR=%s&FORMAT=%s" % (ids, fmt.upper())
    name = _CELESTRAK_ALIASES.get(group.lower(), group)
    return "https://celestrak.org/NORAD/elements/gp.php?GROUP=%s&FORMAT=%s" % (name, fmt.upper())


def _sniff(text):
    return "json" if text.lstrip().startswith(("[", "{")) else "tle"


def _load_tle_text(source, norad_id=None):
    """(text, 'tle'|'json') from a group name, URL, path, raw TLE text or raw OMM JSON
    (reference __init__.py L163-181)."""
    if norad_id is not None:
        return _fetch_url(_celestrak_url(norad_id=norad_id)), "tle"
    if source is None:
        raise ValueError("Must specify 'source' or 'norad_id'")
    if source.startswith(("http://", "https://")):
        text = _fetch_url(source)
        return text, _sniff(text)
    if "1 " in source and "2 " in source:
        return source, "tle"
    if source.lstrip().startswith(("[", "{")):
        return source, "json"
    if len(source) < 4096 and Path(source).exists():
        text = Path(source).read_text()
        return text, _sniff(text)
    return _fetch_url(_celestrak_url(group=source)), "tle"


def _parse_tle_pairs(tle_text):
    """[(line1, line2)] from 2- or 3-line TLE text (reference __init__.py L184-200)."""
    lines = [ln.strip() for ln in tle_text.strip().splitlines() if ln.strip()]
    pairs, i = [], 0
    while i < len(lines):
        if lines[i].startswith("1 ") and i + 1 < len(lines) and lines[i + 1].startswith("2 "):
            pairs.append((lines[i], lines[i + 1]))
            i += 2
        else:
            i += 1
    return pairs


def _tle_checksum(line68):
    return sum(int(c) if c.isdigit() else (1 if c == "-" else 0) for c in line68) % 10


def _exp_field(val):
    if val == 0:
        return " 00000+0"
    av = abs(val)
    exp = math.floor(math.log10(av)) + 1
    mant = round(av * 10 ** (5 - exp))
    return "%s%05d%+d" % ("-" if val < 0 else " ", int(mant), exp)


def _omm_to_tle_pairs(json_text):
    """OMM JSON (object or array) -> [(line1, line2)] (reference __init__.py L203-279)."""
    data = json.loads(json_text)
    if isinstance(data, dict):
        data = [data]
    pairs = []
    for rec in data:
        norad = rec["NORAD_CAT_ID"]
        cls = (rec.get("CLASSIFICATION_TYPE", "U") or "U")[0]
        intl = rec.get("OBJECT_ID", "00000A") or "00000A"
        ndot = rec.get("MEAN_MOTION_DOT", 0) or 0
        nddot = rec.get("MEAN_MOTION_DDOT", 0) or 0
        etype = rec.get("EPHEMERIS_TYPE", 0) or 0
        elset = rec.get("ELEMENT_SET_NO", 0) or 0
        revnum = rec.get("REV_AT_EPOCH", 0) or 0
        dt = datetime.fromisoformat(rec["EPOCH"]).replace(tzinfo=None)
        doy = (dt - datetime(dt.year, 1, 1)).total_seconds() / DAY_SECONDS + 1.0
        if "-" in intl:
            yr, rest = intl.split("-", 1)
            intl_tle = "%s%-6s" % (yr[-2:], rest)
        else:
            intl_tle = "%-8s" % intl
        ndot_s = ("-" if ndot < 0 else " ") + ("%.8f" % abs(ndot))[1:]
        l1 = "1 %05d%s %s %02d%012.8f %s %s %s %s %4d" % (
            norad, cls, intl_tle, dt.year % 100, doy, ndot_s, _exp_field(nddot), _exp_field(rec["BSTAR"]),
            etype, elset)
        l1 = l1[:68].ljust(68)
        l1 += str(_tle_checksum(l1))
        l2 = "2 %05d %8.4f %8.4f %s %8.4f %8.4f %11.8f%5d" % (
            norad, rec["INCLINATION"], rec["RA_OF_ASC_NODE"], ("%.7f" % rec["ECCENTRICITY"])[2:],
            rec["ARG_OF_PERICENTER"], rec["MEAN_ANOMALY"], rec["MEAN_MOTION"], revnum)
        l2 = l2[:68].ljust(68)
        l2 += str(_tle_checksum(l2))
        pairs.append((l1, l2))
    return pairs


def _start_jd(start_time):
    """datetime -> Julian date, default now (reference __init__.py L282-286)."""
    if start_time is None:
        start_time = datetime.now(timezone.utc)
    return 2440587.5 + (start_time.timestamp() / 86400.0)


class Tle:
    """Two-Line Element set: ``astroz.Tle(tle_string)`` (bindings/python/src/tle.zig L14-104) over the
    c_api ``tle_parse`` / ``tle_get_*`` exports.  Text parsing only -- no GPU involved."""

    def __init__(self, tle_string):
        import ctypes as C
        if not isinstance(tle_string, str):
            raise TypeError("tle_string must be str")
        h = C.c_void_p()
        if _native.lib().tle_parse(tle_string.encode(), C.byref(h)) != 0:
            raise ValueError("Failed to parse TLE")
        self._h = h

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            _native.lib().tle_free(h)
            self._h = None

    satellite_number = property(lambda self: int(_native.lib().tle_get_satellite_number(self._h)),
                                doc="NORAD catalog number")
    epoch = property(lambda self: float(_native.lib().tle_get_epoch(self._h)), doc="Epoch (J2000 seconds)")
    inclination = property(lambda self: float(_native.lib().tle_get_inclination(self._h)), doc="Inclination (degrees)")
    eccentricity = property(lambda self: float(_native.lib().tle_get_eccentricity(self._h)), doc="Eccentricity")
    mean_motion = property(lambda self: float(_native.lib().tle_get_mean_motion(self._h)), doc="Mean motion (rev/day)")


class Constellation:
    """Pre-parsed, device-resident orbital elements for repeated propagation.

    ``source``: CelesTrak group name, URL, file path, raw TLE text or raw OMM JSON;
    ``norad_id``: catalog id(s) to fetch.  Output ordering follows the reference: near-earth
    satellites first ``[0, n_sgp4)``, deep-space after (reference __init__.py L374-393) -- but the
    deep-space rows ARE propagated here (the reference leaves them unwritten, L509-530).
    The gravity model is WGS84, the default of the reference's ``from_tle_text``
    (bindings/python/src/sgp4.zig L293-300)."""

    def __init__(self, source=None, *, norad_id=None, gravity_model=WGS84, device=0):
        text, fmt = _load_tle_text(source, norad_id)
        pairs = _omm_to_tle_pairs(text) if fmt == "json" else _parse_tle_pairs(text)
        if not pairs:
            raise ValueError("no TLE records found")
        dev = _native.DeviceConstellation.from_tle_lines(pairs, gravity_model, device)
        err, deep, _ = dev.status
        if err.any():
            bad = int(np.flatnonzero(err)[0])
            raise ValueError("%s (record %d)" % ("Invalid eccentricity" if err[bad] == 1 else "Satellite decayed", bad))
        if deep.any() and not deep.all():
            order = np.concatenate([np.flatnonzero(~deep), np.flatnonzero(deep)])
            if not np.array_equal(order, np.arange(len(pairs))):
                pairs = [pairs[i] for i in order]
                dev.close()
                dev = _native.DeviceConstellation.from_tle_lines(pairs, gravity_model, device)
        self._dev = dev
        self._pairs = pairs
        self._total_sats = len(pairs)
        self._n_sgp4 = int(dev.n_sgp4)

    @property
    def num_satellites(self):
        return self._total_sats

    @property
    def epochs(self):
        """TLE epoch of each satellite as a Julian date."""
        return list(self._dev.epochs)


def propagate(source, times, *, start_time=None, output="ecef", velocities=False, norad_id=None):
    """Propagate satellites to ``times`` (minutes from ``start_time``).

    Returns positions ``(n_times, n_satellites, 3)`` [km; or (lat rad, lon rad, alt km) for
    ``output="geodetic"`` -- radians, as the reference's kernel emits (Constellation.zig L497)],
    plus velocities ``(n_times, n_satellites, 3)`` km/s if ``velocities=True``.
    Reference: __init__.py L411-532."""
    const = source if isinstance(source, Constellation) else Constellation(source, norad_id=norad_id)
    if output not in _native.OUTPUT_MODES:
        raise ValueError("output must be 'ecef', 'teme', or 'geodetic'")
    times = np.ascontiguousarray(times, dtype=np.float64)
    n_sats, n_times = const.num_satellites, len(times)
    start = _start_jd(start_time)
    pos = np.empty((n_times, n_sats, 3), dtype=np.float64)
    vel = np.empty((n_times, n_sats, 3), dtype=np.float64) if velocities else None
    offsets = (start - const._dev.epochs) * 1440.0
    const._dev.propagate_host(times, offsets, pos=pos, vel=vel, mode=_native.OUTPUT_MODES[output],
                              reference_jd=start, layout=_native.TIME_MAJOR)
    return (pos, vel) if velocities else pos


def screen(source, times, threshold=10.0, *, target=None, start_time=None, norad_id=None):
    """Screen a constellation for conjunction events (reference __init__.py L535-658).

    ``target`` set: fused propagate+screen on the GPU against that satellite; returns
    ``(min_distances (n_sats,) float64 km, min_t_indices (n_sats,) uint32)`` -- a satellite that never
    comes within ``threshold`` (and the target itself) reports ``threshold`` and index 0.
    ``target`` None: all-vs-all; returns ``(pairs (k, 2) uint32, t_indices (k,) uint32)`` for every pair
    closer than ``threshold`` km at a grid time, sorted by (t, s, other).  Positions stay in HBM.
    Deep-space members take part in both modes (the reference's fused routine covers pure-SGP4
    constellations only and falls back to propagate-then-screen otherwise, L655-658)."""
    const = source if isinstance(source, Constellation) else Constellation(source, norad_id=norad_id)
    times = np.ascontiguousarray(times, dtype=np.float64)
    start = _start_jd(start_time)
    offsets = (start - const._dev.epochs) * 1440.0
    if target is not None:
        d, ti = const._dev.screen_target(times, int(target), float(threshold), offsets, reference_jd=start)
        return d, ti
    return const._dev.screen_all(times, float(threshold), offsets)


def coarse_screen(positions, num_sats, threshold, valid_mask=None):
    """``_astroz.coarse_screen(positions, num_sats, threshold, [valid_mask])`` (bindings/python/src/
    conjunction.zig L152-260): positions satellite-major ``(num_sats, n_times, 3)`` float64 (any shape
    with that memory order); returns ``(pairs, t_indices)`` as lists like the reference -- sorted by
    (t, s, other) rather than in hash-chain order."""
    pos = np.ascontiguousarray(positions, dtype=np.float64)
    if num_sats <= 0 or pos.size % (num_sats * 3) != 0:
        raise ValueError("positions array size not consistent with num_sats")
    pos = pos.reshape(num_sats, pos.size // (num_sats * 3), 3)
    pairs, tt = _native.coarse_screen(pos, threshold, valid_mask, layout=_native.SAT_MAJOR)
    return [tuple(int(x) for x in p) for p in pairs], [int(x) for x in tt]


__all__ = ["__version__", "Tle", "Constellation", "propagate", "screen", "coarse_screen", "WGS72", "WGS84"]
