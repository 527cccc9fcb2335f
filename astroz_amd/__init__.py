"""astroz_amd -- MI355X-native batched SGP4/SDP4 constellation propagation.

Drop-in for the batched-propagation path of ATTron/astroz's Python package: :class:`Constellation`,
:func:`propagate`, :func:`screen` (bindings/python/astroz/__init__.py L305-658) and the extension type
:class:`Sgp4Constellation` (bindings/python/astroz/_astroz.pyi L501-630) with the same arguments, shapes and
defaults; the python-sgp4 compatible layer is :mod:`astroz_amd.api`.  All arithmetic runs in hand-written
gfx950 HIP kernels behind ``libastroz_hip.so`` (C ABI in ``include/astroz_hip.h``); there is no CPU fallback.

Element text goes to the library as it is: TLE text is read by the library's fixed-column reader, OMM JSON by
its OMM reader (full JSON precision; the reference's Python layer re-renders OMM as 69-column text first and
loses digits there).

Network sources are OPT-IN.  Like the reference, ``source`` may also be a URL or a CelesTrak group name and
``norad_id=`` a catalog number (or a list), but nothing is downloaded unless the caller allows it: pass
``fetch=callable(url) -> str`` (or install one with :func:`set_fetcher`), or ``allow_network=True`` to use the
built-in ``urllib`` fetcher.  Without either such a source raises ``ValueError`` naming the URL it would have
fetched.  The fetched text goes to the same native readers as any other text.
"""
import os
from datetime import datetime, timezone

import numpy as np

from . import _native
from ._native import WGS72, WGS84  # noqa: F401

__version__ = "0.2.0"

_UNIX_EPOCH_JD = 2440587.5


_CELESTRAK_GP = "https://celestrak.org/NORAD/elements/gp.php"
# group aliases of the reference's loader (bindings/python/astroz/__init__.py L131-136): short names -> CelesTrak GROUP values
_GROUP_ALIASES = {"all": "active", "iss": "stations", "gps": "gps-ops", "glonass": "glo-ops"}
_fetcher = None


def set_fetcher(fn):
    """Install the process-wide fetcher used for URL / CelesTrak-group / ``norad_id`` sources: ``fn(url) -> str``
    (``None`` removes it).  Returns the previous one."""
    global _fetcher
    prev, _fetcher = _fetcher, fn
    return prev


def celestrak_url(group=None, norad_id=None, fmt="tle"):
    """The CelesTrak GP query for a group name (``"starlink"``, ``"active"``, ...) or catalog number(s)."""
    if (group is None) == (norad_id is None):
        raise ValueError("exactly one of group / norad_id")
    if norad_id is not None:
        ids = norad_id if isinstance(norad_id, (list, tuple)) else [norad_id]
        return "%s?CATNR=%s&FORMAT=%s" % (_CELESTRAK_GP, ",".join(str(int(i)) for i in ids), fmt)
    name = str(group).strip().lower()
    return "%s?GROUP=%s&FORMAT=%s" % (_CELESTRAK_GP, _GROUP_ALIASES.get(name, name), fmt)


def _urllib_fetch(url):
    import urllib.request
    req = urllib.request.Request(url, headers={"User-Agent": "astroz_amd/%s" % __version__})
    with urllib.request.urlopen(req, timeout=60) as resp:
        return resp.read().decode("utf-8")


def _download(url, fetch, allow_network):
    fn = fetch or _fetcher or (_urllib_fetch if allow_network else None)
    if fn is None:
        raise ValueError("this source needs a download (%s): pass fetch=callable(url) -> str, install one with "
                         "astroz_amd.set_fetcher, or pass allow_network=True" % url)
    text = fn(url)
    if not isinstance(text, str) or not text.strip():
        raise ValueError("the fetcher returned no element text for %s" % url)
    return text


def _as_text(source, norad_id=None, fetch=None, allow_network=False):
    """The element text behind `source`: the string itself when it already is element data, the contents of the
    local file it names, or -- opt-in, see the module docstring -- what a URL / CelesTrak group name / `norad_id`
    resolves to (reference: bindings/python/astroz/__init__.py L163-181)."""
    if norad_id is not None:   # (with both given the catalog number wins, as in the reference: __init__.py L163-166)
        return _download(celestrak_url(norad_id=norad_id), fetch, allow_network)
    if source is None:
        raise ValueError("Must specify 'source' or 'norad_id'")
    if not isinstance(source, str):
        raise TypeError("source must be a string: TLE text, OMM JSON text, a local file path, a URL or a CelesTrak group")
    head = source.lstrip()[:1]
    if head in ("{", "["):
        return source
    one_line = "\n" not in source and len(source) < 4096
    if one_line and source.lower().startswith(("http://", "https://")):
        if source.lower().startswith("http://"):
            import warnings
            warnings.warn("element sets fetched over plain http are not authenticated: prefer https", stacklevel=3)
        return _download(source, fetch, allow_network)
    if one_line and os.path.isfile(source):
        with open(source, "r", encoding="utf-8") as fh:
            return fh.read()
    if any(ln.lstrip().startswith("1 ") for ln in source.splitlines()):
        return source
    name = source.strip()
    if name.lower().startswith("celestrak:"):
        name = name[len("celestrak:"):].strip()   # explicit form: never mistaken for a mistyped file name
        if not name:
            raise ValueError("empty CelesTrak group name")
        return _download(celestrak_url(group=name), fetch, allow_network)
    if one_line and name and all(c.isalnum() or c in "-_" for c in name):
        return _download(celestrak_url(group=name), fetch, allow_network)
    if one_line and name and " " not in name and ("." in name or os.sep in name):
        raise FileNotFoundError(name)   # looks like a path (catalog.tle, data/active.txt) and is not there: not a group name
    raise ValueError("source is neither TLE text, OMM JSON, an existing local file, a URL nor a CelesTrak group name")


def _is_json(text):
    return text.lstrip()[:1] in ("{", "[")


def _jd_of(start_time):
    """Julian date of a datetime; None means now (UTC)."""
    when = datetime.now(timezone.utc) if start_time is None else start_time
    return _UNIX_EPOCH_JD + when.timestamp() / 86400.0


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


class Sgp4Constellation:
    """``_astroz.Sgp4Constellation`` (bindings/python/astroz/_astroz.pyi L501-630): the extension type behind the
    reference's high-level functions, same method names and keywords, device-resident elements.

    Differences that are deliberate: deep-space members are accepted and propagated (the reference's type
    rejects them); ``satellite_mask`` may be bool or uint8."""

    def __init__(self, dev):
        self._dev = dev

    @staticmethod
    def from_tle_text(tle_text, gravity_model=WGS84, device=0):
        """WGS84 is the reference's default for this constructor (bindings/python/src/sgp4.zig L293-300)."""
        return Sgp4Constellation(_native.DeviceConstellation.from_tle_text(tle_text, gravity_model, device))

    @staticmethod
    def from_omm_json(json_text, gravity_model=WGS84, device=0):
        return Sgp4Constellation(_native.DeviceConstellation.from_omm_json(json_text, gravity_model, device))

    @property
    def num_satellites(self):
        return int(self._dev.n)

    @property
    def epochs(self):
        return [float(x) for x in self._dev.epochs]

    def propagate_into(self, times, positions, velocities=None, *, epoch_offsets=None, satellite_mask=None,
                       output="ecef", reference_jd=0.0, time_major=True, output_stride=-1):
        """Propagate into caller-owned arrays (sgp4.zig L170-262): ``(n_times, stride, 3)`` if `time_major`
        else ``(n_sats, n_times, 3)``; ``output_stride`` > 0 overrides the row length of the time-major
        layout; raises ValueError when an array is too small."""
        if output not in _native.OUTPUT_MODES:
            raise ValueError("output must be 'ecef', 'teme', or 'geodetic'")
        mask = None if satellite_mask is None else np.ascontiguousarray(satellite_mask).astype(np.uint8)
        self._dev.propagate_host(times, epoch_offsets, pos=positions, vel=velocities,
                                 mode=_native.OUTPUT_MODES[output], reference_jd=float(reference_jd), mask=mask,
                                 layout=_native.TIME_MAJOR if time_major else _native.SAT_MAJOR,
                                 stride=int(output_stride) if output_stride and output_stride > 0 else 0)

    def screen_conjunction(self, times, target, threshold, *, epoch_offsets=None, reference_jd=0.0):
        """Fused propagate + single-target screen: ``(min_distances, min_t_indices)`` as lists."""
        d, ti = self._dev.screen_target(times, int(target), float(threshold), epoch_offsets, reference_jd=reference_jd)
        return [float(x) for x in d], [int(x) for x in ti]


class Constellation:
    """Pre-parsed, device-resident orbital elements for repeated propagation and screening.

    ``source``: raw TLE text, raw OMM JSON (object or array), a local file holding either, or -- with ``fetch=`` /
    ``allow_network=True`` / :func:`set_fetcher` -- a URL or CelesTrak group name; ``norad_id=`` likewise.
    Output ordering follows the reference: near-earth satellites first ``[0, n_sgp4)``, deep-space after
    (reference __init__.py L374-393) -- but the deep-space rows ARE propagated here (the reference leaves them
    unwritten, L509-530).  The gravity model is WGS84, the default of the reference's ``from_tle_text``."""

    def __init__(self, source=None, *, norad_id=None, gravity_model=WGS84, device=0, fetch=None, allow_network=False):
        text = _as_text(source, norad_id, fetch, allow_network)
        build = _native.DeviceConstellation.from_omm_json if _is_json(text) else _native.DeviceConstellation.from_tle_text
        try:
            full = build(text, gravity_model, device)
        except _native.NativeError as exc:
            if exc.code == _native.AZ_ERR_HIP:
                raise
            raise ValueError("no usable element sets found in source") from exc
        err, deep, _ = full.status
        if err.any():
            bad = int(np.flatnonzero(err)[0])
            raise ValueError("%s (record %d)" % ("Invalid eccentricity" if err[bad] == 1 else "Satellite decayed", bad))
        # catalog index of every output row: near-earth members first, deep-space members after
        self._catalog_index = np.concatenate([np.flatnonzero(~deep), np.flatnonzero(deep)]).astype(np.uint32)
        if np.array_equal(self._catalog_index, np.arange(full.n, dtype=np.uint32)):
            self._dev = full
        else:
            self._dev = full.subset(self._catalog_index)  # device-side re-order: no second parse
            full.close()
        self._n_sgp4 = int((~deep).sum())
        self._total_sats = int(self._dev.n)

    @property
    def num_satellites(self):
        return self._total_sats

    @property
    def epochs(self):
        """TLE epoch of each satellite (output order) as a Julian date."""
        return [float(x) for x in self._dev.epochs]

    @property
    def catalog_index(self):
        """Position in the source text of every output row (near-earth first reorders mixed catalogs)."""
        return self._catalog_index.copy()


def _minutes_and_offsets(const, times, start_time):
    minutes = np.ascontiguousarray(times, dtype=np.float64)
    start = _jd_of(start_time)
    return minutes, (start - const._dev.epochs) * 1440.0, start


def propagate(source, times, *, start_time=None, output="ecef", velocities=False, norad_id=None, fetch=None,
              allow_network=False):
    """Propagate satellites to ``times`` (minutes from ``start_time``, default now).

    Returns positions ``(n_times, n_satellites, 3)`` [km; or (lat rad, lon rad, alt km) for
    ``output="geodetic"`` -- radians, as the reference's kernel emits (Constellation.zig L497)], plus
    velocities ``(n_times, n_satellites, 3)`` km/s if ``velocities=True``.  Reference: __init__.py L411-532."""
    const = source if isinstance(source, Constellation) else Constellation(source, norad_id=norad_id, fetch=fetch,
                                                                           allow_network=allow_network)
    if output not in _native.OUTPUT_MODES:
        raise ValueError("output must be 'ecef', 'teme', or 'geodetic'")
    minutes, offsets, start = _minutes_and_offsets(const, times, start_time)
    shape = (len(minutes), const.num_satellites, 3)
    pos = _native.result_empty(shape)     # (large results: pinned memory the device-to-host DMA writes directly)
    vel = _native.result_empty(shape) if velocities else None
    const._dev.propagate_host(minutes, offsets, pos=pos, vel=vel, mode=_native.OUTPUT_MODES[output],
                              reference_jd=start, layout=_native.TIME_MAJOR)
    return (pos, vel) if velocities else pos


def screen(source, times, threshold=10.0, *, target=None, start_time=None, norad_id=None, fetch=None, allow_network=False):
    """Screen a constellation for conjunction events (reference __init__.py L535-658).

    ``target`` set: fused propagate+screen on the GPU against that satellite; returns
    ``(min_distances (n_sats,) float64 km, min_t_indices (n_sats,) uint32)`` -- a satellite that never
    comes within ``threshold`` (and the target itself) reports ``threshold`` and index 0.
    ``target`` None: all-vs-all; returns ``(pairs (k, 2) uint32, t_indices (k,) uint32)`` for every pair
    closer than ``threshold`` km at a grid time, sorted by (t, s, other).  Positions stay in HBM.
    Deep-space members take part in both modes (the reference's fused routine covers pure-SGP4
    constellations only and falls back to propagate-then-screen otherwise, L655-658)."""
    const = source if isinstance(source, Constellation) else Constellation(source, norad_id=norad_id, fetch=fetch,
                                                                           allow_network=allow_network)
    minutes, offsets, start = _minutes_and_offsets(const, times, start_time)
    if target is not None:
        return const._dev.screen_target(minutes, int(target), float(threshold), offsets, reference_jd=start)
    return const._dev.screen_all(minutes, float(threshold), offsets)


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


# ---- the four closed-form orbital scalars of the reference's module surface (bindings/python/src/main.zig L24-32 over
# src/calculations.zig L83-125) that libastroz_hip.so carries for C clients (orbital_*): same names, arguments and error
# behaviour.  bi_elliptic_transfer, lambert and propagate_numerical are mission analysis, off the propagation path.
EARTH_MU = 398600.5          # km^3/s^2, WGS84 (src/constants.zig L41-52)
EARTH_R_EQ = 6378.137        # km
EARTH_J2 = 0.00108262998905
SUN_MU = 1.32712e11          # km^3/s^2 (src/constants.zig L94-97; module constants of the reference's extension, main.zig L73-74)
MOON_MU = 4.90280e3          # km^3/s^2 (src/constants.zig L167-170)


def hohmann_transfer(mu, r1, r2):
    """hohmann_transfer(mu, r1, r2) -> dict(sma, dv1, dv2, total_dv, transfer_time, transfer_time_days); ValueError for
    non-positive radii or radii closer than 1,000 km (orbital_mechanics.zig L9-19)."""
    import ctypes as C

    class _H(C.Structure):
        _fields_ = [(k, C.c_double) for k in ("sma", "dv1", "dv2", "total_dv", "transfer_time", "transfer_time_days")]
    h = _H()
    rc = _native.lib().orbital_hohmann(float(mu), float(r1), float(r2), C.byref(h))
    if rc != 0:
        raise ValueError("invalid transfer parameters (radii must be positive and differ by >1000 km)")
    return {k: getattr(h, k) for k, _ in _H._fields_}


def _scalar(v, what, bad):
    if v < 0:   # (the c_api's -1.0 for an invalid radius / semi-major axis)
        raise ValueError(bad)
    return v


def orbital_velocity(mu, radius, sma=None):
    """Vis-viva speed; circular if `sma` is omitted."""
    return _scalar(_native.lib().orbital_velocity(float(mu), float(radius), 0.0 if sma is None else float(sma)),
                   "orbital_velocity", "invalid radius / semi-major axis")


def orbital_period(mu, sma):
    """Period in seconds (Kepler's third law)."""
    return _scalar(_native.lib().orbital_period(float(mu), float(sma)), "orbital_period", "invalid semi-major axis")


def escape_velocity(mu, radius):
    return _scalar(_native.lib().orbital_escape_velocity(float(mu), float(radius)), "orbital_escape_velocity", "invalid radius")


__all__ = ["__version__", "Tle", "Sgp4Constellation", "Constellation", "propagate", "screen", "coarse_screen",
           "set_fetcher", "celestrak_url", "WGS72", "WGS84", "hohmann_transfer", "orbital_velocity", "orbital_period",
           "escape_velocity", "EARTH_MU", "EARTH_R_EQ", "EARTH_J2", "SUN_MU", "MOON_MU"]
# (the reference's package also re-exports bi_elliptic_transfer, lambert and propagate_numerical -- its orbital-mechanics and
# numerical-integration modules, outside the SGP4/SDP4 constellation path this package replaces: DESIGN.md 9)
