"""CPU tier, round 4: the quasi-uniform (jd, fr) grids of the reference's own API on the branch-free step.

`SatrecArray.sgp4(jd, fr)` hands over times = ((jd + fr) - reference_jd) * 1440 (bindings/python/astroz/api.py L300-302;
src/Constellation.zig L266-269): jd + fr at 2.46e6 days is quantised to 2^-31 day, so the grid is uniform only to ~4e-7 min.
The fast kernels run along the ideal grid and correct every point to its ACTUAL time to first order (fast_step.h, DELTA).
Here: the device step compiled for the host (tests/host_emul) against the oracle evaluated at the rounded times.
"""
import ctypes as C
import os

import numpy as np
import pytest

from test_host_cpu import _grav6  # noqa: F401  (helper; the emul fixture lives in conftest.py)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def reference_grids(n, start_jd):
    """The three (jd, fr) forms the reference's examples / benchmarks use (examples/python_sgp4.py L31-33, the same from a
    jd = x.5 start, benchmarks/sgp4_compat_test.py L120-133)."""
    return {
        "example": (np.full(n, start_jd), 0.32853009 + np.arange(n) / 1440.0),
        "midnight": (np.full(n, start_jd), np.arange(n) / 1440.0),
        "linspace": (np.full(n, start_jd), np.linspace(0.0, 1.0, n)),
    }


def quasi_uniform(times):
    """Mirror of stage_inputs (astroz_hip.hip): ideal grid through the end points, deviations as fp32."""
    n = len(times)
    t0 = times[0]
    step = (times[-1] - t0) / (n - 1)
    delta = times - (np.arange(n) * step + t0)
    return t0, step, delta.astype(np.float32), float(np.abs(delta).max())


@pytest.mark.parametrize("grid", ["example", "midnight", "linspace"])
def test_emulated_fast_step_on_reference_jd_fr_grids(emul, orc, grid):
    from astroz_amd import synth
    E = emul
    E.emul_rows_fast.argtypes = [C.c_void_p, C.c_uint, C.c_void_p, C.c_double, C.c_double, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    E.emul_rows_fast_delta.argtypes = [C.c_void_p, C.c_uint, C.c_void_p, C.c_double, C.c_double, C.c_int, C.c_int, C.c_int,
                                       C.c_void_p, C.c_double, C.c_void_p, C.c_void_p]
    pairs = synth.synth_catalog(120, 0, seed=21)
    tles = [orc.parse_lines(a, b) for a, b in pairs]
    cat = orc.Catalog(tles, 1)
    g = _grav6(1)
    nf = E.emul_num_fields()
    n = 1440
    jd, fr = reference_grids(n, synth.START_JD)[grid]
    ref = jd[0] + fr[0]
    times = ((jd + fr) - ref) * 1440.0
    off = (ref - cat.epoch_jd) * 1440.0
    t0, step, delta, dmax = quasi_uniform(times)
    # the grids the reference's own callers produce are NOT uniform to rounding, and are inside the fast path's bound
    assert 1e-8 < dmax < 4.0e-6, dmax
    assert dmax > 100 * 4 * 2.2e-16 * 1440
    _, p0, v0 = cat.propagate(times, off)
    worst_r = worst_v = plain_r = 0.0
    accepted = 0
    for i, t in enumerate(tles):
        raw = np.array([t.epoch_jd, t.mm_revday, t.ecc, t.incl_deg, t.raan_deg, t.argp_deg, t.ma_deg, t.bstar])
        fields = np.zeros(nf)
        flags = E.emul_init(raw.ctypes.data, g.ctypes.data, fields.ctypes.data)
        ecc = 1 if t.ecc > 0.003 else 0
        out = np.zeros((n, 6))
        bad = np.zeros(n, dtype=np.int32)
        E.emul_rows_fast_delta(fields.ctypes.data, flags, g.ctypes.data, t0 + off[i], step, n, 768, ecc, delta.ctypes.data, dmax,
                               out.ctypes.data, bad.ctypes.data)
        ok = bad == 0
        accepted += int(ok.sum())
        if ok.any():
            worst_r = max(worst_r, np.abs(out[ok, :3] - p0[i][ok]).max())
            worst_v = max(worst_v, np.abs(out[ok, 3:] - v0[i][ok]).max())
        # the same step WITHOUT the correction (the ideal grid taken for the real one) misses by decimetres: the test can tell
        out2 = np.zeros((n, 6))
        E.emul_rows_fast(fields.ctypes.data, flags, g.ctypes.data, t0 + off[i], step, n, 768, ecc, out2.ctypes.data, bad.ctypes.data)
        ok2 = bad == 0
        if ok2.any():
            plain_r = max(plain_r, np.abs(out2[ok2, :3] - p0[i][ok2]).max())
    assert accepted > 0.9 * n * len(tles), accepted
    assert worst_r < 1e-6 and worst_v < 1e-9, (worst_r, worst_v)
    assert plain_r > 20 * worst_r and plain_r > 2e-5, (plain_r, worst_r)


def test_emulated_mixed_fp32_step_on_a_jd_fr_grid(emul, orc):
    """az_sgp4_fast_step_f32p with the deviations of a (jd, fr) grid: inside the mixed step's 0.6 m / 0.6 mm/s."""
    from astroz_amd import synth
    E = emul
    E.emul_propagate_fast32p_delta.argtypes = [C.c_void_p, C.c_uint, C.c_void_p, C.c_double, C.c_double, C.c_int, C.c_int, C.c_void_p,
                                               C.c_void_p, C.c_void_p]
    pairs = synth.synth_catalog(60, 0, seed=5)
    tles = [orc.parse_lines(a, b) for a, b in pairs]
    cat = orc.Catalog(tles, 1)
    g = _grav6(1)
    nf = E.emul_num_fields()
    n_grid = 1536
    jd, fr = np.full(n_grid, synth.START_JD), 0.32853009 + np.arange(n_grid) / 1440.0
    ref = jd[0] + fr[0]
    times = ((jd + fr) - ref) * 1440.0
    off = (ref - cat.epoch_jd) * 1440.0
    t0, step, delta, dmax = quasi_uniform(times)
    lane_steps, n = 128, n_grid // 128
    idx = (np.arange(n)[:, None] * lane_steps + np.array([0, 1])[None, :]).ravel()  # the lane's points: (128 j, 128 j + 1)
    d2 = np.ascontiguousarray(delta[idx])
    _, p0, v0 = cat.propagate(times[idx], off)
    worst_r = worst_v = 0.0
    used = 0
    for i, t in enumerate(tles):
        if t.ecc > 0.003:
            continue
        raw = np.array([t.epoch_jd, t.mm_revday, t.ecc, t.incl_deg, t.raan_deg, t.argp_deg, t.ma_deg, t.bstar])
        fields = np.zeros(nf)
        flags = E.emul_init(raw.ctypes.data, g.ctypes.data, fields.ctypes.data)
        out = np.zeros((2 * n, 6))
        bad = np.zeros(n, dtype=np.int32)
        E.emul_propagate_fast32p_delta(fields.ctypes.data, flags, g.ctypes.data, t0 + off[i], step, lane_steps, n, d2.ctypes.data,
                                       out.ctypes.data, bad.ctypes.data)
        ok = np.repeat(bad == 0, 2)
        if not ok.any():
            continue
        used += 1
        worst_r = max(worst_r, np.abs(out[ok, :3] - p0[i][ok]).max())
        worst_v = max(worst_v, np.abs(out[ok, 3:] - v0[i][ok]).max())
    assert used > 30
    assert worst_r < 6e-4 and worst_v < 6e-7, (worst_r, worst_v)


def test_emulated_fast_step_on_a_jittered_grid(emul, orc):
    """The WIDE form (fast_step.h, DELTA = 2): a one-minute grid whose points are off by up to +-20 s (time stamps of a periodic
    process).  Least-squares ideal grid, fp64 deviations, M rotated to first order (near-circular) / by the 1/16-rad
    polynomial (eccentric), W to first order, U exactly -- against the oracle at the actual times, at the fp64 gate."""
    from astroz_amd import synth
    E = emul
    E.emul_rows_fast_wide.argtypes = [C.c_void_p, C.c_uint, C.c_void_p, C.c_double, C.c_double, C.c_int, C.c_int, C.c_int,
                                      C.c_void_p, C.c_double, C.c_void_p, C.c_void_p]
    pairs = synth.synth_catalog(150, 0, seed=31)
    tles = [orc.parse_lines(a, b) for a, b in pairs]
    cat = orc.Catalog(tles, 1)
    g = _grav6(1)
    nf = E.emul_num_fields()
    n = 1440
    rng = np.random.default_rng(12)
    times = np.arange(n, dtype=np.float64) + rng.uniform(-1.0 / 3.0, 1.0 / 3.0, n)
    off = (synth.START_JD - cat.epoch_jd) * 1440.0
    idx = np.arange(n, dtype=np.float64)
    step, t0 = np.polyfit(idx, times, 1)            # the least-squares line stage_inputs fits
    delta = times - (idx * step + t0)
    dmax = float(np.abs(delta).max())
    assert 0.2 < dmax < 0.5
    _, p0, v0 = cat.propagate(times, off)
    worst_r = worst_v = 0.0
    accepted = n_ecc = 0
    for i, t in enumerate(tles):
        raw = np.array([t.epoch_jd, t.mm_revday, t.ecc, t.incl_deg, t.raan_deg, t.argp_deg, t.ma_deg, t.bstar])
        fields = np.zeros(nf)
        flags = E.emul_init(raw.ctypes.data, g.ctypes.data, fields.ctypes.data)
        ecc = 1 if t.ecc > 0.003 else 0
        n_ecc += ecc
        out = np.zeros((n, 6))
        bad = np.zeros(n, dtype=np.int32)
        E.emul_rows_fast_wide(fields.ctypes.data, flags, g.ctypes.data, t0 + off[i], step, n, 768, ecc, delta.ctypes.data, dmax,
                              out.ctypes.data, bad.ctypes.data)
        ok = bad == 0
        accepted += int(ok.sum())
        if ok.any():
            worst_r = max(worst_r, np.abs(out[ok, :3] - p0[i][ok]).max())
            worst_v = max(worst_v, np.abs(out[ok, 3:] - v0[i][ok]).max())
    assert n_ecc >= 5
    assert accepted > 0.85 * n * len(tles), accepted
    assert worst_r < 1e-6 and worst_v < 1e-9, (worst_r, worst_v)


def test_source_resolution_round4():
    """CelesTrak group aliases of the reference's loader (bindings/python/astroz/__init__.py L131-136), the explicit
    'celestrak:' prefix, and the conflicts the advisor flagged (source + norad_id; a mistyped file name is not a group)."""
    import astroz_amd as az
    for short, group in (("all", "active"), ("iss", "stations"), ("gps", "gps-ops"), ("glonass", "glo-ops"), ("Starlink", "starlink")):
        assert az.celestrak_url(group=short).endswith("GROUP=%s&FORMAT=tle" % group)
    seen = []
    fetch = lambda url: seen.append(url) or "1 x\n2 y\n"
    az._as_text("celestrak:iss", fetch=fetch)
    az._as_text("gps", fetch=fetch)
    assert seen[0].endswith("GROUP=stations&FORMAT=tle") and seen[1].endswith("GROUP=gps-ops&FORMAT=tle")
    az._as_text("starlink", norad_id=25544, fetch=fetch)              # both given: the catalog number wins (reference L163-166)
    assert "CATNR=25544" in seen[2]
    with pytest.raises(ValueError):
        az._as_text("celestrak:", fetch=fetch)
    with pytest.raises(FileNotFoundError):
        az._as_text("catalog.tle", fetch=fetch)                       # a file name that does not exist is not a group name
    with pytest.raises(FileNotFoundError):
        az._as_text(os.path.join("data", "active.txt"), fetch=fetch)
    assert len(seen) == 3


def test_orbital_scalars_need_no_device():
    """The four closed-form scalars are host functions, as in the reference (src/c_api/root.zig L60-71; round 5: rounds 2-4
    evaluated them in a one-thread kernel and raised NativeError without a device): with or without a GPU the Python wrappers
    return the reference's values (src/calculations.zig L83-125), and argument errors are ValueError."""
    import astroz_amd as az
    assert abs(az.orbital_velocity(az.EARTH_MU, 7000.0) - (az.EARTH_MU / 7000.0) ** 0.5) < 1e-12
    assert abs(az.orbital_period(az.EARTH_MU, 7000.0) - 2 * 3.141592653589793 * (7000.0 ** 3 / az.EARTH_MU) ** 0.5) < 1e-9
    assert abs(az.escape_velocity(az.EARTH_MU, 7000.0) - (2 * az.EARTH_MU / 7000.0) ** 0.5) < 1e-12
    assert abs(az.hohmann_transfer(az.EARTH_MU, 7000.0, 42164.0)["sma"] - 0.5 * (7000.0 + 42164.0)) < 1e-9
    with pytest.raises(ValueError):
        az.orbital_velocity(az.EARTH_MU, -1.0)                        # argument errors stay ValueError either way
    with pytest.raises(ValueError):
        az.hohmann_transfer(az.EARTH_MU, 7000.0, 7000.5)
