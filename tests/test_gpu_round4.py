"""GPU parity tests (-m gpu), fourth batch.

1. The reference's OWN (jd, fr) time grids -- `times = ((jd + fr) - reference_jd) * 1440` (bindings/python/astroz/api.py
   L300-302, src/Constellation.zig L266-269), uniform only to ~4e-7 min after the rounding of jd + fr -- run the
   branch-free kernels (k_rows_fast / k_tiles_fast in their DELTA form), proven through azh_last_path, with parity against
   the oracle AT THE ROUNDED TIMES on every row of BASELINE configs 2 and 3, both layouts.
2. The resonance node cache of the deep-space seeding is invisible in the results.
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOL_R, TOL_V = 1e-6, 1e-9   # km, km/s (north_star: <10 m, <1 um/s)


@pytest.fixture(scope="module")
def native():
    import __graft_entry__ as g
    g.build()
    from astroz_amd import _native
    assert _native.device_count() >= 1, "no HIP device: GPU tests must run on the MI355X box"
    return _native


@pytest.fixture(scope="module")
def synth():
    from astroz_amd import synth
    return synth


def jdfr_grid(kind, n, start_jd):
    """The (jd, fr) forms of the reference's callers: examples/python_sgp4.py L31-33 ('example'), the same from the top of
    a day ('midnight'), benchmarks/sgp4_compat_test.py L120-133 ('linspace')."""
    jd = np.full(n, start_jd)
    fr = {"example": 0.32853009 + np.arange(n) / 1440.0,
          "midnight": np.arange(n) / 1440.0,
          "linspace": np.linspace(0.0, 1.0, n)}[kind]
    return jd, fr


def api_times(jd, fr, epochs):
    """SatrecArray._grid = api.py L300-302."""
    reference_jd = jd[0] + fr[0]
    return ((jd + fr) - reference_jd) * 1440.0, (reference_jd - epochs) * 1440.0, reference_jd


def test_reference_grids_are_only_quasi_uniform():
    """The premise: none of the reference's canonical grids is uniform to rounding, all are inside AZ_DELTA_MAX."""
    for kind in ("example", "midnight", "linspace"):
        jd, fr = jdfr_grid(kind, 1440, 2460800.5)
        t, _, _ = api_times(jd, fr, np.zeros(1))
        step = (t[-1] - t[0]) / (len(t) - 1)
        dev = np.abs(t - (t[0] + np.arange(len(t)) * step)).max()
        assert 4 * 2.3e-16 * 1440 < 1e-8 < dev < 4e-6, (kind, dev)


@pytest.mark.parametrize("n_deep,kind", [(0, "example"), (0, "linspace"), (1522, "example"), (1522, "midnight")])
def test_jdfr_grids_full_size_all_rows_take_the_fast_kernels(native, orc, synth, n_deep, kind):
    """BASELINE configs 2 and 3 at FULL size on the reference's own (jd, fr) grids: every row vs the oracle at the rounded
    times, both layouts, and azh_last_path shows the branch-free kernels in their quasi-uniform form."""
    import torch
    pairs = synth.synth_catalog(n_near=13478, n_deep=n_deep)
    dev = native.DeviceConstellation.from_tle_lines(pairs, 1, 0)
    cat = orc.Catalog.from_pairs(pairs, 1)
    n = 1440
    jd, fr = jdfr_grid(kind, n, synth.START_JD)
    times, off, _ = api_times(jd, fr, dev.epochs)
    e0, p0, v0 = cat.propagate(times, off, layout=orc.SAT_MAJOR, threads=16)
    pos = torch.empty((dev.n, n, 3), dtype=torch.float64, device="cuda")
    vel = torch.empty_like(pos)
    err = torch.empty((dev.n, n), dtype=torch.uint8, device="cuda")
    dev.propagate_device(times, off, pos.data_ptr(), vel.data_ptr(), layout=native.SAT_MAJOR, d_err=err.data_ptr())
    dev.synchronize()
    path = dev.last_path()
    assert path & native.PATH_ROWS_FAST and path & native.PATH_QUASI_UNIFORM, path
    assert not path & (native.PATH_ROWS_GENERIC | native.PATH_LANE_SAT), path
    assert np.array_equal(err.cpu().numpy(), e0)
    dr = float(np.abs(pos.cpu().numpy() - p0).max())
    dv = float(np.abs(vel.cpu().numpy() - v0).max())
    assert dr < TOL_R and dv < TOL_V, (dr, dv)
    del pos, vel
    ptm = torch.empty((n, dev.n, 3), dtype=torch.float64, device="cuda")
    vtm = torch.empty_like(ptm)
    dev.propagate_device_cached(ptm.data_ptr(), vtm.data_ptr(), layout=native.TIME_MAJOR)
    dev.synchronize()
    path = dev.last_path()
    assert path & (native.PATH_TILES_FAST | native.PATH_COLS_FAST) and path & native.PATH_QUASI_UNIFORM, path
    assert not path & native.PATH_LANE_SAT, path
    dr = float(np.abs(ptm.cpu().numpy().transpose(1, 0, 2) - p0).max())
    dv = float(np.abs(vtm.cpu().numpy().transpose(1, 0, 2) - v0).max())
    assert dr < TOL_R and dv < TOL_V, (dr, dv)


def test_jdfr_grid_matches_the_generic_path_and_ecef(native, orc, synth):
    """The quasi-uniform form against the generic kernels on the same grid (fast path off), ECEF / geodetic output, a grid
    running backwards, positions only, and a grid too ragged for the fast path (falls back, still correct)."""
    pairs = synth.synth_catalog(n_near=900, n_deep=80, seed=33)
    dev = native.DeviceConstellation.from_tle_lines(pairs, 1, 0)
    cat = orc.Catalog.from_pairs(pairs, 1)
    n = 700
    jd, fr = jdfr_grid("example", n, synth.START_JD + 3.0)
    times, off, ref = api_times(jd, fr, dev.epochs)
    for layout, olay in ((native.SAT_MAJOR, orc.SAT_MAJOR), (native.TIME_MAJOR, orc.TIME_MAJOR)):
        for mode, omode in ((native.OUT_TEME, orc.TEME), (native.OUT_ECEF, orc.ECEF), (native.OUT_GEODETIC, orc.GEODETIC)):
            shape = (dev.n, n, 3) if layout == native.SAT_MAJOR else (n, dev.n, 3)
            pos, vel = np.empty(shape), np.empty(shape)
            dev.propagate_host(times, off, pos=pos, vel=vel, mode=mode, reference_jd=ref, layout=layout)
            path = dev.last_path()
            assert path & native.PATH_QUASI_UNIFORM and path & (native.PATH_ROWS_FAST | native.PATH_TILES_FAST | native.PATH_COLS_FAST), (layout, mode, path)
            _, p0, v0 = cat.propagate(times, off, layout=olay, mode=omode, reference_jd=ref)
            dq = np.abs(pos - p0)
            if mode == native.OUT_GEODETIC:
                dq[..., 1] = np.minimum(dq[..., 1], 2 * np.pi - dq[..., 1])   # longitude wraps
            assert dq.max() < TOL_R, (layout, mode, dq.max())
            assert np.abs(vel - v0).max() < TOL_V
    # fast path off: the generic kernels on the very same grid
    pos, vel = np.empty((dev.n, n, 3)), np.empty((dev.n, n, 3))
    dev.propagate_host(times, off, pos=pos, vel=vel, layout=native.SAT_MAJOR)
    dev.set_fast_path(False)
    pos2, vel2 = np.empty_like(pos), np.empty_like(vel)
    dev.propagate_host(times, off, pos=pos2, vel=vel2, layout=native.SAT_MAJOR)
    assert not dev.last_path() & (native.PATH_ROWS_FAST | native.PATH_QUASI_UNIFORM)
    dev.set_fast_path(True)
    assert np.abs(pos - pos2).max() < 2e-7 and np.abs(vel - vel2).max() < 2e-10
    # backwards in time, positions only
    tb = times[::-1].copy()
    pos = np.empty((dev.n, n, 3))
    dev.propagate_host(tb, off, pos=pos, layout=native.SAT_MAJOR)
    assert dev.last_path() & native.PATH_QUASI_UNIFORM
    _, p0, _ = cat.propagate(tb, off, velocities=False)
    assert np.abs(pos - p0).max() < TOL_R
    # random times are not a uniform grid in any sense: generic kernels, same answers
    rng = np.random.default_rng(4)
    tj = np.sort(times[0] + rng.uniform(0.0, 700.0, n))
    pos, vel = np.empty((dev.n, n, 3)), np.empty((dev.n, n, 3))
    dev.propagate_host(tj, off, pos=pos, vel=vel, layout=native.SAT_MAJOR)
    assert not dev.last_path() & (native.PATH_ROWS_FAST | native.PATH_QUASI_UNIFORM)
    _, p0, v0 = cat.propagate(tj, off)
    assert np.abs(pos - p0).max() < TOL_R and np.abs(vel - v0).max() < TOL_V


def test_satrec_array_jdfr_runs_the_tile_kernel(native, orc, synth):
    """The call the reference's headline is quoted on: SatrecArray.sgp4(jd, fr) / sgp4_device(jd, fr)."""
    from astroz_amd.api import Satrec, SatrecArray
    pairs = synth.synth_catalog(n_near=1000, n_deep=0, seed=8)
    arr = SatrecArray([Satrec.twoline2rv(a, b) for a, b in pairs])
    cat = orc.Catalog.from_pairs(pairs, 1)
    n = 1440
    jd, fr = jdfr_grid("linspace", n, synth.START_JD)
    times, off, _ = api_times(jd, fr, cat.epoch_jd)
    e0, p0, v0 = cat.propagate(times, off)
    e, r, v = arr.sgp4(jd, fr)
    path = arr._dev.last_path()
    assert path & (native.PATH_TILES_FAST | native.PATH_COLS_FAST) and path & native.PATH_QUASI_UNIFORM and not path & native.PATH_LANE_SAT, path
    assert np.array_equal(e, e0) and np.abs(r - p0).max() < TOL_R and np.abs(v - v0).max() < TOL_V
    e, r_tm, v_tm = arr.sgp4_device(jd, fr)
    arr.synchronize()
    path = arr._dev.last_path()
    assert path & (native.PATH_TILES_FAST | native.PATH_COLS_FAST) and path & native.PATH_QUASI_UNIFORM, path
    assert np.abs(r_tm.cpu().numpy().transpose(1, 0, 2) - p0).max() < TOL_R
    assert np.abs(v_tm.cpu().numpy().transpose(1, 0, 2) - v0).max() < TOL_V
    assert r_tm.is_contiguous() and v_tm.is_contiguous() and tuple(r_tm.shape) == (n, 1000, 3)   # dense by default
    e, r_pad, v_pad = arr.sgp4_device(jd, fr, padded=True)    # opt-in: time rows padded to 16 satellites, zero padding
    arr.synchronize()
    assert not r_pad.is_contiguous() and r_pad.stride(0) == 1008 * 3
    assert bool((r_pad == r_tm).all()) and bool((v_pad == v_tm).all())
    assert float(r_pad._base[:, 1000:].abs().max()) == 0.0 if r_pad._base is not None else True


def test_jdfr_grid_fp32_and_screen(native, orc, synth):
    """fp32 outputs (mixed and packed arithmetic) and the fused screen on a (jd, fr) grid."""
    import torch
    pairs = synth.synth_catalog(n_near=700, n_deep=0, seed=14)
    dev = native.DeviceConstellation.from_tle_lines(pairs, 1, 0)
    cat = orc.Catalog.from_pairs(pairs, 1)
    n = 1536
    jd, fr = jdfr_grid("example", n, synth.START_JD)
    times, off, _ = api_times(jd, fr, dev.epochs)
    _, p0, v0 = cat.propagate(times, off)
    for mode, tr, tv in (("mixed", 6.0e-4, 6.0e-7), ("packed", 4.0e-3, 6.0e-6), ("fp64", 4.95e-4, 4.85e-7)):
        dev.set_f32_arithmetic(mode)
        pos = torch.empty((dev.n, n, 3), dtype=torch.float32, device="cuda")
        vel = torch.empty_like(pos)
        dev.propagate_device(times, off, pos.data_ptr(), vel.data_ptr(), layout=native.SAT_MAJOR, f32=True)
        dev.synchronize()
        assert dev.last_path() & native.PATH_ROWS_FAST and dev.last_path() & native.PATH_QUASI_UNIFORM
        dr = float(np.abs(pos.cpu().numpy().astype(np.float64) - p0).max())
        dv = float(np.abs(vel.cpu().numpy().astype(np.float64) - v0).max())
        assert dr < tr and dv < tv, (mode, dr, dv)
    dev.set_f32_arithmetic("mixed")
    d, t = dev.screen_target(times, 5, 2000.0, off)
    assert dev.last_path() & native.PATH_ROWS_FAST and dev.last_path() & native.PATH_QUASI_UNIFORM
    d0, t0 = cat.screen_target(times, 5, 2000.0, off)
    assert np.array_equal(t, t0) and np.abs(d - d0).max() < TOL_R


def test_resonance_node_cache_is_invisible(native, orc, synth):
    """k_deep_seed continues from the integrator node the previous grid left in the handle (the nodes depend on the satellite
    only).  Whatever the history -- later grid, earlier grid (inside the cached node: restart), other side of epoch,
    different offsets -- a handle with history and a fresh handle agree BIT FOR BIT, and with the oracle."""
    import torch
    pairs = synth.synth_catalog(n_near=40, n_deep=700, seed=77)
    cat = orc.Catalog.from_pairs(pairs, 1)
    used = native.DeviceConstellation.from_tle_lines(pairs, 1, 0)
    off = (synth.START_JD - used.epochs) * 1440.0
    grids = [np.arange(0.0, 1440.0), 40000.0 + np.arange(0.0, 2880.0, 2.0), 9000.0 + np.arange(0.0, 720.0),
             -30000.0 + 3.0 * np.arange(0.0, 500.0), 40000.0 + np.arange(0.0, 300.0), np.arange(-700.0, 800.0, 1.5)]
    for j, t in enumerate(grids):
        o = off + (1000.0 if j == 4 else 0.0)
        for layout, olay in ((native.SAT_MAJOR, orc.SAT_MAJOR), (native.TIME_MAJOR, orc.TIME_MAJOR)):
            fresh = native.DeviceConstellation.from_tle_lines(pairs, 1, 0)
            shape = (used.n, len(t), 3) if layout == native.SAT_MAJOR else (len(t), used.n, 3)
            res = []
            for dev in (used, fresh):
                pos, vel = np.empty(shape), np.empty(shape)
                err = np.zeros((used.n, len(t)), dtype=np.uint8)
                dev.propagate_host(t, o, pos=pos, vel=vel, err=err, layout=layout)
                res.append((pos, vel, err))
            fresh.close()
            assert np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1]) and np.array_equal(res[0][2], res[1][2]), (j, layout)
            e0, p0, v0 = cat.propagate(t, o, layout=olay)
            assert np.array_equal(res[0][2], e0)
            assert np.abs(res[0][0] - p0).max() < TOL_R and np.abs(res[0][1] - v0).max() < TOL_V, (j, layout)


@pytest.mark.parametrize("n_deep", [0, 1522])
def test_jittered_grids_full_size_take_the_fast_kernels(native, orc, synth, n_deep):
    """A one-minute grid whose points are off by up to +-20 s (time stamps of a periodic process) is 'uniform with jitter': the
    fast kernels run along the least-squares grid and rotate every point to its actual time (fast_step.h, DELTA = 2).
    BASELINE configs 2 and 3 at full size, every row vs the oracle at the actual times, both layouts."""
    import torch
    pairs = synth.synth_catalog(n_near=13478, n_deep=n_deep)
    dev = native.DeviceConstellation.from_tle_lines(pairs, 1, 0)
    cat = orc.Catalog.from_pairs(pairs, 1)
    n = 1440
    times = np.arange(n, dtype=np.float64) + np.random.default_rng(7).uniform(-1.0 / 3.0, 1.0 / 3.0, n)
    off = (synth.START_JD - dev.epochs) * 1440.0
    e0, p0, v0 = cat.propagate(times, off, layout=orc.SAT_MAJOR, threads=16)
    pos = torch.empty((dev.n, n, 3), dtype=torch.float64, device="cuda")
    vel = torch.empty_like(pos)
    err = torch.empty((dev.n, n), dtype=torch.uint8, device="cuda")
    dev.propagate_device(times, off, pos.data_ptr(), vel.data_ptr(), layout=native.SAT_MAJOR, d_err=err.data_ptr())
    dev.synchronize()
    path = dev.last_path()
    assert path & native.PATH_ROWS_FAST and path & native.PATH_QUASI_UNIFORM and not path & native.PATH_ROWS_GENERIC, path
    assert np.array_equal(err.cpu().numpy(), e0)
    dr = float(np.abs(pos.cpu().numpy() - p0).max())
    dv = float(np.abs(vel.cpu().numpy() - v0).max())
    assert dr < TOL_R and dv < TOL_V, (dr, dv)
    del pos, vel
    ptm = torch.empty((n, dev.n, 3), dtype=torch.float64, device="cuda")
    vtm = torch.empty_like(ptm)
    dev.propagate_device_cached(ptm.data_ptr(), vtm.data_ptr(), layout=native.TIME_MAJOR)
    dev.synchronize()
    path = dev.last_path()
    assert path & (native.PATH_TILES_FAST | native.PATH_COLS_FAST) and path & native.PATH_QUASI_UNIFORM and not path & native.PATH_LANE_SAT, path
    dr = float(np.abs(ptm.cpu().numpy().transpose(1, 0, 2) - p0).max())
    dv = float(np.abs(vtm.cpu().numpy().transpose(1, 0, 2) - v0).max())
    assert dr < TOL_R and dv < TOL_V, (dr, dv)


def test_jittered_grid_variants_and_limits(native, orc, synth):
    """The wide form with ECEF / geodetic output, fp32 outputs, the fused screen, a coarse grid with large jitter; and what is
    NOT uniform-with-jitter (jitter above 30 s, or as large as the step, or random times) stays with the generic kernels."""
    import torch
    pairs = synth.synth_catalog(n_near=800, n_deep=60, seed=91)
    dev = native.DeviceConstellation.from_tle_lines(pairs, 1, 0)
    cat = orc.Catalog.from_pairs(pairs, 1)
    off = (synth.START_JD - dev.epochs) * 1440.0
    rng = np.random.default_rng(3)
    n = 900
    tj = -300.0 + 2.0 * np.arange(n) + rng.uniform(-0.45, 0.45, n)          # two-minute steps, +-27 s
    for layout, olay in ((native.SAT_MAJOR, orc.SAT_MAJOR), (native.TIME_MAJOR, orc.TIME_MAJOR)):
        for mode, omode in ((native.OUT_TEME, orc.TEME), (native.OUT_ECEF, orc.ECEF), (native.OUT_GEODETIC, orc.GEODETIC)):
            shape = (dev.n, n, 3) if layout == native.SAT_MAJOR else (n, dev.n, 3)
            pos, vel = np.empty(shape), np.empty(shape)
            dev.propagate_host(tj, off, pos=pos, vel=vel, mode=mode, reference_jd=synth.START_JD, layout=layout)
            path = dev.last_path()
            assert path & native.PATH_QUASI_UNIFORM and path & (native.PATH_ROWS_FAST | native.PATH_TILES_FAST | native.PATH_COLS_FAST), (layout, mode, path)
            _, p0, v0 = cat.propagate(tj, off, layout=olay, mode=omode, reference_jd=synth.START_JD)
            dq = np.abs(pos - p0)
            if mode == native.OUT_GEODETIC:
                dq[..., 1] = np.minimum(dq[..., 1], 2 * np.pi - dq[..., 1])
            assert dq.max() < TOL_R and np.abs(vel - v0).max() < TOL_V, (layout, mode, dq.max())
    # fp32 outputs: the wide form runs the fp64 kernels with rounded stores (half an fp32 ulp), in every arithmetic mode
    _, p0, v0 = cat.propagate(tj, off)
    p32 = torch.empty((dev.n, n, 3), dtype=torch.float32, device="cuda")
    v32 = torch.empty_like(p32)
    dev.propagate_device(tj, off, p32.data_ptr(), v32.data_ptr(), layout=native.SAT_MAJOR, f32=True)
    dev.synchronize()
    assert dev.last_path() & native.PATH_QUASI_UNIFORM
    assert np.abs(p32.cpu().numpy().astype(np.float64) - p0).max() < 4.95e-4 + 2.5e-3  # (deep-space rows reach 42,000 km: half an ulp is 2 m there)
    near = ~cat.is_deep
    assert np.abs(p32.cpu().numpy().astype(np.float64)[near] - p0[near]).max() < 4.95e-4
    assert np.abs(v32.cpu().numpy().astype(np.float64)[near] - v0[near]).max() < 4.85e-7
    # fused screen
    d, t = dev.screen_target(tj, 3, 3000.0, off)
    assert dev.last_path() & native.PATH_QUASI_UNIFORM
    d0, t0 = cat.screen_target(tj, 3, 3000.0, off)
    assert np.array_equal(t, t0) and np.abs(d - d0).max() < TOL_R
    # not uniform-with-jitter: generic kernels, same gate
    for bad_grid in (2.0 * np.arange(n) + rng.uniform(-0.7, 0.7, n),                       # jitter above half a minute
                     0.5 * np.arange(n) + rng.uniform(-0.3, 0.3, n),                       # jitter larger than half the step
                     np.sort(rng.uniform(0.0, 1440.0, n))):                                # random times
        pos, vel = np.empty((dev.n, n, 3)), np.empty((dev.n, n, 3))
        dev.propagate_host(bad_grid, off, pos=pos, vel=vel, layout=native.SAT_MAJOR)
        assert not dev.last_path() & (native.PATH_QUASI_UNIFORM | native.PATH_ROWS_FAST), dev.last_path()
        _, p0, v0 = cat.propagate(bad_grid, off)
        assert np.abs(pos - p0).max() < TOL_R and np.abs(vel - v0).max() < TOL_V


def test_api_hygiene_round4(native, orc, synth):
    """the boolean azh_set_f32_arithmetic keeps its original meaning next to azh_set_f32_mode; azh_group_* take the
    length of epoch_offsets; host copy threads are a switch, not a semantic."""
    import ctypes as C
    import torch
    pairs = synth.synth_catalog(n_near=300, n_deep=20, seed=5)
    dev = native.DeviceConstellation.from_tle_lines(pairs, 1, 0)
    times = np.arange(0.0, 256.0)
    off = (synth.START_JD - dev.epochs) * 1440.0

    def f32(mode):
        dev.set_f32_arithmetic(mode)
        p = torch.empty((dev.n, 256, 3), dtype=torch.float32, device="cuda")
        v = torch.empty_like(p)
        dev.propagate_device(times, off, p.data_ptr(), v.data_ptr(), layout=native.SAT_MAJOR, f32=True)
        dev.synchronize()
        return p.cpu().numpy(), v.cpu().numpy()
    a, b = f32(False), f32("fp64")        # False = fp64 arithmetic rounded once at the store: bit-identical to mode 2
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    a, b = f32(True), f32("packed")
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    a, b = f32(0), f32("mixed")
    assert np.array_equal(a[0], b[0])
    dev.set_f32_arithmetic("mixed")
    # host copy threads: same bytes either way
    big = native.DeviceConstellation.from_tle_lines(synth.synth_catalog(n_near=3000, n_deep=0, seed=6), 1, 0)
    tb = np.arange(0.0, 700.0)
    ob = (synth.START_JD - big.epochs) * 1440.0
    res = []
    for thr in (0, 3, -1):
        native.set_host_copy_threads(thr)
        pos, vel = np.empty((700, big.n, 3)), np.empty((700, big.n, 3))
        err = np.zeros((big.n, 700), dtype=np.uint8)
        big.propagate_host(tb, ob, pos=pos, vel=vel, err=err)
        res.append((pos, vel, err))
    native.set_host_copy_threads(-1)
    for r in res[1:]:
        assert np.array_equal(r[0], res[0][0]) and np.array_equal(r[1], res[0][1]) and np.array_equal(r[2], res[0][2])
    # azh_group_*: a short epoch_offsets array is an error code, not an out-of-bounds read
    text = synth.pairs_to_text(pairs)
    grp = native.DeviceGroup(text, [0], native.WGS72, n_chunks=2)
    L = native.lib()
    t = np.arange(0.0, 64.0)
    short = np.zeros(grp.n - 1)
    pos = np.empty((grp.n, 64, 3))
    rc = L.azh_group_propagate_host(grp._h, t.ctypes.data, 64, short.ctypes.data, len(short), pos.ctypes.data, None, 0, 0.0, None)
    assert rc == -20   # AZ_ERR_VALUE
    full = (synth.START_JD - grp.epochs) * 1440.0
    rc = L.azh_group_propagate_host(grp._h, t.ctypes.data, 64, full.ctypes.data, len(full), pos.ctypes.data, None, 0, 0.0, None)
    assert rc == 0
    _, p0, _ = orc.Catalog.from_pairs(pairs, 1).propagate(t, full, velocities=False)
    assert np.abs(pos - p0).max() < TOL_R
    grp.close()


def test_quasi_uniform_forms_ragged_long_and_windowed(native, orc, synth):
    """The quasi-uniform forms away from the benchmark shape: ragged grid lengths (not a multiple of 64; shorter than a
    segment), a 10,000-step (jd, fr) grid a week long (seven-day windows: the plan's bounds, the redo pass), two-minute and
    backwards steps, row windows (the chunked multi-GPU launches) on a (jd, fr) grid, padded time-major strides."""
    import torch
    pairs = synth.synth_catalog(n_near=400, n_deep=30, seed=23)
    dev = native.DeviceConstellation.from_tle_lines(pairs, 1, 0)
    cat = orc.Catalog.from_pairs(pairs, 1)

    def check(jd, fr, tol_r=TOL_R, tol_v=TOL_V, want_fast=True):
        times, off, _ = api_times(jd, fr, dev.epochs)
        n = len(times)
        e0, p0, v0 = cat.propagate(times, off, threads=8)
        for layout in (native.SAT_MAJOR, native.TIME_MAJOR):
            shape = (dev.n, n, 3) if layout == native.SAT_MAJOR else (n, dev.n, 3)
            pos, vel = np.empty(shape), np.empty(shape)
            err = np.zeros((dev.n, n), dtype=np.uint8)
            dev.propagate_host(times, off, pos=pos, vel=vel, err=err, layout=layout)
            if want_fast and n >= 64:
                assert dev.last_path() & native.PATH_QUASI_UNIFORM, (n, layout, dev.last_path())
            if layout == native.TIME_MAJOR:
                pos, vel = pos.transpose(1, 0, 2), vel.transpose(1, 0, 2)
            assert np.array_equal(err, e0)
            assert np.abs(pos - p0).max() < tol_r and np.abs(vel - v0).max() < tol_v, (n, layout, np.abs(pos - p0).max())

    day = synth.START_JD
    for n in (65, 100, 777, 1441):
        check(np.full(n, day), 0.1234567 + np.arange(n) / 1440.0)
    check(np.full(10000, day - 2.0), 0.5 + np.arange(10000) / 1440.0)                      # a week, one-minute steps
    check(np.full(900, day), 0.75 + np.arange(900) / 720.0)                               # two-minute steps across midnight
    check(np.full(900, day + 1.0), 0.4 - np.arange(900) / 1440.0)                         # backwards
    check(day + np.floor(np.arange(3000) / 1440.0), (np.arange(3000) % 1440) / 1440.0)    # jd steps by whole days, fr wraps
    # row windows on a (jd, fr) grid == one launch, bit for bit, both layouts
    n = 500
    times, off, _ = api_times(np.full(n, day), 0.3 + np.arange(n) / 1440.0, dev.epochs)
    for layout, shape in ((native.SAT_MAJOR, (dev.n, n, 3)), (native.TIME_MAJOR, (n, dev.n + 3, 3))):
        stride = dev.n + 3 if layout == native.TIME_MAJOR else 0
        full = torch.full(shape, float("nan"), dtype=torch.float64, device="cuda")
        part = torch.full_like(full, float("nan"))
        torch.cuda.synchronize()
        dev.propagate_device(times, off, full.data_ptr(), None, layout=layout, stride=stride)
        for lo, hi in ((0, 7), (7, 130), (130, 131), (131, 10**6)):
            dev.propagate_device_window(lo, hi, part.data_ptr(), None, layout=layout, stride=stride)
        dev.synchronize()
        assert dev.last_path() & native.PATH_QUASI_UNIFORM
        assert torch.equal(torch.nan_to_num(full, nan=-1.0), torch.nan_to_num(part, nan=-1.0))


@pytest.mark.parametrize("n_near,n_deep", [(1, 0), (5, 2), (40, 3), (64, 0)])
def test_few_rows_long_series_rotate_over_the_xcds(native, orc, synth, n_near, n_deep):
    """A handful of satellites x a long time series: the lane = time kernels deal their rows out in eight ranges, one per XCD,
    and with at most 64 rows rotate the ranges by time segment (kernels.h az_xcd_row: one satellite x 10^7 times otherwise
    runs on one XCD of eight).  Every (row, segment) pair must still be computed exactly once: all rows x 40,000 steps
    against the oracle, both layouts, exact and (jd, fr) grids, every output element written."""
    pairs = synth.synth_catalog(n_near=n_near, n_deep=n_deep, seed=31)
    dev = native.DeviceConstellation.from_tle_lines(pairs, 1, 0)
    cat = orc.Catalog.from_pairs(pairs, 1)
    n = 40_000
    day = synth.START_JD
    grids = [(np.arange(n) * 0.25, (day - dev.epochs) * 1440.0)]
    t_, o_, _ = api_times(np.full(n, day), 0.25 + np.arange(n) / 5760.0, dev.epochs)
    grids.append((t_, o_))
    pick = np.unique(np.concatenate([np.arange(0, n, 997), np.arange(0, 200), np.arange(n - 200, n), np.arange(639, n, 640)[:50],
                                     np.arange(640, n, 640)[:50]]))
    for times, off in grids:
        e0, p0, v0 = cat.propagate(times[pick], off, threads=8)
        for layout in (native.SAT_MAJOR, native.TIME_MAJOR):
            shape = (dev.n, n, 3) if layout == native.SAT_MAJOR else (n, dev.n, 3)
            pos, vel = np.full(shape, np.nan), np.full(shape, np.nan)
            err = np.zeros((dev.n, n), dtype=np.uint8)
            dev.propagate_host(times, off, pos=pos, vel=vel, err=err, layout=layout)
            assert dev.last_path() & (native.PATH_ROWS_FAST | native.PATH_TILES_FAST | native.PATH_COLS_FAST), dev.last_path()
            assert not np.isnan(pos).any() and not np.isnan(vel).any(), (layout, "an output element was left unwritten")
            if layout == native.TIME_MAJOR:
                pos, vel = pos.transpose(1, 0, 2), vel.transpose(1, 0, 2)
            assert np.array_equal(err[:, pick], e0)
            assert np.abs(pos[:, pick] - p0).max() < TOL_R and np.abs(vel[:, pick] - v0).max() < TOL_V, (
                layout, np.abs(pos[:, pick] - p0).max(), np.abs(vel[:, pick] - v0).max())


def test_one_satellite_long_series_fast_and_handed_over_segments(native, orc, synth):
    """azh_propagate_one_host / _device on long series (k_one_fast: every wave fits a grid through its own 1,024 points and runs
    the branch-free step where they are (quasi-)uniform and the window passes the bounds; k_one_satellite works off the rest):
    uniform, (jd, fr)-rounded, backwards, irregular, half-and-half, ragged tails, windows too long for the fast step, an
    eccentric and a deep-space member, positions only -- every point against the oracle at the times given."""
    import torch
    pairs = synth.synth_catalog(n_near=600, n_deep=8, seed=41)
    dev_all = native.DeviceConstellation.from_tle_lines(pairs, 1, 0)
    ecc = dev_all.field("ecco")
    _, deep, _ = dev_all.status
    near = np.flatnonzero(~deep)
    picks = {"leo": int(near[0]), "ecc": int(near[np.argmax(ecc[near])]), "deep": int(np.flatnonzero(deep)[0])}
    assert ecc[picks["ecc"]] > 0.05
    rng = np.random.default_rng(5)
    n = (1 << 20) + 77 + 64              # (just over the fast launch's threshold: 1,024 whole segments of 1,024 and a tail of 141)
    day = synth.START_JD
    jd, fr = np.full(n, day), 0.2 + np.arange(n) / 864000.0
    grids = {
        "uniform": 3.0 + np.arange(n) * 0.002,
        "jdfr": ((jd + fr) - (jd[0] + fr[0])) * 1440.0 + 100.0,
        "backwards": 2000.0 - np.arange(n) * 0.003,
        "irregular": np.sort(rng.uniform(0.0, 3000.0, n)),
        "half": np.concatenate([np.arange(n // 2) * 0.004, np.sort(rng.uniform(3000.0, 4000.0, n - n // 2))]),
        "long_windows": (np.arange(n) % 1024) * 5.0,    # every 1,024-point segment spans 5,120 minutes: the window cap says no
        "short_tail": 5.0 + np.arange((1 << 20) + 8) * 0.001,   # a tail of 8 points
    }
    for name, s in picks.items():
        dev = native.DeviceConstellation.from_tle_lines([pairs[s]], 1, 0)
        cat = orc.Catalog.from_pairs([pairs[s]], 1)
        for gname, ts in grids.items():
            m = len(ts)
            sel = np.unique(np.concatenate([np.arange(0, m, 211), np.arange(m - 300, m), np.arange(1023, m, 1024), np.arange(1024, m, 1024),
                                            np.arange(m // 2 - 70, m // 2 + 70)]))
            e0, p0, v0 = cat.propagate(ts[sel], None, layout=orc.SAT_MAJOR)
            e, r, v = dev.propagate_one(0, ts)
            segs, handed = dev.last_one_stats()
            want_segs = 0 if name == "deep" else (m + 1023) // 1024
            assert segs == want_segs, (name, gname, segs)
            if name != "deep":
                # what took the branch-free kernel: all of a (quasi-)uniform series but a short tail; none of an irregular one
                if gname in ("uniform", "jdfr", "backwards"):
                    assert handed == 0, (name, gname, handed)
                elif gname == "short_tail":
                    assert handed == 1, (name, gname, handed)        # the 8-point tail
                elif gname == "irregular":
                    assert handed == segs, (name, gname, handed)
                elif gname == "long_windows":
                    assert handed == segs - 1, (name, gname, handed)  # (the 141-point tail spans 705 minutes: inside the cap)
                else:
                    assert segs // 2 - 1 <= handed <= segs // 2 + 2, (name, gname, handed, segs)
            assert np.array_equal(e[sel], e0[0]), (name, gname)
            assert not e.any() or name == "deep" or gname == "long_windows", (name, gname)
            good = e0[0] == 0
            assert np.abs(r[sel][good] - p0[0][good]).max() < TOL_R and np.abs(v[sel][good] - v0[0][good]).max() < TOL_V, (
                name, gname, np.abs(r[sel][good] - p0[0][good]).max(), np.abs(v[sel][good] - v0[0][good]).max())
            # the device-pointer entry, positions only, every element written
            dts = torch.as_tensor(ts, device="cuda")
            dp = torch.full((m, 3), float("nan"), dtype=torch.float64, device="cuda")
            torch.cuda.synchronize()
            dev.propagate_one_device(0, dts.data_ptr(), m, dp.data_ptr(), None, None)
            dev.synchronize()
            hp = dp.cpu().numpy()
            assert not np.isnan(hp).any(), (name, gname)
            assert np.array_equal(hp, r), (name, gname, "positions-only and pos+vel calls disagree")
            if gname == "uniform":
                # the same member addressed inside the 608-satellite handle: bit for bit the 1-satellite handle's result
                e2, r2, v2 = dev_all.propagate_one(s, ts)
                assert dev_all.last_one_stats() == (segs, handed)
                assert np.array_equal(e2, e) and np.array_equal(r2, r) and np.array_equal(v2, v), (name, gname)


def test_repeated_inputs_skip_the_staging_and_nothing_else(native, orc, synth):
    """stage_inputs returns early on byte-identical (times, offsets, mask, mode, reference_jd) on the same stream: results of
    A, B, A sequences equal a fresh handle's bit for bit when only the times, only the offsets, only the mask, only the output
    mode or only the reference_jd change between calls -- and a changed input is never mistaken for the staged one."""
    pairs = synth.synth_catalog(n_near=90, n_deep=9, seed=47)
    n = 300
    day = synth.START_JD

    def run(dev, times, off, mask=None, mode=0, ref=0.0, layout=None):
        layout = native.SAT_MAJOR if layout is None else layout
        shape = (dev.n, n, 3) if layout == native.SAT_MAJOR else (n, dev.n, 3)
        pos, vel = np.full(shape, -7.0), np.full(shape, -7.0)
        err = np.zeros((dev.n, n), dtype=np.uint8)
        dev.propagate_host(times, off, pos=pos, vel=vel, err=err, layout=layout, mask=mask, mode=mode, reference_jd=ref)
        return pos, vel, err

    dev = native.DeviceConstellation.from_tle_lines(pairs, 1, 0)
    t1 = np.arange(n) * 1.0
    t2 = t1.copy(); t2[137] += 1e-9                       # one time moved by a nanominute
    o1 = (day - dev.epochs) * 1440.0
    o2 = o1.copy(); o2[5] += 1e-9
    m1 = np.ones(dev.n, dtype=np.uint8); m1[::7] = 0
    m2 = m1.copy(); m2[7] ^= 1
    cases = [dict(times=t1, off=o1), dict(times=t2, off=o1), dict(times=t1, off=o1), dict(times=t1, off=o2), dict(times=t1, off=o1),
             dict(times=t1, off=o1, mask=m1), dict(times=t1, off=o1, mask=m2), dict(times=t1, off=o1, mask=m1), dict(times=t1, off=o1),
             dict(times=t1, off=o1, mode=1, ref=day), dict(times=t1, off=o1, mode=1, ref=day + 0.25), dict(times=t1, off=o1, mode=1, ref=day),
             dict(times=t1, off=o1, mode=2, ref=day), dict(times=t1, off=o1), dict(times=t1, off=o1, layout=native.TIME_MAJOR),
             dict(times=t1, off=o1, layout=native.TIME_MAJOR), dict(times=t1, off=None), dict(times=t1, off=o1)]
    for k, kw in enumerate(cases):
        got = run(dev, **kw)
        fresh = run(native.DeviceConstellation.from_tle_lines(pairs, 1, 0), **kw)
        for a, b in zip(got, fresh):
            assert np.array_equal(a, b), (k, sorted(kw))
