"""Round 5, GPU tier (all through the C ABI)."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu
TOL_R, TOL_V = 1e-6, 1e-9


@pytest.fixture(scope="module")
def native():
    import __graft_entry__ as g
    g.build()
    from astroz_amd import _native
    if _native.device_count() < 1:
        pytest.fail("no HIP device visible on a GPU-tier run")
    return _native


@pytest.fixture(scope="module")
def synth():
    from astroz_amd import synth as s
    return s


@pytest.mark.parametrize("threads", [3, 6, 9, 16])
def test_staged_copy_back_covers_every_byte(native, orc, synth, threads):
    """the staged host copy split a chunk into k pieces of floor(len / k) rounded up to 64 bytes; when
    the floor is a multiple of 64 and len % k != 0 the last len % k bytes stayed in the pinned slot.  100 satellites x 1,801
    times: the error matrix is 180,100 bytes (30,016 x 6 + 4).  Every output array is pre-filled with a sentinel and must come
    back fully written and equal to the oracle's."""
    pairs = synth.synth_catalog(n_near=100, n_deep=0, seed=31)
    dev = native.DeviceConstellation.from_tle_lines(pairs, native.WGS72, 0)
    cat = orc.Catalog.from_pairs(pairs, orc.WGS72)
    times = np.arange(0.0, 1801.0)
    off = (synth.START_JD - dev.epochs) * 1440.0
    e0, p0, v0 = cat.propagate(times, off, layout=orc.SAT_MAJOR)
    native.set_host_copy_threads(threads)
    try:
        pos = np.full((dev.n, 1801, 3), np.nan)
        vel = np.full((dev.n, 1801, 3), np.nan)
        err = np.full((dev.n, 1801), 0xAB, dtype=np.uint8)
        dev.propagate_host(times, off, pos=pos, vel=vel, err=err, layout=native.SAT_MAJOR)
    finally:
        native.set_host_copy_threads(-1)
    assert not np.isnan(pos).any() and not np.isnan(vel).any()
    assert np.array_equal(err, e0), int((err != e0).sum())
    assert np.abs(pos - p0).max() < TOL_R and np.abs(vel - v0).max() < TOL_V


@pytest.mark.parametrize("grid", ["uniform", "jdfr"])
def test_cols_kernel_opt_in(native, orc, synth, grid):
    """k_cols_fast, north_star's literal kernel (one lane = one satellite, time-major inner loop, per-satellite constants in
    LDS), selected with set_tile_kernel(2): a catalog that mixes near-circular, eccentric, deep-space and failed members in
    catalog order, ragged sizes (last wave partly filled), pos / pos+vel, TEME / ECEF / geodetic, an exact and a (jd, fr)
    grid, a row window, members weeks from epoch (windows the plan rejects -> redo pass) -- every row against the oracle."""
    import torch
    pairs = synth.synth_catalog(n_near=2500, n_deep=300, seed=91)
    rng = np.random.default_rng(3)
    order = rng.permutation(len(pairs))
    pairs = [pairs[i] for i in order]
    # a member whose initialisation fails (perigee below the surface): zeros, error code per row
    l1, l2 = pairs[17]
    bad2 = l2[:26] + "9990000" + l2[33:52] + "16.50000000" + l2[63:]
    pairs[17] = (l1, bad2)
    dev = native.DeviceConstellation.from_tle_lines(pairs, native.WGS72, 0)
    cat = orc.Catalog.from_pairs(pairs, orc.WGS72)
    assert dev.n == len(pairs) and dev.n % 64 != 0
    good = cat.init_rc == 0          # (the oracle propagates whatever it is given: a failed member's rows are NaN there, zeros + its code here)
    assert not good[17] and good.sum() == dev.n - 1
    n_t = 333
    if grid == "uniform":
        times = 5.0 + 2.0 * np.arange(n_t)
        off = (synth.START_JD - dev.epochs) * 1440.0
    else:
        jd = np.full(n_t, synth.START_JD)
        fr = 0.32853009 + np.arange(n_t) / 1440.0
        rjd = jd[0] + fr[0]
        times = ((jd + fr) - rjd) * 1440.0
        off = (rjd - dev.epochs) * 1440.0
    off[::41] += 25000.0          # weeks from epoch: rejected windows
    dev.set_tile_kernel(2)
    for mode, vel_on in ((native.OUT_TEME, True), (native.OUT_TEME, False), (native.OUT_ECEF, True), (native.OUT_GEODETIC, False)):
        e0, p0, v0 = cat.propagate(times, off, mode=mode, reference_jd=synth.START_JD, layout=orc.TIME_MAJOR)
        pos = torch.full((n_t, dev.n, 3), float("nan"), dtype=torch.float64, device="cuda")
        vel = torch.full((n_t, dev.n, 3), float("nan"), dtype=torch.float64, device="cuda") if vel_on else None
        err = torch.empty((dev.n, n_t), dtype=torch.uint8, device="cuda")
        dev.propagate_device(times, off, pos.data_ptr(), vel.data_ptr() if vel_on else None, mode=mode, reference_jd=synth.START_JD,
                             layout=native.TIME_MAJOR, d_err=err.data_ptr())
        dev.synchronize()
        path = dev.last_path()
        assert path & native.PATH_COLS_FAST and not path & (native.PATH_TILES_FAST | native.PATH_LANE_SAT), path
        assert bool(path & native.PATH_QUASI_UNIFORM) == (grid == "jdfr")
        eg = err.cpu().numpy()
        assert (eg[17] == eg[17, 0]).all() and eg[17, 0] != 0
        bad_rows = np.nonzero((eg != e0).any(axis=1) & good)[0]
        assert len(bad_rows) == 0, (bad_rows[:10], eg[bad_rows[:3], :4], e0[bad_rows[:3], :4], [order[b] for b in bad_rows[:10]])
        pg = pos.cpu().numpy()
        assert np.all(pg[:, 17] == 0.0)
        d = (pg - p0)[:, good]
        if mode == native.OUT_GEODETIC:
            d[..., 1] = (d[..., 1] + np.pi) % (2 * np.pi) - np.pi
            assert np.abs(d[..., :2]).max() < 1e-9 and np.abs(d[..., 2]).max() < TOL_R
        else:
            assert np.abs(d).max() < TOL_R, (mode, float(np.abs(d).max()))
        if vel_on:
            assert np.abs((vel.cpu().numpy() - v0)[:, good]).max() < TOL_V
    # a row window that cuts through waves: rows outside stay untouched
    lo, hi = 100, 1777
    pos = torch.full((n_t, dev.n, 3), -7.0, dtype=torch.float64, device="cuda")
    vel = torch.full((n_t, dev.n, 3), -7.0, dtype=torch.float64, device="cuda")
    dev.propagate_device(times, off, pos.data_ptr(), vel.data_ptr(), layout=native.TIME_MAJOR)
    dev.synchronize()
    pos.fill_(-7.0)
    vel.fill_(-7.0)
    dev.propagate_device_window(lo, hi, pos.data_ptr(), vel.data_ptr(), layout=native.TIME_MAJOR)
    dev.synchronize()
    e0, p0, v0 = cat.propagate(times, off, layout=orc.TIME_MAJOR)
    pn, vn = pos.cpu().numpy(), vel.cpu().numpy()
    assert np.all(pn[:, :lo] == -7.0) and np.all(pn[:, hi:] == -7.0) and np.all(vn[:, :lo] == -7.0) and np.all(vn[:, hi:] == -7.0)
    assert np.abs(pn[:, lo:hi] - p0[:, lo:hi]).max() < TOL_R and np.abs(vn[:, lo:hi] - v0[:, lo:hi]).max() < TOL_V   # (row 17 lies outside)
    # ... and the same bytes as the tile kernel's to rounding
    dev.set_tile_kernel(1)
    pos2 = torch.empty_like(pos)
    dev.propagate_device(times, off, pos2.data_ptr(), None, layout=native.TIME_MAJOR)
    dev.synchronize()
    assert dev.last_path() & native.PATH_TILES_FAST
    assert np.abs((pos2.cpu().numpy() - p0)[:, good]).max() < TOL_R
