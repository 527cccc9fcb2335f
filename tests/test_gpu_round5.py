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
