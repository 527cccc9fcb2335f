#!/usr/bin/env python3
"""Round 6: the host route of few-point calls (astroz_amd/csrc/host_step.h) against the kernel route, by series length --
where the two cross (the default of azh_set_host_points), and what the scalar Python call costs through each binding."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g  # noqa: E402

g.build()
from astroz_amd import _native  # noqa: E402
from astroz_amd.api import Satrec, WGS72  # noqa: E402


def wall(fn, k):
    for _ in range(max(3, k // 10)):
        fn()
    t0 = time.perf_counter()
    for _ in range(k):
        fn()
    return (time.perf_counter() - t0) / k * 1e6


for name, l1, l2 in (("ISS (near-earth)", "1 25544U 98067A   24127.82853009  .00015698  00000+0  27310-3 0  9995",
                      "2 25544  51.6393 160.4574 0003580 140.6673 205.7250 15.50957674452123"),
                     ("GPS (deep space)", "1 20413U 83020D   05363.79166667  .00000000  00000-0  00000+0 0  7041",
                      "2 20413  12.3514 187.4253 7864447 196.3027 356.5478  0.24690082 78320")):
    sat = Satrec.twoline2rv(l1, l2, WGS72)
    dev = sat._ensure()
    jd, fr = sat.jdsatepoch, sat.jdsatepochF + 0.3
    n0 = _native.get_host_points()
    print("== %s ==" % name)
    print("Satrec.sgp4, host route, %s: %.3f us" % ("CPython shim" if _native.fast_scalar() else "ctypes", wall(lambda: sat.sgp4(jd, fr), 50000)))
    sc = sat._scalar
    sat._scalar = False
    print("Satrec.sgp4, host route, ctypes + numpy wrapper: %.3f us" % wall(lambda: sat.sgp4(jd, fr), 5000))
    _native.set_host_points(0)
    print("Satrec.sgp4, kernel route (round 5): %.3f us" % wall(lambda: sat.sgp4(jd, fr), 500))
    sat._scalar = sc
    for npts in (1, 4, 16, 32, 64, 128, 256, 512, 1024, 4096):
        tt = np.linspace(0.0, 1440.0, npts)
        _native.set_host_points(1 << 20)
        h = wall(lambda: dev.propagate_one(sat._idx, tt), 2000 if npts <= 64 else 200)
        _native.set_host_points(0)
        k = wall(lambda: dev.propagate_one(sat._idx, tt), 300)
        print("propagate_one x %5d points: host route %8.2f us (%.3f us/point)   kernel route %8.2f us" % (npts, h, h / npts, k))
    _native.set_host_points(n0)
