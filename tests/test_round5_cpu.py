"""Round 5, CPU tier: the bench record the driver parses and the host-side pieces added this round."""
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


@pytest.mark.parametrize("record", ["r04_bench_default.json", "r04_bench_force_sharded.json", "r03_bench_default.json"])
def test_compact_bench_line_from_canned_records(record):
    """bench_common.compact_line on real full records of earlier rounds (21 KB / 12 KB on one line -- what round 4 printed and
    the driver could not parse): the line is below 4 KB, round-trips through json, and carries every contract field."""
    import bench_common
    out = json.load(open(os.path.join(ROOT, "profiles", record)))
    line = bench_common.compact_line(out, "gpurun_out/bench_full.json")
    assert "\n" not in line and len(line.encode()) < 4096, len(line)
    j = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline"):
        assert k in j, k
    assert abs(j["value"] - out["value"]) <= 1e-5 * out["value"] and j["steps"] == out["steps"] and j["n_gpus"] == out["n_gpus"]
    assert j["config"]["workload"] and not any(k in j["config"] for k in ("model", "seq_len"))
    rf = j["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-4
    assert rf["kernel"] and rf["algorithmic_bytes_per_launch"] > 0 and "traffic" in rf
    if "cpu_baseline" in out and out["cpu_baseline"].get("value"):
        cb = j["cpu_baseline"]
        assert cb["value"] > 0 and cb["cores"] >= 1 and cb["kind"] in ("port", "reference") and cb["sample"]
    if out.get("secondary"):
        keys = [e["key"] for e in out["secondary"] if "key" in e]
        assert set(j["secondary_summary"]) == set(keys)
        for e in out["secondary"]:
            if "ms_per_step" in e and "failed" not in e:
                assert abs(j["secondary_summary"][e["key"]][0] - e["ms_per_step"]) <= 1e-3 * e["ms_per_step"]
    if out["config"].get("t_kernel_ms") is not None:
        assert j["config"]["t_kernel_ms"] > 0 and j["config"]["replicate_value"] > 0 and j["config"]["kernel_only_value"] > 0


def test_compact_bench_line_degrades_in_a_fixed_order():
    """An oversized record (hundreds of secondary entries, kilobytes of free text) still yields a valid short line: free text
    is cut, optional blocks are dropped, the contract fields stay."""
    import bench_common
    out = json.load(open(os.path.join(ROOT, "profiles", "r04_bench_default.json")))
    out["config"]["workload"] = "w" * 5000
    out["roofline"]["kernel"] = "k" * 5000
    out["cpu_baseline"]["sample"] = "s" * 5000
    out["secondary"] = [dict(out["secondary"][0], key="entry_%03d" % i) for i in range(400)]
    line = bench_common.compact_line(out, None)
    assert len(line.encode()) < 4096
    j = json.loads(line)
    assert j["value"] == pytest.approx(out["value"], rel=1e-5) and j["roofline"]["frac"] > 0 and j["cpu_baseline"]["value"] > 0
    assert "secondary_summary" not in j


def test_c_api_closed_forms_on_the_host(orc):
    """coords_* and orbital_* (the reference's c_api, src/c_api/root.zig L60-81) are pure host functions there and here:
    no device needed (this tier has none).  Against the oracle's restatement of WorldCoordinateSystem.zig and the closed
    forms of src/calculations.zig L83-125; argument checks of src/c_api/orbital_mechanics.zig."""
    import ctypes as C
    import numpy as np
    from astroz_amd import _native as native
    import astroz_amd as az
    for jd in (2451545.0, 2460500.5, 2460800.75, 2433282.5):
        assert abs(native.julian_to_gmst(jd) - orc.julian_to_gmst(jd)) < 1e-12
    L = orc.lib()
    rng = np.random.default_rng(8)
    for k in range(40):
        eci = rng.uniform(-9000.0, 9000.0, 3) * (1.0 if k % 4 else 6.0)   # (LEO to beyond GEO)
        gm = rng.uniform(0, 2 * np.pi)
        want = np.empty(3)
        L.orc_eci_to_ecef(eci.ctypes.data_as(C.c_void_p), C.c_double(np.sin(gm)), C.c_double(np.cos(gm)), want.ctypes.data_as(C.c_void_p))
        got = native.eci_to_ecef(eci, gm)
        assert np.abs(got - want).max() < 1e-11
        lla = np.empty(3)
        L.orc_ecef_to_geodetic(want.ctypes.data_as(C.c_void_p), lla.ctypes.data_as(C.c_void_p))
        got = native.ecef_to_geodetic(want)
        assert abs(got[0] - np.degrees(lla[0])) < 1e-11 and abs(got[1] - np.degrees(lla[1])) < 1e-11 and abs(got[2] - lla[2]) < 1e-9
    lib = native.lib()
    mu, r1, r2 = 398600.4418, 6778.0, 42164.0

    class H(C.Structure):
        _fields_ = [(n, C.c_double) for n in ("sma", "dv1", "dv2", "dvt", "t", "t_days")]
    h = H()
    assert lib.orbital_hohmann(mu, r1, r2, C.byref(h)) == 0
    sma = 0.5 * (r1 + r2)
    v1, v2 = np.sqrt(mu / r1), np.sqrt(mu / r2)
    dv1, dv2 = v1 * np.sqrt(2 * r2 / (r1 + r2)) - v1, v2 - v2 * np.sqrt(2 * r1 / (r1 + r2))
    for got, want in ((h.sma, sma), (h.dv1, dv1), (h.dv2, dv2), (h.dvt, abs(dv1) + abs(dv2)), (h.t, np.pi * np.sqrt(sma ** 3 / mu)),
                      (h.t_days, np.pi * np.sqrt(sma ** 3 / mu) / 86400.0)):
        assert abs(got - want) <= 1e-14 * abs(want)
    assert lib.orbital_hohmann(mu, -1.0, r2, C.byref(h)) == -20 and lib.orbital_hohmann(mu, r1, r1 + 10.0, C.byref(h)) == -20
    assert lib.orbital_hohmann(mu, r1, r2, None) == -101
    assert abs(lib.orbital_velocity(mu, r1, 0.0) - v1) < 1e-13 and abs(lib.orbital_velocity(mu, r1, sma) - np.sqrt(mu * (2 / r1 - 1 / sma))) < 1e-13
    assert abs(lib.orbital_period(mu, r2) - 2 * np.pi * np.sqrt(r2 ** 3 / mu)) < 1e-9
    assert lib.orbital_velocity(mu, -1.0, 0.0) == -1.0 and lib.orbital_period(mu, 0.0) == -1.0 and lib.orbital_escape_velocity(mu, 0.0) == -1.0
    d = az.hohmann_transfer(az.EARTH_MU, r1, r2)
    assert abs(d["sma"] - sma) < 1e-9 and abs(az.escape_velocity(az.EARTH_MU, r1) - np.sqrt(2 * az.EARTH_MU / r1)) < 1e-12
    # the device evaluation is a known-answer test and fails loudly without a device
    out = (C.c_double * 5)()
    if native.device_count() == 0:
        assert lib.azh_selftest_coords(0, (C.c_double * 4)(2451545.0, 0, 0, 0), out) == native.AZ_ERR_HIP


def test_bench_workload_descriptions():
    """bench_common.describe_workload: the `config.workload` / `roofline.kernel` strings of the bench record name the BASELINE
    config, the arithmetic and the kernel family for every flag combination the driver or the profile set uses."""
    import argparse
    import bench_common

    def ns(**kw):
        d = dict(sats=13478, deep=0, config5_share=False, f32_out=False, f32_fp64=False, f32_arith=False, no_fast_path=False,
                 no_tile_kernel=False, mode="teme", layout="sat")
        d.update(kw)
        return argparse.Namespace(**d)
    wl, par, arith, kn = bench_common.describe_workload(ns(), 1, False, False, 0, 13478, 1440, True, 0, True)
    assert wl.startswith("config 2: 13478-sat") and "fp64 TEME pos+vel" in wl and par == "single GPU" and arith == "fp64 arithmetic"
    assert kn.startswith("k_rows_fast<pos+vel>")
    wl, par, arith, kn = bench_common.describe_workload(ns(deep=1522, layout="time"), 1, False, False, 0, 15000, 1440, False, 0, True)
    assert wl.startswith("config 3: 15000-sat") and "time-major" in wl and kn.startswith("k_tiles_fast<pos+vel>")
    wl, par, arith, kn = bench_common.describe_workload(ns(), 8, True, True, 1, 13478, 1440, True, 0, True)
    assert wl.startswith("config 4:") and "RCCL all-gather" in wl and "1-chunk" in wl and par == "satellite-sharded x8 + RCCL all-gather"
    wl, par, arith, kn = bench_common.describe_workload(ns(sats=125000, config5_share=True, f32_out=True), 1, False, False, 0, 125000, 10000, True, 0, True)
    assert wl.startswith("config 5, ONE GPU's share") and "mixed-precision" in arith and "k_rows_fast32<MIXED," in kn
    wl, par, arith, kn = bench_common.describe_workload(ns(layout="time", no_tile_kernel=True), 1, False, False, 0, 13478, 1440, False, 0, False)
    assert kn.startswith("k_propagate<time-major,pos>") and "pos only" in wl
