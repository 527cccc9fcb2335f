"""Round 6, CPU tier: the host route of few-point calls (the library's own per-point step compiled for the host,
astroz_amd/csrc/host_step.h) held to the oracle through the SHIPPED object, and its Python plumbing."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from test_host_cpu import _grav6  # noqa: E402


def _table(emul, tles, g):
    """element table el[field * n_pad + sat] from the host-compiled init (the device runs the same source in k_init)"""
    nf = emul.emul_num_fields()
    n_pad = (len(tles) + 63) // 64 * 64
    el = np.zeros((nf, n_pad))
    flags = np.zeros(len(tles), dtype=np.uint32)
    for i, t in enumerate(tles):
        raw = np.array([t.epoch_jd, t.mm_revday, t.ecc, t.incl_deg, t.raan_deg, t.argp_deg, t.ma_deg, t.bstar])
        f = np.zeros(nf)
        flags[i] = emul.emul_init(raw.ctypes.data, g.ctypes.data, f.ctypes.data)
        el[:, i] = f
    return el, n_pad, flags


def test_host_step_in_the_library_matches_oracle(native, emul, orc):
    """azh_selftest_host_step = the code path run_one_satellite takes for calls of <= azh_get_host_points() points, on a
    mixed catalog (near-earth incl. simplified-drag and eccentric members, every resonance class), +-2 weeks: the fp64 gate."""
    from astroz_amd import synth
    L = native.lib()
    L.azh_selftest_host_step.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_uint32, C.c_int32, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
    pairs = synth.synth_catalog(120, 40, seed=9)
    tles = [orc.parse_lines(a, b) for a, b in pairs]
    cat = orc.Catalog(tles, 1)
    el, n_pad, flags = _table(emul, tles, _grav6(1))
    rng = np.random.default_rng(3)
    worst_r = worst_v = 0.0
    for i in range(len(tles)):
        ts = np.ascontiguousarray(rng.uniform(-20160.0, 20160.0, 24))
        out = np.zeros((len(ts), 6))
        err = np.zeros(len(ts), dtype=np.uint8)
        rc = L.azh_selftest_host_step(el.ctypes.data, n_pad, i, int(flags[i]), 1, ts.ctypes.data, len(ts), out.ctypes.data, err.ctypes.data)
        assert rc == 0
        for k in range(len(ts)):
            orc_rc, r, v = cat.propagate_one(i, ts[k])
            assert orc_rc == err[k]
            if orc_rc:
                assert not out[k].any()
                continue
            worst_r = max(worst_r, np.abs(out[k, :3] - r).max())
            worst_v = max(worst_v, np.abs(out[k, 3:] - v).max())
    assert worst_r < 1e-6 and worst_v < 1e-9, (worst_r, worst_v)
    assert worst_r < 2e-7 and worst_v < 2e-10, (worst_r, worst_v)   # (what the kernels hold on such spans: DESIGN.md 2)


def test_host_step_golden_vectors(native, emul, orc, golden):
    """the reference's own published vectors (Vallado's 00005 / 06251, src/Sgp4Batch.zig L240-296) straight through the host
    route, to print precision"""
    L = native.lib()
    L.azh_selftest_host_step.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_uint32, C.c_int32, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
    g1 = golden["G1_vallado_near_earth"]
    assert g1["grav"] == "wgs72"
    for case in g1["cases"]:
        t = orc.parse_lines(case["line1"], case["line2"])
        el, n_pad, flags = _table(emul, [t], _grav6(1))
        ts = np.ascontiguousarray([st["t"] for st in case["states"]], dtype=np.float64)
        out = np.zeros((len(ts), 6))
        assert L.azh_selftest_host_step(el.ctypes.data, n_pad, 0, int(flags[0]), 1, ts.ctypes.data, len(ts), out.ctypes.data, None) == 0
        for st, o in zip(case["states"], out):
            np.testing.assert_allclose(o[:3], st["r"], atol=2e-8, rtol=0)
            np.testing.assert_allclose(o[3:], st["v"], atol=2e-9, rtol=0)


def test_host_points_switch_and_scalar_shim(native):
    """the point limit is a process-wide switch; the CPython shim of the scalar call binds to the loaded library"""
    n0 = native.get_host_points()
    assert n0 == 128 or "ASTROZ_AMD_HOST_POINTS" in os.environ
    native.set_host_points(0)
    assert native.get_host_points() == 0
    native.set_host_points(n0)
    assert native.get_host_points() == n0
    mod = native.fast_scalar()
    assert mod is not None and callable(mod.sgp4)      # built by __graft_entry__.build() wherever Python.h exists
    with pytest.raises(TypeError):
        mod.sgp4(1, 2.0)


def test_host_route_is_not_a_fallback(native):
    """no device -> no handle -> nothing for the host route to run on: creation fails with AZ_ERR_HIP exactly as before, and
    the product never imports the oracle (the host route is the library's own step source)"""
    if native.device_count() > 0:
        pytest.skip("a GPU is visible")
    l1 = "1 25544U 98067A   24127.82853009  .00015698  00000+0  27310-3 0  9995"
    l2 = "2 25544  51.6393 160.4574 0003580 140.6673 205.7250 15.50957674452123"
    from astroz_amd.api import Satrec
    sat = Satrec.twoline2rv(l1, l2)
    with pytest.raises(native.NativeError) as ei:
        sat.sgp4(2460437.5, 0.0)
    assert ei.value.code == native.AZ_ERR_HIP


def test_product_never_references_the_checker():
    """the host route is the library's own step source: nothing under astroz_amd/, include/ or bindings/ names the oracle"""
    import subprocess
    hits = subprocess.run(["grep", "-rlE", r"^\s*(from|import)\s+oracle|oracle/|liboracle|astroz_oracle", os.path.join(ROOT, "astroz_amd"),
                           os.path.join(ROOT, "include"), os.path.join(ROOT, "bindings")], capture_output=True, text=True).stdout.split()
    assert [h for h in hits if not h.endswith((".so", ".o", ".pyc"))] == []


def test_compact_line_carries_n_gt_1_certificates():
    """an N > 1 line is self-certifying -- per-rank sampled-row parity + checksum, a CPU baseline, the
    sharded screen -- and still fits the driver's 4-KB window.  Built from canned records (round 5's --force-sharded line and
    default line with its secondary block) with eight ranks' worth of certificates injected."""
    import json
    import bench_common
    for record, world in (("r05_bench_force_sharded.json", 8), ("r05_bench_default_full.json", 8)):
        path = os.path.join(ROOT, "profiles", record)
        out = json.load(open(path))
        assert "metric" in out
        out["n_gpus"] = world
        out["parity"] = {"per_rank": [{"rank": r, "n_sats": 1685, "sample_sats": 24, "max_dr": 1.1e-8 + r * 1e-10, "max_dv": 9.0e-12,
                                        "checksum": 123456789.123456 + r, "absmax": 42164.9, "finite": True} for r in range(world)],
                         "per_rank_note": "x" * 300, "max_abs_dr_km": 1.2e-8, "max_abs_dv_kms": 8.9e-12}
        out["cpu_baseline"] = {"value": 6.3e8, "unit": "propagations/s", "cores": 16, "kind": "port", "sample": "SUBSET of the rank-0 workload " + "y" * 400,
                               "threads_1": {"value": 4.9e7}, "cpu_model": "AMD EPYC 9575F 64-Core Processor"}
        out.setdefault("config", {})["sharded_screen"] = {"ms": 0.031, "value": 6.2e11, "index_mismatches": 0, "max_dd_km": 6.8e-9,
                                                          "per_rank": [{"rank": r, "rows": 1685} for r in range(world)], "what": "z" * 200}
        line = bench_common.compact_line(out, "gpurun_out/bench_full.json")
        assert len(line.encode()) < 4096, len(line)
        j = json.loads(line)
        pr = j["parity"]["per_rank"]
        assert len(pr) == world and pr[3][0] == 3 and pr[3][4] is True and abs(pr[3][1] - 1.13e-8) < 1e-10
        assert abs(pr[7][3] - (123456789.123456 + 7)) < 1e-3          # the checksum keeps 12 significant digits
        assert j["cpu_baseline"]["value"] == 6.3e8 and j["cpu_baseline"]["cores"] == 16 and j["cpu_baseline"]["kind"] == "port"
        assert j["config"]["sharded_screen"] == {"ms": 0.031, "value": 6.2e11, "index_mismatches": 0, "max_dd_km": 6.8e-9}


def test_rank_certificate_on_cpu_tensors(orc):
    """bench_common.rank_certificate (what every rank of an N > 1 run computes about its own output) on CPU tensors filled by the
    oracle: zero differences, the checksum is the plain sum, a poisoned element is caught by `finite`; both layouts, fp32 too."""
    import torch
    import bench_common
    from astroz_amd import synth
    pairs = synth.synth_catalog(60, 8, seed=3)
    cat = orc.Catalog.from_pairs(pairs, orc.WGS72)
    times = np.arange(0.0, 200.0, 2.0)
    off = (synth.START_JD - cat.epoch_jd) * 1440.0
    _, p, v = cat.propagate(times, off, layout=orc.SAT_MAJOR)
    for sat_major in (True, False):
        tp = torch.from_numpy(p if sat_major else np.ascontiguousarray(p.transpose(1, 0, 2)))
        tv = torch.from_numpy(v if sat_major else np.ascontiguousarray(v.transpose(1, 0, 2)))
        c = bench_common.rank_certificate(torch, pairs, times, off, tp, tv, sat_major, rank=5, threads=2)
        assert c["rank"] == 5 and c["n_sats"] == len(pairs) and c["sample_sats"] >= 20
        assert c["max_dr"] == 0.0 and c["max_dv"] == 0.0 and c["finite"] is True
        assert abs(c["checksum"] - (p.sum() + v.sum())) <= 1e-9 * abs(p).sum()
    c32 = bench_common.rank_certificate(torch, pairs, times, off, torch.from_numpy(p.astype(np.float32)), None, True, rank=0, threads=2)
    assert 0.0 < c32["max_dr"] < 4e-3 and "max_dv" not in c32
    bad = torch.from_numpy(p.copy())
    bad[7, 3, 1] = float("nan")
    assert bench_common.rank_certificate(torch, pairs, times, off, bad, None, True, rank=0, threads=2)["finite"] is False
