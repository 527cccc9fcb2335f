"""The non-headline configurations of bench.py (`secondary`): every other layout / grid / frame / precision of the hot path,
the API-shaped calls, the screens and the ingest -- run AFTER the headline's timed region.  bench.py writes the full block to a
file and to an earlier stdout line; only `key -> [ms_per_step, frac]` pairs enter the final line (bench_common.compact_line)."""
import ctypes
import os
import time

import numpy as np

from bench_common import (BYTES_OUT_P, BYTES_OUT_PV, ELEM_BYTES_PER_SAT, HBM_PEAK_GBS, _sample_rows, usable_cpus)


def run_secondary(torch, _native, synth, cuda, stream, dev2, pairs2, skip=()):
    """The non-headline configurations, each a few steps, run AFTER the headline's timed region on the same device
    and stream (the headline fields never depend on anything here).  One entry per workload: ms_per_step (HIP events
    on the launch stream around K back-to-back steps after W warm-ups), value, HBM roofline fraction of the step, and
    parity against the fp64 oracle on rows spread over the catalog (all times).  A failing entry reports the
    exception and never takes the bench line down."""
    from oracle import oracle
    sptr = stream.cuda_stream
    res = []

    def timed(fn, warm, steps, pre_ms=200.0):
        # the device idles while the CPU baseline / the oracle of the previous entry run: bring the clocks back to their
        # loaded level first (the same back-to-back preconditioning as the headline's, shorter)
        fn()
        torch.cuda.synchronize()
        t_pre = time.perf_counter()
        while (time.perf_counter() - t_pre) * 1e3 < pre_ms:
            for _ in range(4):
                fn()
            torch.cuda.synchronize()
        for _ in range(warm):
            fn()
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(steps):
            fn()
        e1.record(stream)
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / steps

    def case(key, workload, kernel, dev, pairs, n_times, *, layout, vel=True, f32=False, mode=0, steps=50, warm=20,
             cold=False, rows=16, ref_jd=0.0, arith32="mixed", stride_align=0, grid="uniform"):
        if key in skip:
            return
        ent = {"key": key, "workload": workload, "kernel": kernel}
        try:
            n = dev.n
            dev.set_f32_arithmetic(arith32)
            times = np.arange(n_times, dtype=np.float64)
            offs = (synth.START_JD - dev.epochs) * 1440.0
            if grid == "jdfr":
                # the reference's own call: SatrecArray.sgp4(jd, fr) with jd = full(n, day), fr = f0 + arange(n)/1440
                # (examples/python_sgp4.py L31-33) -> times = ((jd + fr) - reference_jd) * 1440 (api.py L300-302): uniform only to
                # ~4e-7 min after the rounding of jd + fr at 2.46e6 days
                jd = np.full(n_times, synth.START_JD)
                fr = 0.32853009 + np.arange(n_times) / 1440.0
                rjd = jd[0] + fr[0]
                times = ((jd + fr) - rjd) * 1440.0
                offs = (rjd - dev.epochs) * 1440.0
                st_ = (times[-1] - times[0]) / (n_times - 1)
                ent["grid"] = {"kind": "jd+fr (api.py L300-302)", "max_dev_from_uniform_min": float(np.abs(times - (times[0] + np.arange(n_times) * st_)).max())}
            elif grid == "jitter":
                # one-minute grid with +-20 s of jitter per point: "uniform with jitter" (fast_step.h, the wide DELTA form)
                times = times + np.random.default_rng(7).uniform(-1.0 / 3.0, 1.0 / 3.0, n_times)
                ent["grid"] = {"kind": "one-minute steps + uniform(-20 s, 20 s) jitter"}
            elif grid == "random":
                # sorted random times over the same day: no uniform structure at all, the generic kernels
                times = np.sort(np.random.default_rng(7).uniform(0.0, float(n_times), n_times))
                ent["grid"] = {"kind": "sorted uniform-random times over the span"}
            stride = (n + stride_align - 1) // stride_align * stride_align if (stride_align and layout == _native.TIME_MAJOR) else n
            shape = (n_times, stride, 3) if layout == _native.TIME_MAJOR else (n, n_times, 3)
            odt = torch.float32 if f32 else torch.float64
            pos = torch.empty(shape, dtype=odt, device=cuda)
            v = torch.empty(shape, dtype=odt, device=cuda) if vel else None
            pp, vp = pos.data_ptr(), (v.data_ptr() if vel else None)
            dev.propagate_device(times, offs, pp, vp, mode=mode, reference_jd=ref_jd, layout=layout, stride=(stride if layout == _native.TIME_MAJOR else 0), stream=sptr, f32=f32)
            torch.cuda.synchronize()
            ent["path"] = dev.last_path()  # azh_last_path: 1 k_rows_fast, 2 k_tiles_fast, 4 k_rows, 8 k_propagate, 16 k_rows_deep, 32 quasi-uniform form
            ms = timed(lambda: dev.propagate_device_cached(pp, vp, layout=layout, stride=(stride if layout == _native.TIME_MAJOR else 0), stream=sptr, f32=f32), warm, steps)
            props = n * n_times
            nbytes = props * (BYTES_OUT_PV if vel else BYTES_OUT_P) * (0.5 if f32 else 1.0) + n_times * 8 + n * ELEM_BYTES_PER_SAT
            ent.update({"ms_per_step": ms, "value": props / (ms / 1e3), "unit": "propagations/s", "steps": steps, "warmup": warm,
                        "n_sats": n, "n_times": n_times, "dtype_out": "f32" if f32 else "f64",
                        "roofline": {"bound": "hbm", "achieved": nbytes / (ms / 1e3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                     "frac": nbytes / (ms / 1e3) / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes_per_launch": nbytes}})
            if cold:
                # a NEW time grid every call: input staging (H2D of times/offsets), k_prep_inc, k_deep_seed and the step
                # itself, host wall clock around call + synchronize
                cs = []
                for j in range(5):
                    tj = times + 0.25 * (j + 1)
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    dev.propagate_device(tj, offs, pp, vp, mode=mode, reference_jd=ref_jd, layout=layout, stride=(stride if layout == _native.TIME_MAJOR else 0), stream=sptr, f32=f32)
                    torch.cuda.synchronize()
                    cs.append((time.perf_counter() - t0) * 1e3)
                ent["cold_grid_call_ms"] = {"median": sorted(cs)[len(cs) // 2], "min": min(cs),
                                            "what": "first call on a new time grid: H2D staging + k_prep_inc + k_deep_seed + the step, "
                                                    "host wall clock incl. the final synchronize"}
                dev.propagate_device(times, offs, pp, vp, mode=mode, reference_jd=ref_jd, layout=layout, stride=(stride if layout == _native.TIME_MAJOR else 0), stream=sptr, f32=f32)
                torch.cuda.synchronize()
            # parity on sampled rows, all times
            rws = _sample_rows(n, rows)
            cat = oracle.Catalog.from_pairs([pairs[i] for i in rws], oracle.WGS72)
            _, p0, v0 = cat.propagate(times, offs[rws], mode=mode, reference_jd=ref_jd, layout=oracle.SAT_MAJOR,
                                      threads=usable_cpus())
            idx = torch.as_tensor(rws, device=cuda)
            take = (lambda x: x[:, idx].permute(1, 0, 2)) if layout == _native.TIME_MAJOR else (lambda x: x[idx])
            gp = take(pos).cpu().numpy().astype(np.float64)
            if mode == 2:
                # geodetic rows are (lat rad, lon rad, alt km): longitude differences modulo 2 pi
                d = gp - p0
                d[..., 1] = (d[..., 1] + np.pi) % (2 * np.pi) - np.pi
                ent["parity"] = {"rows": int(len(rws)), "max_abs_dlatlon_rad": float(np.abs(d[..., :2]).max()),
                                 "max_abs_dalt_km": float(np.abs(d[..., 2]).max())}
            else:
                ent["parity"] = {"rows": int(len(rws)), "max_abs_dr_km": float(np.abs(gp - p0).max())}
            if vel:
                ent["parity"]["max_abs_dv_kms"] = float(np.abs(take(v).cpu().numpy().astype(np.float64) - v0).max())
            del pos, v
        except Exception as exc:
            ent["failed"] = repr(exc)
        res.append(ent)

    TM, SM = _native.TIME_MAJOR, _native.SAT_MAJOR
    ref_jd = synth.START_JD
    n2 = dev2.n
    case("config2_pos_only", "config 2, positions only (%d x 1,440, fp64 TEME, satellite-major)" % n2,
         "k_rows_fast<pos> + redo", dev2, pairs2, 1440, layout=SM, vel=False)
    case("config2_time_major", "config 2, TIME-major output (the reference benchmark's physical layout, api.py L304-314), fp64 TEME pos+vel",
         "k_tiles_fast<pos+vel> + redo", dev2, pairs2, 1440, layout=TM)
    case("config2_time_major_aligned", "config 2, TIME-major output with the time rows padded to a multiple of 16 satellites (out_stride_sats = "
         "13,488: every 384-byte tile run on whole cache lines; what SatrecArray.sgp4_device(padded=True) allocates), fp64 TEME pos+vel",
         "k_tiles_fast<pos+vel> (streaming flush) + redo", dev2, pairs2, 1440, layout=TM, stride_align=16)
    case("config2_time_major_jdfr", "config 2, TIME-major, on the grid the reference's own API call produces: SatrecArray.sgp4(jd, fr), "
         "times = ((jd + fr) - reference_jd) * 1440 (quasi-uniform: first-order correction of every point to its rounded time)",
         "k_tiles_fast<pos+vel,DELTA> + redo", dev2, pairs2, 1440, layout=TM, grid="jdfr")
    case("config2_sat_major_jdfr", "config 2, satellite-major, on the (jd, fr) grid of the reference's API call",
         "k_rows_fast<pos+vel,DELTA> + redo", dev2, pairs2, 1440, layout=SM, grid="jdfr")
    case("config2_time_major_irregular", "config 2, TIME-major, one-minute steps with +-20 s jitter (VERDICT r03's irregular grid: uniform with jitter, "
         "the wide quasi-uniform form)", "k_tiles_fast<pos+vel,DELTA=2> + redo", dev2, pairs2, 1440, layout=TM, grid="jitter", steps=20, warm=5)
    case("config2_sat_major_irregular", "config 2, satellite-major, one-minute steps with +-20 s jitter",
         "k_rows_fast<pos+vel,DELTA=2> + redo", dev2, pairs2, 1440, layout=SM, grid="jitter", steps=20, warm=5)
    case("config2_time_major_random", "config 2, TIME-major, sorted random times (no uniform structure: the generic kernels)",
         "k_propagate<time-major> (lane = satellite, generic step)", dev2, pairs2, 1440, layout=TM, grid="random", steps=20, warm=5)
    case("config2_sat_major_random", "config 2, satellite-major, sorted random times",
         "k_rows (generic, lane = time)", dev2, pairs2, 1440, layout=SM, grid="random", steps=20, warm=5)
    case("config2_ecef_time_major", "config 2, ECEF time-major (the default of the reference's high-level propagate(), "
         "Constellation.zig L489-506), fp64 pos+vel", "k_tiles_fast<pos+vel,ECEF> + redo", dev2, pairs2, 1440, layout=TM, mode=1, ref_jd=ref_jd)
    case("config2_ecef_sat_major", "config 2, ECEF satellite-major, fp64 pos+vel", "k_rows_fast<pos+vel,FRAME> + redo",
         dev2, pairs2, 1440, layout=SM, mode=1, ref_jd=ref_jd)
    case("config2_geodetic_time_major", "config 2, geodetic (lat, lon [rad], alt km) time-major, positions only",
         "k_propagate<time-major,pos,FRAME> (lane = satellite)", dev2, pairs2, 1440, layout=TM, vel=False, mode=2, ref_jd=ref_jd, steps=10)
    if "config1" not in skip:
        ent = {"key": "config1", "kernel": "k_one_satellite (the kernel reads the times from and writes into a pinned buffer itself) / "
                                            "the constellation kernels on a 1-satellite catalog",
               "workload": "BASELINE config 1: the ISS TLE x 1,440 one-minute steps through the Python API mirror (examples/python_sgp4.py "
                           "L31-33): Satrec.sgp4_array(jd, fr), SatrecArray([sat]).sgp4(jd, fr) and 1,440 scalar Satrec.sgp4 calls; host "
                           "wall clock per call, host arrays in and out (reference: 30.8 M/s single-thread sgp4_array, README.md L25-33)"}
        try:
            from astroz_amd.api import Satrec, SatrecArray, WGS72
            l1 = "1 25544U 98067A   24127.82853009  .00015698  00000+0  27310-3 0  9995"
            l2 = "2 25544  51.6393 160.4574 0003580 140.6673 205.7250 15.50957674452123"
            sat = Satrec.twoline2rv(l1, l2, WGS72)
            jd = np.full(1440, sat.jdsatepoch)
            fr = sat.jdsatepochF + np.arange(1440) / 1440.0

            def wall(fn, k):
                for _ in range(max(3, k // 10)):
                    fn()
                t0 = time.perf_counter()
                for _ in range(k):
                    fn()
                return (time.perf_counter() - t0) / k * 1e6
            us_arr = wall(lambda: sat.sgp4_array(jd, fr), 300)
            sa = SatrecArray([sat], device=cuda.index or 0)
            us_sa = wall(lambda: sa.sgp4(jd, fr), 200)
            us_one = wall(lambda: sat.sgp4(jd[0], fr[7]), 20000)
            scalar_route = "host step" if sat._ensure().last_path() == _native.PATH_HOST_STEP else "kernel"
            # the same scalar call with the host route switched off (one launch + one synchronize per point: round 5's figure),
            # and the per-point cost of the host route as a function of the series length (where the two routes cross)
            n_host = _native.get_host_points()
            _native.set_host_points(0)
            us_one_dev = wall(lambda: sat.sgp4(jd[0], fr[7]), 500)
            _native.set_host_points(1 << 20)
            sweep = {}
            for npts in (1, 8, 64, 256, 1024):
                tt = np.linspace(0.0, 1440.0, npts)
                sweep[str(npts)] = {"host_route_us": wall(lambda: sat._ensure().propagate_one(sat._idx, tt), 2000 if npts <= 64 else 200)}
            _native.set_host_points(0)
            for npts in (1, 8, 64, 256, 1024):
                tt = np.linspace(0.0, 1440.0, npts)
                sweep[str(npts)]["kernel_route_us"] = wall(lambda: sat._ensure().propagate_one(sat._idx, tt), 300)
            _native.set_host_points(n_host)
            # the same calls on a grid that CHANGES every call (stage_inputs skips the staging of byte-identical
            # inputs, so the loops above time the repeated-grid case): fr moved by a few microseconds per call
            box = {"k": 0}

            def fresh():
                box["k"] += 1
                return fr + box["k"] * 1e-10
            us_sa_fresh = wall(lambda: sa.sgp4(jd, fresh()), 200)
            us_arr_fresh = wall(lambda: sat.sgp4_array(jd, fresh()), 300)
            e_, r_, v_ = sat.sgp4_array(jd, fr)
            e2, r2, v2 = sa.sgp4(jd, fr)
            cat = oracle.Catalog.from_pairs([(l1, l2)], oracle.WGS72)
            ts = ((jd + fr) - (sat.jdsatepoch + sat.jdsatepochF)) * 1440.0
            _, p0, v0 = cat.propagate(ts, None, layout=oracle.SAT_MAJOR)
            ent.update({"ms_per_step": us_arr / 1e3, "value": 1440 / (us_arr / 1e6), "unit": "propagations/s (Satrec.sgp4_array, host arrays)",
                        "sgp4_array_us": us_arr, "satrec_array_sgp4_us": us_sa, "scalar_sgp4_us": us_one,
                        "scalar_sgp4_kernel_route_us": us_one_dev, "scalar_route": scalar_route,
                        "scalar_binding": "CPython shim" if _native.fast_scalar() else "ctypes", "host_points": n_host,
                        "one_satellite_series_us": sweep,
                        "sgp4_array_fresh_grid_us": us_arr_fresh, "satrec_array_sgp4_fresh_grid_us": us_sa_fresh,
                        "note": "sgp4_array_us / satrec_array_sgp4_us: the SAME (jd, fr) every call (the staged grid is reused); "
                                "*_fresh_grid_us: a different grid every call (times and offsets re-staged, increments / record / plan rebuilt)",
                        "parity": {"max_abs_dr_km": float(max(np.abs(r_ - p0[0]).max(), np.abs(r2[0] - p0[0]).max())),
                                   "max_abs_dv_kms": float(max(np.abs(v_ - v0[0]).max(), np.abs(v2[0] - v0[0]).max())),
                                   "err_nonzero": int(np.count_nonzero(e_) + np.count_nonzero(e2))}})
            del sa
        except Exception as exc:
            ent["failed"] = repr(exc)
        res.append(ent)
    if "api_host" not in skip:
        ent = {"key": "api_host", "kernel": "k_tiles_fast<pos+vel,DELTA> + redo, then device -> host over PCIe",
               "workload": "the reference's flagship Python call, host arrays out: SatrecArray.sgp4(jd, fr) -> (e, r, v) numpy, %d x 1,440, "
                           "fp64 TEME pos+vel, time-major physical layout (api.py L296-320; the call the reference's 290 M/s figure is "
                           "quoted on).  Host wall clock per call INCLUDING fresh result arrays, staging, kernels and the two 466-MB "
                           "device-to-host copies" % n2}
        try:
            from astroz_amd.api import Satrec, SatrecArray
            arr = SatrecArray([Satrec.twoline2rv(a_, b_) for a_, b_ in pairs2], device=cuda.index or 0)
            jd = np.full(1440, synth.START_JD)
            fr = 0.32853009 + np.arange(1440) / 1440.0

            def calls(k):
                ws = []
                e_ = r_ = v_ = None
                for _ in range(k):
                    del e_, r_, v_          # (the previous result is returned to the OS outside the timed call)
                    t0 = time.perf_counter()
                    res_ = arr.sgp4(jd, fr)
                    ws.append((time.perf_counter() - t0) * 1e3)
                    e_, r_, v_ = res_
                    del res_
                return ws, (e_, r_, v_)
            calls(2)
            ws, (e_, r_, v_) = calls(7)
            _native.set_pinned_results(False)          # numpy.empty results: pinned staging slots + host copy threads (round 4's path)
            ws1, _ = calls(4)
            _native.set_host_copy_threads(0)           # ... and the plain path: pageable D2H straight into the fresh arrays
            ws0, _ = calls(3)
            _native.set_host_copy_threads(-1)
            _native.set_pinned_results(True)
            ms = sorted(ws)[len(ws) // 2]
            out_bytes = r_.nbytes + v_.nbytes + e_.nbytes
            ent.update({"ms_per_step": ms, "min_ms": min(ws), "value": n2 * 1440 / (ms / 1e3), "unit": "propagations/s (host arrays, PCIe-inclusive)",
                        "n_sats": n2, "n_times": 1440, "calls_ms": ws, "d2h_GB_per_s_of_wall": out_bytes / (ms / 1e3) / 1e9,
                        "path": arr._dev.last_path(),
                        "numpy_empty_staged_ms": sorted(ws1[1:])[len(ws1[1:]) // 2], "numpy_empty_staged_calls_ms": ws1,
                        "direct_pageable_copy_ms": sorted(ws0)[len(ws0) // 2],
                        "what": "the result arrays are ndarrays over pinned blocks of the library's pool (azh_host_alloc; returned to the pool "
                                "when the caller drops them): the device-to-host DMA lands in them directly.  numpy_empty_staged_ms: the same "
                                "call with plain numpy.empty results (_native.set_pinned_results(False)): device -> pinned staging slots -> "
                                "the fresh arrays by host copy threads; direct_pageable_copy_ms: ... with plain pageable D2H copies (the "
                                "runtime pins the fresh range first).  PCIe Gen5 x16 moves the 932 MB in ~16.4 ms: that, not the 0.3-ms "
                                "kernel, bounds this call"})
            rws = _sample_rows(n2, 16)
            cat = oracle.Catalog.from_pairs([pairs2[i] for i in rws], oracle.WGS72)
            rjd = jd[0] + fr[0]
            _, p0, v0 = cat.propagate(((jd + fr) - rjd) * 1440.0, (rjd - arr._epochs[rws]) * 1440.0, layout=oracle.SAT_MAJOR)
            ent["parity"] = {"rows": int(len(rws)), "max_abs_dr_km": float(np.abs(r_[rws] - p0).max()),
                             "max_abs_dv_kms": float(np.abs(v_[rws] - v0).max()), "err_nonzero": int(np.count_nonzero(e_))}
            del arr, e_, r_, v_
        except Exception as exc:
            ent["failed"] = repr(exc)
        res.append(ent)
    dev3 = pairs3 = None
    if not {"config3_sat_major", "config3_time_major"} <= set(skip):
        pairs3 = synth.synth_catalog(n_near=13478, n_deep=1522, seed=20260926)
        dev3 = _native.DeviceConstellation.from_tle_lines(pairs3, _native.WGS72, cuda.index or 0)
        dev3.set_timing(False)
    if dev3 is not None:
        case("config3_sat_major", "config 3: 13,478 near-earth + 1,522 deep-space SDP4 x 1,440, fp64 TEME pos+vel, satellite-major "
             "(steady: resonance seeds of the grid cached in the handle; cold_grid_call_ms: a new grid every call)",
             "k_rows_fast + k_rows_deep + redo (+ k_deep_seed, k_prep_inc on a new grid)", dev3, pairs3, 1440, layout=SM, cold=True, rows=48)
        case("config3_time_major", "config 3, TIME-major output, fp64 TEME pos+vel",
             "k_tiles_fast + deep-space rows + redo", dev3, pairs3, 1440, layout=TM, rows=48)
        dev3.close()
    if "fused_screen" not in skip:
        ent = {"key": "fused_screen", "kernel": "k_rows_fast<SINK = screen> (+ generic pass over rejected windows), k_screen_finalize",
               "workload": "config 2 catalog, fused propagate + single-target conjunction screen (src/Constellation.zig L683-756): minimum "
                           "distance and its grid point of every satellite against satellite 0 over 1,440 steps, nothing stored; one "
                           "call = input staging + window plan + the kernels (azh_screen_target_host)"}
        try:
            times = np.arange(1440, dtype=np.float64)
            offs = (synth.START_JD - dev2.epochs) * 1440.0
            ks, ws = [], []
            dev2.set_timing(True)          # (the library's own event pair around the screen's kernels)
            for _ in range(8):
                t0 = time.perf_counter()
                d, ti = dev2.screen_target(times, 0, 500.0, offs)
                ws.append((time.perf_counter() - t0) * 1e3)
                ks.append(dev2.last_kernel_ms())
            dev2.set_timing(False)
            ms = sorted(ks)[len(ks) // 2]
            ent.update({"ms_per_step": ms, "value": dev2.n * 1440 / (ms / 1e3), "unit": "propagations/s",
                        "call_wall_ms_median": sorted(ws)[len(ws) // 2],
                        "what": "ms_per_step: the library's HIP event pair around the screen's kernels; call_wall_ms: host wall clock of the "
                                "whole call incl. staging and the D2H of the two result vectors"})
            cat = oracle.Catalog.from_pairs(pairs2, oracle.WGS72)
            d0, t0_ = cat.screen_target(times, 0, 500.0, offs)
            ent["parity"] = {"rows": int(dev2.n), "max_abs_dmin_km": float(np.abs(d - d0).max()),
                             "t_index_mismatches": int((ti != t0_).sum())}
        except Exception as exc:
            ent["failed"] = repr(exc)
        res.append(ent)
    if "screen_all" not in skip:
        ent = {"key": "screen_all", "kernel": "k_tiles_fast / k_propagate + the cell-list screen kernels (azh_screen_all_host)",
               "workload": "config 2 catalog, all-vs-all conjunction screen (SURVEY 8 f3; bindings/python/astroz/__init__.py L535-658, "
                           "bindings/python/src/conjunction.zig L152-260): every pair closer than 10 km at any of 120 one-minute "
                           "steps, propagate + spatial hash on the GPU, the positions never leave HBM; host wall clock per call"}
        try:
            times = np.arange(120, dtype=np.float64)
            offs = (synth.START_JD - dev2.epochs) * 1440.0
            ws = []
            for _ in range(4):
                t0 = time.perf_counter()
                pr, tt = dev2.screen_all(times, 10.0, offs)
                ws.append((time.perf_counter() - t0) * 1e3)
            ms = sorted(ws[1:])[len(ws[1:]) // 2]
            ent.update({"ms_per_step": ms, "calls_ms": ws, "value": dev2.n * len(times) / (ms / 1e3), "unit": "propagations/s screened",
                        "pairs_found": int(len(tt))})
            # parity: the same screen by the oracle on the first 24 steps (propagate + its own cell list), as sets of (t, i, j)
            cat = oracle.Catalog.from_pairs(pairs2, oracle.WGS72)
            _, rp, _ = cat.propagate(times[:24], offs, layout=oracle.SAT_MAJOR, velocities=False)
            pr0, tt0 = oracle.coarse_screen(rp, 10.0)
            keep = np.asarray(tt) < 24
            got = set(zip(np.asarray(tt)[keep].tolist(), np.asarray(pr)[keep, 0].tolist(), np.asarray(pr)[keep, 1].tolist()))
            want = set(zip(np.asarray(tt0).tolist(), np.asarray(pr0)[:, 0].tolist(), np.asarray(pr0)[:, 1].tolist()))
            ent["parity"] = {"steps": 24, "pairs": len(want), "missing": len(want - got), "extra": len(got - want)}
        except Exception as exc:
            ent["failed"] = repr(exc)
        res.append(ent)
    if "one_satellite" not in skip:
        ent = {"key": "one_satellite", "kernel": "k_one_fast (every wave fits its own 1,024 points; k_one_satellite behind it for what it hands over)",
               "workload": "one satellite (ISS-like, near-earth) x 10,000,000 times through azh_propagate_one_device: device-resident "
                           "tsince in, pos+vel out (reference: 30.8 M/s single-thread sgp4_array, README.md L25-33)"}
        try:
            n = 10_000_000
            ts = torch.linspace(0.0, 14400.0, n, dtype=torch.float64, device=cuda)
            po = torch.empty((n, 3), dtype=torch.float64, device=cuda)
            ve = torch.empty((n, 3), dtype=torch.float64, device=cuda)
            ms = timed(lambda: dev2.propagate_one_device(0, ts.data_ptr(), n, po.data_ptr(), ve.data_ptr(), None, sptr), 2, 5)
            nbytes = n * 56.0
            ent["segments_fast_handed_over"] = list(dev2.last_one_stats())
            # the same count of sorted RANDOM times: every segment is handed over to the generic kernel (one full step per point)
            tr = torch.sort(torch.rand(n, dtype=torch.float64, device=cuda, generator=torch.Generator(device=cuda).manual_seed(3)) * 14400.0).values
            ms_irr = timed(lambda: dev2.propagate_one_device(0, tr.data_ptr(), n, po.data_ptr(), ve.data_ptr(), None, sptr), 1, 3)
            ent["irregular_times"] = {"ms_per_step": ms_irr, "value": n / (ms_irr / 1e3), "segments_fast_handed_over": list(dev2.last_one_stats()),
                                      "frac": nbytes / (ms_irr / 1e3) / 1e9 / HBM_PEAK_GBS}
            dev2.propagate_one_device(0, ts.data_ptr(), n, po.data_ptr(), ve.data_ptr(), None, sptr)   # (po / ve: the uniform series again, for the parity below)
            torch.cuda.synchronize()
            del tr
            ent.update({"ms_per_step": ms, "value": n / (ms / 1e3), "unit": "propagations/s",
                        "roofline": {"bound": "hbm", "achieved": nbytes / (ms / 1e3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                     "frac": nbytes / (ms / 1e3) / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes_per_launch": nbytes}})
            cat = oracle.Catalog.from_pairs([pairs2[0]], oracle.WGS72)
            pick = _sample_rows(n, 4096)
            tsel = ts[torch.as_tensor(pick, device=cuda)].cpu().numpy()
            _, p0, v0 = cat.propagate(tsel, None, layout=oracle.SAT_MAJOR)
            ent["parity"] = {"points": int(len(pick)),
                             "max_abs_dr_km": float(np.abs(po[torch.as_tensor(pick, device=cuda)].cpu().numpy() - p0[0]).max()),
                             "max_abs_dv_kms": float(np.abs(ve[torch.as_tensor(pick, device=cuda)].cpu().numpy() - v0[0]).max())}
            # the same series through HOST pointers (Satrec.sgp4_array / sgp4_propagate_batch: tsince in, e / r / v numpy out)
            th = np.linspace(0.0, 14400.0, n)
            hw = []
            for _ in range(4):
                t0 = time.perf_counter()
                eh, rh, vh = dev2.propagate_one(0, th)
                hw.append((time.perf_counter() - t0) * 1e3)
                del eh, rh, vh
            ent["host_pointers"] = {"ms_per_call": sorted(hw[1:])[len(hw[1:]) // 2], "calls_ms": hw,
                                    "value": n / (sorted(hw[1:])[len(hw[1:]) // 2] / 1e3), "unit": "propagations/s (80 MB in, 490 MB out over PCIe, fresh arrays)"}
            del ts, po, ve, th
        except Exception as exc:
            ent["failed"] = repr(exc)
        res.append(ent)
    if "ingest" not in skip:
        ent = {"key": "ingest", "kernel": "host text reader (azh::parse_all, threads) + k_init",
               "workload": "SURVEY 8-f4: 13,478 TLEs of text -> device-resident constellation (azh_constellation_from_tle_text: parse, "
                           "H2D, element initialisation kernel, classification), host wall clock; and the host reader alone on the "
                           "same text repeated to 1,010,850 records (140 MB, config 5's catalog size)"}
        try:
            text = "\n".join(a + "\n" + b for a, b in pairs2).encode()
            n2 = len(pairs2)
            ws = []
            for _ in range(5):
                t0 = time.perf_counter()
                d = _native.DeviceConstellation.from_tle_text(text, _native.WGS72, cuda.index or 0)
                d.synchronize()
                ws.append((time.perf_counter() - t0) * 1e3)
                ok_n = d.n
                d.close()
            big = text + b"\n"
            big = big * 75
            cap = len(big) // 138 + 1
            buf = np.empty((cap, 16))
            k = ctypes.c_size_t(0)
            L = _native.lib()
            per = {}
            for thr in (1, 0):
                _native.set_parse_threads(thr)
                ts = []
                for _ in range(3):
                    t0 = time.perf_counter()
                    L.azh_parse_tle_text(big, len(big), buf.ctypes.data, cap, ctypes.byref(k))
                    ts.append((time.perf_counter() - t0) * 1e3)
                per["serial" if thr == 1 else "threads"] = {"ms": min(ts), "records": int(k.value), "records_per_s": k.value / (min(ts) / 1e3)}
            _native.set_parse_threads(0)
            t0 = time.perf_counter()
            d = _native.DeviceConstellation.from_tle_text(big, _native.WGS72, cuda.index or 0)
            d.synchronize()
            per["text_to_device"] = {"ms": (time.perf_counter() - t0) * 1e3, "records": int(d.n)}
            d.close()
            ent.update({"ms_per_step": float(np.median(ws)), "value": n2 / (float(np.median(ws)) / 1e3), "unit": "TLEs/s (text -> initialised on device)",
                        "n_sats": int(ok_n), "min_ms": float(min(ws)), "reader_1M": per, "host_cores": os.cpu_count()})
            del big, buf
        except Exception as exc:
            ent["failed"] = repr(exc)
        res.append(ent)
    if not {"config5_share", "config5_share_f32arith", "config5_share_fp64"} <= set(skip):
        try:
            pairs5 = synth.synth_catalog(n_near=125000, n_deep=0, seed=20260927)
            dev5 = _native.DeviceConstellation.from_tle_lines(pairs5, _native.WGS72, cuda.index or 0)
            dev5.set_timing(False)
            c5 = "config 5, ONE GPU's share of 8: 125,000 synthetic satellites (seed 20260927) x 10,000 one-minute steps, fp32 pos+vel (30 GB), satellite-major, "
            case("config5_share", c5 + "DEFAULT arithmetic: mixed precision (O(1) quantities fp64, small ones packed fp32; eccentric members fp64 rounded at the store)",
                 "k_rows_fast32<MIXED> (+ eccentric members, redo)", dev5, pairs5, 10000, layout=SM, f32=True, steps=5, warm=2, rows=24)
            case("config5_share_fp64", c5 + "fp64 arithmetic, every component rounded once at the store (azh_set_f32_mode(c, 2))",
                 "k_rows_fast<SINK_F32> (+ eccentric members, redo)", dev5, pairs5, 10000, layout=SM, f32=True, steps=5, warm=2, rows=24,
                 arith32="fp64")
            case("config5_share_f32arith", c5 + "OPT-IN packed fp32 arithmetic (azh_set_f32_mode(c, 1): 4 m / 6 mm/s)",
                 "k_rows_fast32 (+ eccentric members, redo)", dev5, pairs5, 10000, layout=SM, f32=True, steps=5, warm=2, rows=24,
                 arith32="packed")
            dev5.close()
        except Exception as exc:
            res.append({"key": "config5_share", "failed": repr(exc)})
    # the entries for the reference's own call shapes go LAST (a record that keeps only the tail of this line keeps them)
    last = ["config2_time_major", "config2_sat_major_jdfr", "config2_time_major_jdfr", "api_host"]
    res.sort(key=lambda e: last.index(e["key"]) if e.get("key") in last else -1)
    return res
