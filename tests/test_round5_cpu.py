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
