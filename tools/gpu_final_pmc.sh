#!/bin/bash
# re-stamp the headline's counter pass on the final sources (bench.py reports roofline.traffic only when the fingerprint in
# profiles/latest_pmc.json is the one of the sources it runs on), then the default line once more
mkdir -p gpurun_out/profiles gpurun_out/final
timeout 900 python tools/profile_run.py r06_config2_sat_major --pmc -- > gpurun_out/final/prof.log 2>&1; tail -2 gpurun_out/final/prof.log
cp gpurun_out/profiles/r06_config2_sat_major.json gpurun_out/profiles/latest_pmc.json
cp gpurun_out/profiles/latest_pmc.json profiles/latest_pmc.json
timeout 900 python bench.py > gpurun_out/final/bench_default.out 2> gpurun_out/final/bench_default.err; tail -1 gpurun_out/final/bench_default.out > gpurun_out/final/bench_default_line.json
cp gpurun_out/bench_full.json gpurun_out/final/bench_default_full.json
python tools/show_bench.py gpurun_out/final/bench_default_line.json | head -5
rm -rf gpurun_out/prof_raw
