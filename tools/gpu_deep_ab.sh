for v in head w3; do echo "== $v"; ASTROZ_AMD_LIB=tools/variants/lib_$v.so python tools/deep_probe.py sat 2>&1 | grep -E "mix6064"; done
SWEEP_ARGS="--deep 1522" TAG=sw17 TILES="" bash tools/gpu_sweep.sh
SWEEP_ARGS="--deep 1522" TAG=sw17 TILES="" bash tools/gpu_sweep.sh
SWEEP_ARGS="--deep 1522 --layout time" TAG=sw17 TILES="" bash tools/gpu_sweep.sh
