#!/bin/bash
# sample DPM clocks while a bench variant runs back to back
D=$(ls -d /sys/class/drm/card*/device | head -1)
ls $D | grep -i "pp_dpm\|power" | tr '\n' ' '; echo
for args in "--layout time" "--deep 1522 --layout time" "--deep 1522"; do
  echo "=== $args"
  python bench.py --no-cpu-baseline --no-secondary --steps 6000 --warmup 200 $args > /tmp/b.json 2>/dev/null &
  BP=$!
  sleep 9
  for i in 1 2 3; do
    for f in pp_dpm_sclk pp_dpm_mclk pp_dpm_fclk pp_dpm_socclk; do [ -f $D/$f ] && echo "$f: $(grep '\*' $D/$f | tr '\n' ' ')"; done
    sleep 0.3
  done
  wait $BP
  python tools/show_bench.py /tmp/b.json
done
