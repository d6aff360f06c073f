#!/bin/bash
# bench.py over the reference driver's size sweep (main.cpp:279): one line per size with the whole-job
# rate, the time per step and the dominant kernel's average duration.  Run through gpurun.
OUT=${1:-gpurun_out/bench_sweep.txt}
SIZES=${2:-"32 64 96 128 192 256 384 512 640 768"}
: > $OUT
for n in $SIZES; do
  timeout -k 10 200 python3 bench.py --grid $n --no-cpu-baseline --reps 3 > gpurun_out/_sweep_line.json 2>> gpurun_out/_sweep.err || exit 1
  python3 - $n >> $OUT <<'PY'
import json, sys
d = json.load(open("gpurun_out/_sweep_line.json"))
r = d["roofline"]
print(f"{sys.argv[1]:>5}^3  {d['value']/1e3:8.1f} Gcells/s  {d['ms_per_step']*1e3:9.2f} us/step  kernel {r['avg_kernel_ms']*1e3:9.2f} us "
      f"({r['cell_updates_per_launch']/float(sys.argv[1])**3:.0f} updates/launch)  frac {r['frac']:.3f}  stddev {d['reps']['Mcells_per_s_stddev']/1e3:.1f}")
PY
done
cat $OUT
