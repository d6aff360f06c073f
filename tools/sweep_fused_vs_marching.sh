#!/bin/bash
# forced fused vs forced marching at sizes around the thresholds (warm-up with pairs)
for n in ${SIZES:-192 224 256 320 384 448 576 704}; do
  for v in march march2; do
    timeout -k 10 120 python bench.py --no-cpu-baseline --grid $n --variant $v --steps ${STEPS:-200} --warmup 20 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$n $v', round(d['value']/1e3,1), d['ms_per_step'])" || exit 1
  done
done
