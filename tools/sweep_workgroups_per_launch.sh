#!/bin/bash
# workgroups per launch of the fused sweep at cross-sections of more than 256 tiles
run() { FDTD_FUSE2_WGS=$2 timeout -k 10 170 python bench.py --no-cpu-baseline --grid $1 --variant march2 --steps ${STEPS:-40} --warmup 6 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1 wgs=$2', round(d['value']/1e3,1), d['ms_per_step'])" || exit 1; }
for w in 325 256 512 768 1024; do run 1032 $w; done
for w in 576 256 512 768 1024 1536; do run 1536 $w; done
