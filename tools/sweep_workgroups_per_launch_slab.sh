#!/bin/bash
run() { FDTD_FUSE2_WGS=$1 timeout -k 10 170 python bench.py --no-cpu-baseline --self-exchange --grid 2048 --global-x 256 --steps 40 --warmup 6 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('slab256x2048x2048 wgs=$1', round(d['value']/1e3,1), d['ms_per_step'])" || exit 1; }
for w in 1024 256 512; do run $w; done
run2() { FDTD_FUSE2_WGS=$1 timeout -k 10 170 python bench.py --no-cpu-baseline --self-exchange --grid 1024 --global-x 512 --steps 40 --warmup 6 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('slab512x1024x1024 wgs=$1', round(d['value']/1e3,1), d['ms_per_step'])" || exit 1; }
for w in 256 512; do run2 $w; done
