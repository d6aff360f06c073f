#!/bin/bash
b() { timeout -k 10 280 python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$*', round(d['value']/1e3,1), d['ms_per_step'], d['roofline']['frac'])" || exit 1; }
b
b --grid 256
b --grid 384
b --grid 480
b --grid 500
b --grid 640
b --grid 768
b --grid 896
b --grid 1024
b --grid 1536 --steps 20
b --self-exchange
b --self-exchange --grid 1024 --global-x 512
b --self-exchange --grid 2048 --global-x 256
