#!/bin/bash
# ab_lib.sh "libA.so libB.so ..." -- bench.py arguments: one bench.py line per build of the library ("-" = the in-tree one).
libs=$1; shift 2
for l in $libs; do
  if [ "$l" = "-" ]; then unset FDTD_HIP_LIB; else export FDTD_HIP_LIB=$PWD/$l; fi
  line=$(timeout -k 10 180 python bench.py --no-traffic --no-breakdown --no-cpu-baseline "$@" | tail -1) || exit 1
  python -c 'import sys, json; d = json.loads(sys.argv[2]); print(sys.argv[1], round(d["value"] / 1e3, 1), "Gcells/s frac", round(d["roofline"]["frac"], 4))' "$l" "$line" || exit 1
done
