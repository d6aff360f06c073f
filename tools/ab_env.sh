#!/bin/bash
# ab_env.sh VAR "v1 v2 ..." -- bench.py arguments: one bench.py line per value of an environment switch.
# Prints value, Gcells/s and roofline fraction; stops at the first failure.
var=$1; vals=$2; shift 3
for v in $vals; do
  line=$(env "$var=$v" timeout -k 10 180 python bench.py --no-traffic --no-breakdown "$@" | tail -1) || exit 1
  python -c 'import sys, json; d = json.loads(sys.argv[3]); print(sys.argv[1], "=", sys.argv[2], round(d["value"] / 1e3, 1), "Gcells/s frac", round(d["roofline"]["frac"], 4))' "$var" "$v" "$line" || exit 1
done
