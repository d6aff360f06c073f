#!/usr/bin/env python3
"""Scans gfx950 assembly (hipcc -S --cuda-device-only) for a 16-byte buffer store whose data registers are written by the
very next instruction.  The compiler leaves no wait state there when the store's scalar-offset field holds a register (the
documented gfx9 rule); on MI355X such a pair was seen to store the overwritten values in some lanes.

    hipcc --offload-arch=gfx950 -O3 ... -S --cuda-device-only -o /tmp/k.s csrc/fdtd_march2.hip && python tools/check_store_hazard.py /tmp/k.s
Exit code 1 if a pair is found."""
import re
import sys

lines = open(sys.argv[1]).read().split("\n")
found = total = 0
for i, l in enumerate(lines):
    m = re.search(r"buffer_store_dwordx[34] v\[(\d+):(\d+)\]", l)
    if not m:
        continue
    total += 1
    lo, hi = int(m.group(1)), int(m.group(2))
    j = i + 1
    while j < len(lines) and (not lines[j].startswith("\t") or lines[j].strip().startswith((";", "."))):
        j += 1
    if j >= len(lines):
        continue
    nx = lines[j].strip()
    w = re.match(r"(v_\S+|ds_read\S*|buffer_load\S*|global_load\S*)\s+v\[?(\d+)(?::(\d+))?\]?", nx)
    if w and not nx.startswith("v_cmp"):
        a = int(w.group(2)); b = int(w.group(3) or a)
        if not (b < lo or a > hi):
            found += 1
            print(f"line {i + 1}: {l.strip()[:90]}\n      -> {nx[:90]}")
print(f"{found} of {total} wide buffer stores are followed directly by a write of their data registers")
sys.exit(1 if found else 0)
