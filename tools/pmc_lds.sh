#!/bin/bash
# LDS and issue counters of the dominant kernel of bench.py (one --pmc pass, SQ block only).
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_lds; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $O -- python3 $R/bench.py --no-cpu-baseline --steps 10 > $O.log 2>&1 || exit 1
python3 - "$O" <<'PY'
import csv, glob, sys, collections
rows = list(csv.DictReader(open(glob.glob(sys.argv[1] + "/*/*_counter_collection.csv")[0])))
acc = collections.defaultdict(list)
for r in rows:
    if "stencil_march2" in r["Kernel_Name"]:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    print(f"{k:24s} per launch {sum(v)/len(v):.4g}  ({len(v)} launches)")
PY
