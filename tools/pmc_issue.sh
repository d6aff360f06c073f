#!/bin/bash
# Issue-slot accounting of the dominant stencil kernel of bench.py: how much of a wave's life is vector
# arithmetic, how much is waiting.  One --pmc pass per block (SQ: 8 counters; GRBM on its own).
# usage: tools/pmc_issue.sh <tag> [bench.py arguments]     e.g.  tools/pmc_issue.sh f32   /   tools/pmc_issue.sh f16 --storage f16
TAG=$1; shift
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_issue_$TAG; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_ACTIVE_INST_LDS --output-format csv -d $O/sq -- python3 $R/bench.py --no-cpu-baseline --reps 1 --steps 10 "$@" > $O/sq.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $O/grbm -- python3 $R/bench.py --no-cpu-baseline --reps 1 --steps 10 "$@" > $O/grbm.log 2>&1 || exit 1
timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --no-cpu-baseline --reps 1 --steps 10 "$@" > $O/stats.log 2>&1 || exit 1
python3 - "$O" <<'PY'
import csv, glob, sys, collections
O = sys.argv[1]
acc = collections.defaultdict(list)
kname = None
for d in ("sq", "grbm"):
    for r in csv.DictReader(open(glob.glob(f"{O}/{d}/*/*_counter_collection.csv")[0])):
        k = r["Kernel_Name"]
        if "stencil_march" in k:
            kname = k.split("(")[0]
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
avg = {k: sum(v) / len(v) for k, v in acc.items()}
dur = None
for r in csv.DictReader(open(glob.glob(f"{O}/stats/*/*_kernel_stats.csv")[0])):
    if "stencil_march" in r["Name"]:
        dur = float(r["AverageNs"]); break
print(f"kernel {kname}: average duration {dur/1e3:.1f} us")
for k, v in sorted(avg.items()):
    print(f"{k:22s} per launch {v:.4g}")
wc = avg["SQ_WAVE_CYCLES"]
print(f"share of wave life (SQ_* in quad-cycles): issuing anything {avg['SQ_ACTIVE_INST_ANY']/wc:.3f}, "
      f"issuing VALU {avg['SQ_ACTIVE_INST_VALU']/wc:.3f}, parked at a wait/barrier {avg['SQ_WAIT_ANY']/wc:.3f}, "
      f"issue stalled {avg['SQ_WAIT_INST_ANY']/wc:.3f}")
clk = avg["GRBM_GUI_ACTIVE"] / 8 / (dur * 1e-9) / 1e9
simd_cycles = dur * 1e-9 * clk * 1e9 * 1024          # 256 CUs x 4 SIMDs
print(f"effective clock {clk:.2f} GHz; VALU issue cycles / SIMD cycles available = {avg['SQ_ACTIVE_INST_VALU']*4/simd_cycles:.3f} "
      f"({avg['SQ_INSTS_VALU']/1024:.0f} VALU instructions per SIMD and launch, {avg['SQ_ACTIVE_INST_VALU']*4/avg['SQ_INSTS_VALU']:.2f} cycles each)")
PY
