#!/bin/bash
# The per-rank slab of the 2048^3 / 8-GPU configuration with RCCL messages to itself on one GPU, under different limits on
# RCCL's channels (= workgroups of its send/receive kernel): step time, and the kernel's grid and duration from a trace.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/rccl_channels; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for ch in default 32 8 2; do
  if [ "$ch" = default ]; then unset NCCL_MAX_NCHANNELS NCCL_MIN_NCHANNELS NCCL_MAX_P2P_NCHANNELS NCCL_MIN_P2P_NCHANNELS;
  else export NCCL_MAX_NCHANNELS=$ch NCCL_MIN_NCHANNELS=1 NCCL_MAX_P2P_NCHANNELS=$ch NCCL_MIN_P2P_NCHANNELS=1; fi
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/t_$ch -- python3 $R/bench.py --self-exchange --grid 2048 --global-x 256 --steps 12 --warmup 6 --reps 1 > $O/log_$ch.txt 2>&1 || exit 1
  python3 - $O $ch <<'PY'
import csv, glob, json, sys
O, ch = sys.argv[1], sys.argv[2]
line = [l for l in open(f"{O}/log_{ch}.txt") if l.startswith("{")][-1]
d = json.loads(line)
rows = list(csv.DictReader(open(glob.glob(f"{O}/t_{ch}/*/*kernel_trace.csv")[0])))
rc = [r for r in rows if "rccl" in r["Kernel_Name"]]
m2 = sorted((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows if "march2" in r["Kernel_Name"])
dur = sorted((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rc)
print(f"channels {ch}: {d['ms_per_step']:.3f} ms/step, {d['value']/1e3:.1f} Gcells/s; RCCL kernel: {int(rc[-1]['Grid_Size_X'])//int(rc[-1]['Workgroup_Size_X'])} workgroups, median {dur[len(dur)//2]:.0f} us; "
      f"sweep launches: median {m2[len(m2)//2]:.0f} us, longest {m2[-1]:.0f} us, {sum(1 for x in m2 if x > 1.2 * m2[len(m2)//2])} of {len(m2)} slowed by more than 20 %")
PY
done
