#!/bin/bash
# rocprofv3 evidence for the marching sweep of the space orders and for the damped sweep (run through gpurun):
#   1. --kernel-trace --stats of tools/bench_orders.py (orders 6, 8, 12 at 512^3) and tools/bench_forward.py
#   2. separate --pmc passes for FETCH_SIZE and WRITE_SIZE of the order-8 sweep
# Results land in gpurun_out/profile_orders_<tag>/ ; copy the summaries into profiles/ afterwards.
TAG=${1:-r02}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/profile_orders_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/tools/bench_orders.py --orders 6,8,12 --modes 1 > $O/stats.log 2>&1 || exit 1
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_fw -- python3 $R/tools/bench_forward.py > $O/stats_fw.log 2>&1 || exit 1
timeout -s KILL 150 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 $R/tools/bench_orders.py --orders 8 --modes 1 --steps 6 > $O/fetch.log 2>&1 || exit 1
timeout -s KILL 150 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 $R/tools/bench_orders.py --orders 8 --modes 1 --steps 6 > $O/write.log 2>&1 || exit 1
python3 - "$O" <<'PY'
import csv, glob, json, sys
O = sys.argv[1]
out = {"kernel": "stencil_order_march_kernel<4,...> (space order 8)", "grid": 512}
for d, c in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    f = glob.glob(f"{O}/{d}/*/*_counter_collection.csv")[0]
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "stencil_order_march" in r["Kernel_Name"]]
    out[c + "_KB_per_launch"] = sum(vals) / len(vals)
    out[c + "_launches"] = len(vals)
# gfx950: FETCH_SIZE reports half of the bytes of a wide coalesced streaming read; WRITE_SIZE is exact
out["hbm_read_bytes_per_launch"] = 2 * 1024 * out["FETCH_SIZE_KB_per_launch"]
out["hbm_write_bytes_per_launch"] = 1024 * out["WRITE_SIZE_KB_per_launch"]
out["hbm_bytes_per_launch"] = out["hbm_read_bytes_per_launch"] + out["hbm_write_bytes_per_launch"]
out["algorithmic_bytes_per_launch"] = 16 * 512 ** 3
out["note"] = "FETCH_SIZE doubled (gfx950 counts 128-B requests as 64 B); separate --pmc passes"
json.dump(out, open(f"{O}/traffic.json", "w"), indent=1)
print(json.dumps(out))
PY
cp $O/stats/*/*_kernel_stats.csv $O/kernel_stats.csv
cp $O/stats_fw/*/*_kernel_stats.csv $O/kernel_stats_forward.csv
head -6 $O/kernel_stats.csv
head -4 $O/kernel_stats_forward.csv
