#!/bin/bash
# HBM traffic of the z-tiled two-step sweep at 1024^3 (8 sweep launches + 1 seam pass per two steps):
# separate --pmc passes for FETCH_SIZE and WRITE_SIZE, summed over all launches of a two-step call.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/profile_1024
mkdir -p $O; cd /tmp && export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 $R/bench.py --grid 1024 --no-cpu-baseline --steps 10 > $O/fetch.log 2>&1 || exit 1
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 $R/bench.py --grid 1024 --no-cpu-baseline --steps 10 > $O/write.log 2>&1 || exit 1
python3 - "$O" <<'PY'
import csv, glob, json, sys
O = sys.argv[1]
out = {"workload": "bench.py --grid 1024 --steps 10 (all two-step calls of the run, untimed ones included)"}
for d, c in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    rows = list(csv.DictReader(open(glob.glob(f"{O}/{d}/*/*_counter_collection.csv")[0])))
    for k in ("stencil_march2", "seam_kernel"):
        v = [float(r["Counter_Value"]) for r in rows if k in r["Kernel_Name"]]
        out[f"{c}_KB_{k}_total"] = sum(v)
        out[f"{c}_{k}_launches"] = len(v)
calls = out["FETCH_SIZE_seam_kernel_launches"]
rd = 2 * 1024 * (out["FETCH_SIZE_KB_stencil_march2_total"] + out["FETCH_SIZE_KB_seam_kernel_total"]) / calls
wr = 1024 * (out["WRITE_SIZE_KB_stencil_march2_total"] + out["WRITE_SIZE_KB_seam_kernel_total"]) / calls
out["two_step_calls"] = calls
out["hbm_read_bytes_per_two_steps"] = rd
out["hbm_write_bytes_per_two_steps"] = wr
out["seam_read_bytes_per_two_steps"] = 2 * 1024 * out["FETCH_SIZE_KB_seam_kernel_total"] / calls
out["seam_write_bytes_per_two_steps"] = 1024 * out["WRITE_SIZE_KB_seam_kernel_total"] / calls
out["compulsory_bytes_per_two_steps"] = 5 * 4 * 1024 ** 3
out["algorithmic_bytes_per_two_steps"] = 32 * 1024 ** 3
out["note"] = "FETCH_SIZE doubled (gfx950 counts 128-B requests as 64 B); separate --pmc passes"
json.dump(out, open(f"{O}/traffic_1024.json", "w"), indent=1)
print(json.dumps(out))
PY
