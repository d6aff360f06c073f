#!/bin/bash
# usage: tools/pmc.sh <outdir-name> <counters...>   (env selects the kernel variant); run from anywhere on the GPU box
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; shift
cd /tmp && export TMPDIR=/tmp
KBENCH_NO_STREAM=1 timeout -s KILL 90 rocprofv3 --pmc "$@" --output-format csv -d $O -- $R/tools/kbench 512 3 > $O.log 2>&1
python3 - "$O" <<'PY'
import csv,glob,collections,sys
f=glob.glob(sys.argv[1]+'/*/*_counter_collection.csv')[0]
agg=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if 'fill' in r['Kernel_Name']: continue
    agg[(r['Kernel_Name'][:44],r['Counter_Name'])].append(float(r['Counter_Value']))
for k,v in sorted(agg.items()): print(k[0],k[1],len(v),'%.4g'%(sum(v)/len(v)))
PY
