#!/bin/bash
# The measurements kept under profiles/, collected on the GPU box in one go (run through gpurun): the bench line, its
# rocprofv3 stats and counter traffic, the side configurations' bench lines, the damped two-step sweep's stats, and
# the reference driver's CSV.  usage: tools/collect_evidence.sh [tag]   (results in gpurun_out/evidence_<tag>/)
TAG=${1:-latest}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/evidence_$TAG; mkdir -p $O
cd $R
timeout -k 10 300 python3 bench.py --steps 45 --warmup 5 > $O/bench_line.json 2> $O/bench_line.err || exit 1
bash tools/profile_bench.sh $TAG > $O/profile_bench.log 2>&1 || exit 1
timeout -k 10 200 python3 bench.py --storage f16 --no-cpu-baseline > $O/bench_fp16_storage.json 2>> $O/side.err || exit 1
timeout -k 10 200 python3 bench.py --space-order 8 --no-cpu-baseline > $O/bench_space_order8.json 2>> $O/side.err || exit 1
timeout -k 10 200 python3 bench.py --velocity layered --no-cpu-baseline > $O/bench_layered_velocity.json 2>> $O/side.err || exit 1
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/fwd_stats -- python3 $R/tools/bench_forward.py --grid 512 --steps 40 > $O/fwd_stats.log 2>&1 || exit 1
cp $O/fwd_stats/*/*_kernel_stats.csv $O/forward_kernel_stats.csv
cd $R
bash tools/run_reference_driver.sh $TAG > $O/refdriver.log 2>&1 || exit 1
echo evidence done
