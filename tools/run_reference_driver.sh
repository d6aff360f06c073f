#!/bin/bash
# Runs the reference's unmodified driver (oracle/_ref/fdtd_benchmark, built by `make -C oracle driver`) against the
# engine on the GPU box and keeps its outputs: benchmark.csv and the console report of both binaries.
TAG=${1:-r02}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/refdriver_$TAG; rm -rf $O; mkdir -p $O
cd $O
# fresh zero pages for every driver buffer (it never initialises the third time level, main.cpp:520-532)
# (this binary's Kernel_OpenACC is the reference's serial host loop nest: only its STEP 1, the parity run, is wanted;
# its STEP 2 would time that serial code for minutes per size, so it is cut off there)
MALLOC_MMAP_THRESHOLD_=65536 timeout -k 10 100 stdbuf -oL $R/oracle/_ref/fdtd_benchmark_vs_ref > step1_vs_reference_openacc.txt 2>&1
rm -f benchmark.csv
grep -c "PASS" step1_vs_reference_openacc.txt
timeout -k 10 500 $R/oracle/_ref/fdtd_benchmark > console.txt 2>&1 || exit 1
grep -n "CUDA_Optimized" benchmark.csv | cut -d, -f1-9
