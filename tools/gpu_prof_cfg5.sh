#!/bin/bash
set -o pipefail
export TMPDIR=/tmp; R=$PWD; mkdir -p gpurun_out; rm -rf gpurun_out/prof5
cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof5 -- python3 $R/bench.py --workload cfg5 --steps 30 --warmup 5 --cpu-sample 0 --fit-iters 50 --spin-up 0 > $R/gpurun_out/bench_cfg5_prof.json 2> $R/gpurun_out/prof5_run.log
cd $R
f=$(find gpurun_out/prof5 -name "*kernel_stats.csv" | head -1); head -16 $f | cut -c1-170
