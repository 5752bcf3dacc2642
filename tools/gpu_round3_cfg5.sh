#!/bin/bash
# Round-3 evidence for the ONF fit (cfg5): bench line + rocprofv3 kernel stats on one box, then the PMC passes of both fit kernels.
set -o pipefail
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/r03; mkdir -p $O
timeout -k 10 500 python bench.py --workload cfg5 --steps 30 --warmup 5 --cpu-sample 0 > $O/bench_cfg5.json 2> $O/bench_cfg5.err || { tail -20 $O/bench_cfg5.err; exit 1; }
echo cfg5 done; head -c 300 $O/bench_cfg5.json; echo
cd /tmp
rm -rf $O/prof5
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof5 -- python3 $R/bench.py --workload cfg5 --steps 30 --warmup 5 --cpu-sample 0 --fit-iters 50 --spin-up 0 > $O/prof5.log 2>&1
cd $R
find $O/prof5 -name "*kernel_stats.csv" | head -1 | xargs -r head -8 | cut -c1-200
timeout -k 10 500 bash tools/gpu_pmc_wgrad.sh > $O/pmc_wgrad.log 2>&1; tail -3 $O/pmc_wgrad.log
