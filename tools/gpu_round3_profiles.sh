#!/bin/bash
# Round-3 evidence run (on the GPU box via gpurun): bench lines for cfg3 / cfg4 / cfg5, rocprofv3 kernel stats for cfg3 and
# cfg5, PMC passes for the fused ONF kernel.  Outputs under gpurun_out/r03/; the summaries are copied into profiles/.
set -o pipefail
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/r03; mkdir -p $O
timeout -k 10 500 python bench.py --steps 200 --warmup 20 > $O/bench_cfg3.json 2> $O/bench_cfg3.err || { tail -20 $O/bench_cfg3.err; exit 1; }
echo cfg3 done; head -c 400 $O/bench_cfg3.json; echo
timeout -k 10 500 python bench.py --workload cfg4 --steps 200 --warmup 20 > $O/bench_cfg4.json 2> $O/bench_cfg4.err || { tail -20 $O/bench_cfg4.err; exit 1; }
echo cfg4 done; head -c 400 $O/bench_cfg4.json; echo
timeout -k 10 500 python bench.py --workload cfg5 --steps 30 --warmup 5 --cpu-sample 0 > $O/bench_cfg5.json 2> $O/bench_cfg5.err || { tail -20 $O/bench_cfg5.err; exit 1; }
echo cfg5 done; head -c 300 $O/bench_cfg5.json; echo
cd /tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof3 -- python3 $R/bench.py --steps 200 --warmup 20 --cpu-sample 0 --fit-iters 100 --spin-up 0 > $O/prof3.log 2>&1
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof5 -- python3 $R/bench.py --workload cfg5 --steps 30 --warmup 5 --cpu-sample 0 --fit-iters 50 --spin-up 0 > $O/prof5.log 2>&1
cd $R
find $O/prof3 -name "*kernel_stats.csv" | head -1 | xargs -r head -8 | cut -c1-200
find $O/prof5 -name "*kernel_stats.csv" | head -1 | xargs -r head -12 | cut -c1-200
bash tools/gpu_pmc_x32.sh > $O/pmc.log 2>&1; tail -45 $O/pmc.log
cp gpurun_out/pmc_x32/summary_onf_x32_kernel.json $O/ 2>/dev/null
