#!/bin/bash
# Round-4 evidence run, part 1 (GPU box): bench lines for cfg3 / cfg4 / cfg5, the self-launched 2-rank rehearsals, rocprofv3
# kernel stats for cfg3 and cfg5, the B = 1 step latency.  Outputs under gpurun_out/r04/; summaries are copied into profiles/.
set -o pipefail
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/r04; mkdir -p $O
timeout -k 10 500 python bench.py --steps 200 --warmup 20 > $O/bench_cfg3.json 2> $O/bench_cfg3.err || { tail -20 $O/bench_cfg3.err; exit 1; }
echo cfg3 done; head -c 300 $O/bench_cfg3.json; echo
timeout -k 10 500 python bench.py --workload cfg4 --steps 200 --warmup 20 > $O/bench_cfg4.json 2> $O/bench_cfg4.err || { tail -20 $O/bench_cfg4.err; exit 1; }
echo cfg4 done; head -c 300 $O/bench_cfg4.json; echo
timeout -k 10 500 python bench.py --workload cfg5 --steps 30 --warmup 5 --cpu-sample 0 > $O/bench_cfg5.json 2> $O/bench_cfg5.err || { tail -20 $O/bench_cfg5.err; exit 1; }
echo cfg5 done; head -c 300 $O/bench_cfg5.json; echo
NFOPP_DIST_BACKEND=gloo timeout -k 10 300 python3 bench.py --gpus 2 --batch-per-gpu 1024 --cpu-sample 0 --steps 20 --warmup 3 --fit-iters 20 > $O/bench_2rank_cfg3.json 2> $O/bench_2rank_cfg3.err || { tail -20 $O/bench_2rank_cfg3.err; exit 1; }
NFOPP_DIST_BACKEND=gloo timeout -k 10 300 python3 bench.py --gpus 2 --batch-per-gpu 1024 --cpu-sample 0 --steps 10 --warmup 2 --fit-iters 20 --workload cfg5 --spin-up 5 > $O/bench_2rank_cfg5.json 2> $O/bench_2rank_cfg5.err || { tail -20 $O/bench_2rank_cfg5.err; exit 1; }
echo 2-rank rehearsals done
cd /tmp
rm -rf $O/prof3 $O/prof5
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof3 -- python3 $R/bench.py --steps 200 --warmup 20 --cpu-sample 0 --fit-iters 100 --spin-up 0 > $O/prof3.log 2>&1
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof5 -- python3 $R/bench.py --workload cfg5 --steps 30 --warmup 5 --cpu-sample 0 --fit-iters 50 --spin-up 0 > $O/prof5.log 2>&1
cd $R
find $O/prof3 -name "*kernel_stats.csv" | head -1 | xargs -r head -8 | cut -c1-200
find $O/prof5 -name "*kernel_stats.csv" | head -1 | xargs -r head -12 | cut -c1-200
timeout -k 10 300 python tools/b1_latency.py 2>&1 | grep -v amdgpu.ids | tee $O/b1_latency.txt
timeout -k 10 200 python tools/multistep_probe.py 2>&1 | grep -v amdgpu.ids | tee $O/multistep_probe.txt
