#!/bin/bash
# Runs on the GPU box (via gpurun): smoke, bench, rocprofv3 kernel trace.  Outputs under gpurun_out/.
set -o pipefail
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1 || { tail -20 gpurun_out/smoke.log; exit 1; }
tail -1 gpurun_out/smoke.log
timeout -k 10 500 python bench.py --steps ${STEPS:-100} --warmup 10 > gpurun_out/bench.json 2> gpurun_out/bench.err || { tail -30 gpurun_out/bench.err; exit 1; }
cat gpurun_out/bench.json
export TMPDIR=/tmp
R=$PWD
cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof -- python3 $R/bench.py --steps 200 --warmup 20 --cpu-sample 0 --fit-iters 100 --spin-up 0 > $R/gpurun_out/prof_run.log 2>&1
cd $R
find gpurun_out/prof -name "*kernel_stats.csv" | head -1 | xargs -r head -12
