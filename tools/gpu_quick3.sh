#!/bin/bash
# parity (all gpu tests) + short bench + rocprofv3 kernel stats
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 800 python -m pytest tests -x -q -m gpu > gpurun_out/pytest_gpu.log 2>&1 || { tail -40 gpurun_out/pytest_gpu.log; exit 1; }
tail -2 gpurun_out/pytest_gpu.log
export TMPDIR=/tmp
R=$PWD
rm -rf $R/gpurun_out/prof
cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof -- python3 $R/bench.py --steps 100 --warmup 10 --cpu-sample 0 --fit-iters 50 > $R/gpurun_out/bench_prof.json 2> $R/gpurun_out/prof_run.log
cd $R
python - <<'PY'
import json,glob,csv
d=json.loads([l for l in open("gpurun_out/bench_prof.json") if l.startswith("{")][-1])
print("value %.4g evals/s  ms/step %.4f  K1 ms %.4f  frac %.4f (under rocprof)" % (d["value"], d["ms_per_step"], d["roofline"]["kernel_ms"], d["roofline"]["frac"]))
f=glob.glob("gpurun_out/prof/**/*kernel_stats.csv",recursive=True)[0]
for row in csv.DictReader(open(f)):
    if "nfopp" in row["Name"]: print("%-60s calls %5s avg %10.1f ns" % (row["Name"][:60], row["Calls"], float(row["AverageNs"])))
PY
