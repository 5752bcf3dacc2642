#!/bin/bash
# quick loop on the GPU box: ONF/trajectory parity tests + short bench
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/pytest_quick.log 2>&1 || { tail -40 gpurun_out/pytest_quick.log; exit 1; }
tail -2 gpurun_out/pytest_quick.log
timeout -k 10 300 python bench.py --steps ${STEPS:-100} --warmup 10 --cpu-sample 0 --fit-iters ${FIT:-100} > gpurun_out/bench_quick.json 2> gpurun_out/bench_quick.err || { tail -20 gpurun_out/bench_quick.err; exit 1; }
python - <<'PY'
import json
d=json.load(open("gpurun_out/bench_quick.json"))
print("value %.4g evals/s  ms/step %.4f  K1 ms %.4f  frac %.4f" % (d["value"], d["ms_per_step"], d["roofline"]["kernel_ms"], d["roofline"]["frac"]))
PY
