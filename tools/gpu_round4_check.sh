#!/bin/bash
# Round 4, first GPU call: the whole -m gpu suite, the plain `bench.py --gpus 2` self-launch on ONE card (gloo rehearsal:
# throughput meaningless, the flow is what is checked) for cfg3 and cfg5, then the default bench line.
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 700 python -m pytest tests -m gpu -x -q > gpurun_out/r4_gpu_tests.log 2>&1 || { tail -30 gpurun_out/r4_gpu_tests.log; exit 1; }
tail -3 gpurun_out/r4_gpu_tests.log
NFOPP_DIST_BACKEND=gloo timeout -k 10 300 python3 bench.py --gpus 2 --batch-per-gpu 1024 --cpu-sample 0 --steps 20 --warmup 3 --fit-iters 20 > gpurun_out/r4_bench_2rank_cfg3.json 2> gpurun_out/r4_bench_2rank_cfg3.err || { tail -20 gpurun_out/r4_bench_2rank_cfg3.err; exit 1; }
NFOPP_DIST_BACKEND=gloo timeout -k 10 300 python3 bench.py --gpus 2 --batch-per-gpu 1024 --cpu-sample 0 --steps 10 --warmup 2 --fit-iters 20 --workload cfg5 --spin-up 5 > gpurun_out/r4_bench_2rank_cfg5.json 2> gpurun_out/r4_bench_2rank_cfg5.err || { tail -20 gpurun_out/r4_bench_2rank_cfg5.err; exit 1; }
python3 - <<'PY'
import json
for w in ("cfg3", "cfg5"):
    d = json.loads(open("gpurun_out/r4_bench_2rank_%s.json" % w).read().strip().splitlines()[-1])
    print(w, "n_gpus", d["n_gpus"], "ranks_seen", d["ranks_seen"], d["backend"], "finite", d["config"]["paths_finite"])
PY
timeout -k 10 400 python3 bench.py > gpurun_out/r4_bench_cfg3.json 2> gpurun_out/r4_bench_cfg3.err; echo "bench rc=$?"
python3 - <<'PY'
import json
d = json.loads(open("gpurun_out/r4_bench_cfg3.json").read().strip().splitlines()[-1])
print("ms/step", d["ms_per_step"], "value", d["value"], "frac", d["roofline"]["frac"], "k1_ms", d["roofline"]["kernel_ms"])
print("parity ok", d["parity"]["ok"], json.dumps(d["parity"]["short"]["gate"]["checks"]))
PY
