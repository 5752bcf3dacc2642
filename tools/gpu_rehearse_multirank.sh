#!/bin/bash
# Rehearsal of the N-rank bench flow on ONE GPU: plain `python3 bench.py --gpus 2` (no launcher around it: bench.py starts its own
# rank processes), gloo rendezvous, two ranks sharing the card: checks the self-launch, sharding, barriers, max-over-ranks timing,
# ranks_seen and the single JSON line.  Throughput numbers from this run are meaningless.
set -o pipefail
mkdir -p gpurun_out
NFOPP_DIST_BACKEND=gloo timeout -k 10 500 python3 bench.py --gpus 2 --steps 20 --warmup 3 --cpu-sample 0 --fit-iters 20 --batch-per-gpu 1024 ${EXTRA_ARGS:-} > gpurun_out/bench_2rank.json 2> gpurun_out/bench_2rank.err || { tail -20 gpurun_out/bench_2rank.err; exit 1; }
grep -c '"metric"' gpurun_out/bench_2rank.json
python3 - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/bench_2rank.json") if l.startswith("{")][-1])
print(d["n_gpus"], d["ranks_seen"], d["backend"], d["config"]["global_batch"], d["config"]["paths_finite"], "%.3g" % d["value"])
PY
