#!/bin/bash
# Final check of a round (GPU box): the whole -m gpu suite, smoke(), the default bench line.
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/final_gpu_tests.log 2>&1; rc=$?
tail -5 gpurun_out/final_gpu_tests.log
[ $rc -eq 0 ] || exit $rc
python -c "import __graft_entry__ as g; g.smoke()" || exit 1
timeout -k 10 400 python3 bench.py > gpurun_out/final_bench.json 2> gpurun_out/final_bench.err; echo "bench rc=$?"
python3 - <<'PY'
import json
d = json.loads(open("gpurun_out/final_bench.json").read().strip().splitlines()[-1])
print("ms/step %.4f value %.4g k1 %.4f frac %.3f parity %s cpu %.3g" % (d["ms_per_step"], d["value"], d["roofline"]["kernel_ms"], d["roofline"]["frac"], d["parity"]["ok"], d["cpu_baseline"]["value"]))
PY
