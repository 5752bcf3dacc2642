#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_multistep.py tests/test_gpu_split_path.py tests/test_gpu_torch_ops.py -x -q > gpurun_out/r4_multistep_tests.log 2>&1 || { tail -40 gpurun_out/r4_multistep_tests.log; exit 1; }
tail -3 gpurun_out/r4_multistep_tests.log
timeout -k 10 300 python tools/b1_latency.py 2>&1 | tee gpurun_out/r4_b1_latency.txt
