#!/bin/bash
# Round-4 dev loop: K2 A/B against the round-3 kernel (same process, outputs compared bit for bit), the whole GPU suite, K5 timings.
set -o pipefail
mkdir -p gpurun_out
python tools/ab_libs.py build/k2old/libnfopp_hip.so k2 2>&1 | tee gpurun_out/r4_k2_ab.txt
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r4_gpu_tests.log 2>&1 || { tail -40 gpurun_out/r4_gpu_tests.log; exit 1; }
tail -3 gpurun_out/r4_gpu_tests.log
bash tools/run_train_kernels.sh product 2>&1 | tee gpurun_out/r4_k5_times.txt
NFOPP_DEV_LIB=$PWD/build/wgprof/libnfopp_hip.so python tools/train_speed.py 2>&1 | grep -v "^P=" | sort | uniq -c | sort -rn | awk '{$1=""; print}' | cut -c1-330 | sort -u -k1,3 | head -4
