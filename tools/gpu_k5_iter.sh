#!/bin/bash
# Dev loop for the ONF fit (K5): the fit's parity tests, per-kernel times at cfg5 scale, optional in-kernel stamps.
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_split_path.py tests/test_gpu_learning.py -x -q -k "train or fit or learning or onf_training" > gpurun_out/r4_k5_tests.log 2>&1 || { tail -40 gpurun_out/r4_k5_tests.log; exit 1; }
tail -2 gpurun_out/r4_k5_tests.log
bash tools/run_train_kernels.sh product ${K5_VARIANTS:-} 2>&1 | tee gpurun_out/r4_k5_times.txt
if [ -f build/wgprof/libnfopp_hip.so ]; then
  NFOPP_DEV_LIB=$PWD/build/wgprof/libnfopp_hip.so python tools/train_speed.py 2>&1 | grep -v "^P=" | sort | uniq -c | sort -rn | awk '{$1=""; print}' | cut -c1-330 | sort -u -k1,3 | head -4
fi
