#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
python tools/ab_libs.py build/k2old/libnfopp_hip.so k2 2>&1 | tee gpurun_out/r4_k2_ab.txt
bash tools/run_train_kernels.sh product genfeat k5old product genfeat k5old 2>&1 | tee gpurun_out/r4_k5_times.txt
