#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -q -m gpu ${PYTEST_ARGS:-} > gpurun_out/pytest_gpu.log 2>&1
rc=$?
tail -${TAIL:-60} gpurun_out/pytest_gpu.log
exit $rc
