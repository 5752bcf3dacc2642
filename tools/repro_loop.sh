#!/bin/bash
# Dev tool: which box is this, and how often does the full-size reproducibility test fail on it?
echo "host $(hostname) gpu $(rocm-smi --showuniqueid 2>/dev/null | grep -i unique | head -1 | awk '{print $NF}')"
n=${1:-8}; fail=0
for i in $(seq 1 $n); do
  out=$(python -m pytest tests/test_gpu_benchmr.py -q -m gpu -x 2>&1 | grep "AssertionError: rep\|passed\|failed" | cut -c1-300 | head -2)
  case "$out" in *failed*) fail=$((fail+1)); echo "$out";; esac
done
echo "failures: $fail of $n"
