#!/bin/bash
set -o pipefail
bash tools/gpu_quick.sh || exit 1
python tools/gpu_margins.py 2>/dev/null | grep -E "^G[13]"
