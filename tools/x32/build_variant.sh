#!/bin/bash
# Dev: build a variant of libnfopp_hip.so that differs from the product build in the compile flags of onf_x32*.hip only.
# Usage: tools/x32/build_variant.sh NAME "-DFLAG ..."   ->  build/NAME/libnfopp_hip.so   (for tools/x32/ab_multi.py)
set -e
R=$(cd "$(dirname "$0")/../.." && pwd)
N=$1; shift
make -C $R/pytorch-motion-planner_amd/csrc -j6 >/dev/null
rm -rf $R/build/csrc_$N && cp -r $R/build/csrc $R/build/csrc_$N && rm -f $R/build/csrc_$N/onf_x32*.o
make -C $R/pytorch-motion-planner_amd/csrc OBJ=$R/build/csrc_$N OUT=$R/build/$N EXTRA="$*" $R/build/$N/libnfopp_hip.so 2>&1 | grep -E "error|warning" || true
ls -la $R/build/$N/libnfopp_hip.so
