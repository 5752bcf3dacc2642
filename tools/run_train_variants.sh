#!/bin/bash
# Dev tool: kernel times of the ONF fit (pass 1 = onf_split_kernel<.., MODE 1>, pass 2 = onf_wgrad_*) for development builds.
# Usage: bash tools/run_train_variants.sh build/<variant>/libnfopp_hip.so ...   (the product build is always measured first)
export TMPDIR=/tmp; R=$PWD; mkdir -p gpurun_out
for lib in $R/pytorch-motion-planner_amd/nfopp/lib/libnfopp_hip.so "$@"; do
  case $lib in /*) ;; *) lib=$R/$lib;; esac
  [ -f $lib ] || { echo "missing $lib"; continue; }
  d=$R/gpurun_out/prof_variant; rm -rf $d
  (cd /tmp && NFOPP_DEV_LIB=$lib rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 $R/tools/train_speed.py > $d.log 2>&1)
  f=$(find $d -name "*kernel_stats.csv" | head -1)
  echo "== $lib"; grep "P=" $d.log | cut -c1-90
  grep "onf_split_kernel\|onf_fwd_bwd\|wgrad_split\|wgrad_kernel" $f | awk -F'","' '{printf "   %-60s calls %s  min %s ns  max %s ns\n", substr($1,2,60), $2, $6, $7}'
done
