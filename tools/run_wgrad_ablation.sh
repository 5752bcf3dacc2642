#!/bin/bash
# Dev tool: kernel time of onf_wgrad_split_kernel in the ablation builds (make ... EXTRA=-DNFOPP_ABL2_<x>  OUT=build/abl2_<x>)
export TMPDIR=/tmp; R=$PWD; mkdir -p gpurun_out
for v in "" NO_MFMA NO_SPLIT NO_FRAGREAD; do
  lib=$R/pytorch-motion-planner_amd/nfopp/lib/libnfopp_hip.so
  [ -n "$v" ] && lib=$R/build/abl2_$v/libnfopp_hip.so
  [ -f $lib ] || continue
  d=$R/gpurun_out/prof_abl2_${v:-product}
  rm -rf $d
  (cd /tmp && NFOPP_DEV_LIB=$lib rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 $R/tools/train_speed.py > /dev/null 2>&1)
  f=$(find $d -name "*kernel_stats.csv" | head -1)
  echo "${v:-product}: $(grep wgrad_split $f | cut -d, -f2-4,6,7)"
done
