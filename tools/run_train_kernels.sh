# Dev (GPU box): per-kernel times (rocprofv3) of the fit at cfg5 scale for variant libraries.  Usage: bash tools/run_train_kernels.sh product nost ...
export TMPDIR=/tmp; R=$PWD
for v in "$@"; do
  lib=$([ $v = product ] && echo $R/pytorch-motion-planner_amd/nfopp/lib/libnfopp_hip.so || echo $R/build/$v/libnfopp_hip.so)
  export NFOPP_DEV_LIB=$lib
  rm -rf $R/gpurun_out/prof_tk_$v
  cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_tk_$v -- python3 $R/tools/train_speed.py > /dev/null 2>&1; cd $R
  f=$(find gpurun_out/prof_tk_$v -name "*kernel_stats.csv" | head -1)
  echo "== $v"; python3 -c "
import csv,sys
for r in csv.DictReader(open('$f')):
    if 'nfopp' in r['Name'] and float(r['AverageNs']) > 20000: print('  %-64s %8.1f us' % (r['Name'][:64], float(r['AverageNs'])/1e3))"
done
