# Dev (GPU box): the ONF fit at cfg5 scale (P = 2 543 616) with an older build and the product build, per-kernel times of both.
# Usage: bash tools/run_train_ab.sh build/OLD/libnfopp_hip.so
export TMPDIR=/tmp; R=$PWD; mkdir -p gpurun_out
OLD=${1:-build/look6/libnfopp_hip.so}
NEW=$R/pytorch-motion-planner_amd/nfopp/lib/libnfopp_hip.so
for tag in old new old new; do
  lib=$([ $tag = old ] && echo $R/$OLD || echo $NEW)
  echo "== $tag"; NFOPP_DEV_LIB=$lib python tools/train_speed.py 2>&1 | grep "P="
done
for tag in old new; do
  lib=$([ $tag = old ] && echo $R/$OLD || echo $NEW)
  export NFOPP_DEV_LIB=$lib
  cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_train_$tag -- python3 $R/tools/train_speed.py > /dev/null 2>&1; cd $R
  f=$(find gpurun_out/prof_train_$tag -name "*kernel_stats.csv" | head -1); echo "== $tag"; head -6 $f | cut -d, -f1-4 | cut -c1-150
done
