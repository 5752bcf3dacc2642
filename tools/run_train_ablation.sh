# Dev: per-kernel times of the ONF fitting gradient at cfg5 scale for the product build and ablation builds (build/abl_*)
export TMPDIR=/tmp; R=$PWD; mkdir -p gpurun_out
for v in product $(ls build | grep '^abl_'); do
  lib=""; [ "$v" != product ] && lib=$R/build/$v/libnfopp_hip.so
  cd /tmp && NFOPP_DEV_LIB=$lib rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_abl_$v -- python3 $R/tools/train_speed.py > $R/gpurun_out/abl_$v.log 2>&1; cd $R
  echo "== $v"; grep "P=" gpurun_out/abl_$v.log | tail -1
  f=$(find gpurun_out/prof_abl_$v -name "*kernel_stats.csv" | head -1); head -5 $f | cut -d, -f1-4 | cut -c1-120
done
