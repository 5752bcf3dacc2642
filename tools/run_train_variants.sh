# Dev (GPU box): fit time at cfg5 scale for variant libraries, interleaved with the product build.  Usage: bash tools/run_train_variants.sh nost t256 ...
export TMPDIR=/tmp; R=$PWD
for rep in 1 2; do
  for v in product "$@"; do
    lib=$([ $v = product ] && echo $R/pytorch-motion-planner_amd/nfopp/lib/libnfopp_hip.so || echo $R/build/$v/libnfopp_hip.so)
    printf "%-10s " $v; NFOPP_DEV_LIB=$lib python tools/train_speed.py 2>&1 | grep "P="
  done
done
