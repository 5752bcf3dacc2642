# Dev (GPU box): pass 2 of the fit alone (pass 1 skipped after two calls) in both factor orders; needs build/skip1 (-DNFOPP_DEV_SKIP_PASS1)
export TMPDIR=/tmp; R=$PWD; export NFOPP_DEV_LIB=$R/build/skip1/libnfopp_hip.so
for mp in 2 1 2 1; do
  echo "== matrix path $mp (2: old slot order, 1: x32 order): both passes, then pass 2 alone"
  NFOPP_MATRIX_PATH=$mp python tools/train_speed.py 2>&1 | grep "P="
  NFOPP_MATRIX_PATH=$mp NFOPP_DEV_SKIP_PASS1=1 python tools/train_speed.py 2>&1 | grep "P="
done
