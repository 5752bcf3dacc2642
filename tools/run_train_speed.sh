export TMPDIR=/tmp; R=$PWD; mkdir -p gpurun_out
python tools/train_speed.py 2>&1 | grep "P="
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_train -- python3 $R/tools/train_speed.py > /dev/null 2>&1; cd $R
f=$(find gpurun_out/prof_train -name "*kernel_stats.csv" | head -1); head -6 $f | cut -c1-160
