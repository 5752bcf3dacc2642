#!/bin/bash
set -o pipefail
export TMPDIR=/tmp; R=$PWD; mkdir -p $R/gpurun_out/pmck2; cd /tmp
run() { name=$1; shift
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $R/gpurun_out/pmck2/$name -- python3 $R/tools/k2_probe.py > $R/gpurun_out/pmck2/$name.log 2>&1 || echo "pass $name failed"; }
run a SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_TRANS_F32
run b GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES SQ_WAVES SQ_INST_CYCLES_VMEM_RD
run c FETCH_SIZE
run d WRITE_SIZE
cd $R
python3 - <<'PY'
import csv, glob, collections
for name in "abcd":
    agg = collections.defaultdict(list)
    for f in glob.glob("gpurun_out/pmck2/%s/**/*counter_collection.csv" % name, recursive=True):
        for row in csv.DictReader(open(f)):
            if "traj_update_kernel" in row.get("Kernel_Name",""):
                agg[row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k,v in sorted(agg.items()):
        print(name, k, "n=%d mean=%.6g" % (len(v), sum(v)/len(v)))
PY
