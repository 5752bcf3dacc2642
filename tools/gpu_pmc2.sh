#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
R=$PWD
mkdir -p $R/gpurun_out/pmc2
cd /tmp
run() { name=$1; shift
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $R/gpurun_out/pmc2/$name -- python3 $R/bench.py --steps 6 --warmup 2 --cpu-sample 0 --fit-iters 20 > $R/gpurun_out/pmc2/$name.log 2>&1 || { echo "pass $name failed"; tail -5 $R/gpurun_out/pmc2/$name.log; }
}
run a SQ_VALU_MFMA_COEXEC_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CU_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES
run b GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_VALU_TRANS_F32
cd $R
python3 - <<'PY'
import csv, glob, collections
for name in ("a","b"):
    agg = collections.defaultdict(list)
    for f in glob.glob("gpurun_out/pmc2/%s/**/*counter_collection.csv" % name, recursive=True):
        for row in csv.DictReader(open(f)):
            if "onf_fwd_bwd_kernel<14, 2, 0>" in row.get("Kernel_Name",""):
                agg[row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k,v in sorted(agg.items()):
        print(name, k, "n=%d mean=%.6g" % (len(v), sum(v)/len(v)))
PY
