#!/bin/bash
# PMC passes for the fused ONF kernel (separate runs per counter group, as the microarch guide prescribes).
set -o pipefail
export TMPDIR=/tmp
R=$PWD
mkdir -p $R/gpurun_out/pmc
cd /tmp
rocprofv3 -L > $R/gpurun_out/pmc/counters_list.txt 2>&1 || true
run() {  # name, counters...
  name=$1; shift
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $R/gpurun_out/pmc/$name -- python3 $R/bench.py --matrix-path fp32 --steps 6 --warmup 2 --cpu-sample 0 --fit-iters 20 --spin-up 0 > $R/gpurun_out/pmc/$name.log 2>&1 || { echo "pass $name failed"; tail -5 $R/gpurun_out/pmc/$name.log; }
}
run fetch FETCH_SIZE
run write WRITE_SIZE
run sq1 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
run sq2 SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE
cd $R
python3 - <<'PY'
import csv, glob, collections
for name in ("fetch","write","sq1","sq2"):
    files = glob.glob("gpurun_out/pmc/%s/**/*counter_collection.csv" % name, recursive=True)
    agg = collections.defaultdict(list)
    for f in files:
        for row in csv.DictReader(open(f)):
            if "onf_fwd_bwd_kernel<14, 2, 0>" in row.get("Kernel_Name",""):
                agg[row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k,v in agg.items():
        print(name, k, "n=%d mean=%.6g" % (len(v), sum(v)/len(v)))
PY
