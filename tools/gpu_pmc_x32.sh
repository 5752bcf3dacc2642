#!/bin/bash
# PMC passes for the 32x32x16 ONF kernel (csrc/onf_x32.hip); separate runs per counter group.
set -o pipefail
export TMPDIR=/tmp
unset NFOPP_MATRIX_PATH
R=$PWD
mkdir -p $R/gpurun_out/pmc_x32
cd /tmp
run() {  # name, counters...
  name=$1; shift
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $R/gpurun_out/pmc_x32/$name -- python3 $R/bench.py --steps 6 --warmup 2 --cpu-sample 0 --fit-iters 20 --spin-up 0 > $R/gpurun_out/pmc_x32/$name.log 2>&1 || { echo "pass $name failed"; tail -5 $R/gpurun_out/pmc_x32/$name.log; }
}
run fetch FETCH_SIZE
run write WRITE_SIZE
run sq1 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
run sq2 SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE
run sq3 SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_SALU
cd $R
python3 - <<'PY'
import csv, glob, collections, json
out = {}
for name in ("fetch","write","sq1","sq2","sq3"):
    files = glob.glob("gpurun_out/pmc_x32/%s/**/*counter_collection.csv" % name, recursive=True)
    agg = collections.defaultdict(list)
    for f in files:
        for row in csv.DictReader(open(f)):
            if "onf_x32_kernel<14, 0" in row.get("Kernel_Name",""):
                agg[row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k,v in agg.items():
        out[k] = sum(v)/len(v)
        print(name, k, "n=%d mean=%.6g" % (len(v), out[k]))
json.dump(out, open("gpurun_out/pmc_x32/summary_onf_x32_kernel.json","w"), indent=1, sort_keys=True)
PY
