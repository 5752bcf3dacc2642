#!/bin/bash
# PMC passes for both passes of the ONF fit (pass 1 = onf_x32_kernel<.., 1, ..>, pass 2 = onf_wgrad_split_kernel) at the cfg5
# size (P = 2 543 616 only: NFOPP_DEV_LIB makes tools/train_speed.py skip the smaller sizes); separate runs per group.
set -o pipefail
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/pmc_wgrad; mkdir -p $O
export NFOPP_DEV_LIB=${NFOPP_DEV_LIB:-$R/pytorch-motion-planner_amd/nfopp/lib/libnfopp_hip.so}
cd /tmp
run() {  # name, counters...
  name=$1; shift
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $O/$name -- python3 $R/tools/train_speed.py > $O/$name.log 2>&1 || { echo "pass $name failed"; tail -5 $O/$name.log; }
}
run fetch FETCH_SIZE
run write WRITE_SIZE
run sq1 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
run sq2 SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE
run sq3 SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_LDS
cd $R
python3 - <<'PY'
import csv, glob, collections, json
for kern, fname in (("onf_wgrad_split_kernel", "summary_onf_wgrad_split_kernel.json"), ("onf_x32_kernel<14, 1", "summary_onf_x32_train_kernel.json")):
    out = {}
    for name in ("fetch", "write", "sq1", "sq2", "sq3"):
        files = glob.glob("gpurun_out/pmc_wgrad/%s/**/*counter_collection.csv" % name, recursive=True)
        agg = collections.defaultdict(list)
        for f in files:
            for row in csv.DictReader(open(f)):
                if kern in row.get("Kernel_Name", ""):
                    agg[row["Counter_Name"]].append(float(row["Counter_Value"]))
        for k, v in agg.items():
            out[k] = sum(v) / len(v)
            print(kern, name, k, "n=%d mean=%.6g" % (len(v), out[k]))
    json.dump(out, open("gpurun_out/pmc_wgrad/" + fname, "w"), indent=1, sort_keys=True)
PY
