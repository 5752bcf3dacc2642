#!/bin/bash
# PMC passes on the vector-memory path of the 32x32x16 ONF kernel (is the third-level fragment stream what bounds it?):
# TA / TD busy and the TCP stall reasons, summed over instances, per launch of onf_x32_kernel<14,0,...>, next to GRBM_GUI_ACTIVE.
set -o pipefail
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/pmc_x32_mem; mkdir -p $O
cd /tmp
run() { name=$1; shift
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $O/$name -- python3 $R/bench.py --steps 6 --warmup 2 --cpu-sample 0 --fit-iters 20 --spin-up 0 > $O/$name.log 2>&1 || { echo "pass $name failed"; tail -5 $O/$name.log; }
}
run m1 TA_TA_BUSY_sum TD_TD_BUSY_sum GRBM_GUI_ACTIVE
run m2 TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TD_TCP_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum
# (round 3 asked for these four in ONE pass: rocprofv3 aborted with "Request exceeds the capabilities of the hardware to collect";
# two counters per pass fit)
run m3a TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum
run m3b TA_DATA_STALLED_BY_TC_CYCLES_sum TD_TC_STALL_sum
run m4 TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum
cd $R
python3 - <<'PY'
import csv, glob, collections
for name in ("m1","m2","m3a","m3b","m4"):
    agg = collections.defaultdict(list)
    for f in glob.glob("gpurun_out/pmc_x32_mem/%s/**/*counter_collection.csv" % name, recursive=True):
        for row in csv.DictReader(open(f)):
            if "onf_x32_kernel<14, 0" in row.get("Kernel_Name",""):
                agg[row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, v in agg.items():
        print(name, k, "n=%d mean=%.6g" % (len(v), sum(v)/len(v)))
PY
