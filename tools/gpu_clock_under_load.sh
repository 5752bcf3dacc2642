#!/bin/bash
# Is K1 power-limited?  The same launch (cfg3 shape, same instruction stream) on the fitted-like random weights and on all-zero
# weights: duration and GRBM_GUI_ACTIVE (busy cycles summed over the 8 XCDs) per launch -> the engine clock each one ran at.
export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out/clock; rm -rf $O; mkdir -p $O
cd /tmp
for z in 0 1; do
  if [ $z = 1 ]; then export AB_ZERO=1; else unset AB_ZERO; fi
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $O/z$z -- python3 $R/tools/x32/ab_multi.py > $O/z$z.log 2>&1
done
cd $R
python3 - <<'PY'
import csv, glob
for z in (0, 1):
    cyc, dur = [], {}
    for f in glob.glob("gpurun_out/clock/z%d/**/*kernel_trace.csv" % z, recursive=True):
        for r in csv.DictReader(open(f)):
            if "onf_x32_kernel<14, 0" in r["Kernel_Name"]:
                dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
    for f in glob.glob("gpurun_out/clock/z%d/**/*counter_collection.csv" % z, recursive=True):
        for r in csv.DictReader(open(f)):
            if "onf_x32_kernel<14, 0" in r["Kernel_Name"] and r["Counter_Name"] == "GRBM_GUI_ACTIVE" and r["Dispatch_Id"] in dur:
                cyc.append((float(r["Counter_Value"]) / 8, dur[r["Dispatch_Id"]]))
    cyc = cyc[len(cyc) // 2:]          # the later launches: clocks settled
    c = sum(a for a, _ in cyc) / len(cyc); t = sum(b for _, b in cyc) / len(cyc)
    print("%s weights: %d launches, %.4f ms, %.3f M busy cycles per XCD -> %.2f GHz" % ("zero" if z else "random", len(cyc), t * 1e3, c / 1e6, c / t / 1e9))
PY
