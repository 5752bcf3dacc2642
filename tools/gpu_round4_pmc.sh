#!/bin/bash
# Round-4 evidence run, part 2 (GPU box): PMC passes (separate rocprofv3 --pmc runs, --kernel-trace only) for K1 and for both passes
# of the ONF fit, and the vector-memory counters of K1 with the pass rocprofv3 refused in round 3 split in two.
set -o pipefail
R=$PWD; O=$R/gpurun_out/r04; mkdir -p $O
bash tools/gpu_pmc_x32.sh > $O/pmc_x32.log 2>&1; tail -30 $O/pmc_x32.log
timeout -k 10 600 bash tools/gpu_pmc_wgrad.sh > $O/pmc_wgrad.log 2>&1; tail -60 $O/pmc_wgrad.log
bash tools/gpu_pmc_x32_mem.sh > $O/pmc_x32_mem.log 2>&1; tail -16 $O/pmc_x32_mem.log
