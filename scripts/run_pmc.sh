#!/bin/bash
# Counter passes for the two dominant kernels (10M x 768): one rocprofv3 --pmc run per
# counter group, nothing but counters in a run.  Results under gpurun_out/pmc/.
set -e
set -o pipefail
export TMPDIR=/tmp
OUT=gpurun_out/pmc
mkdir -p $OUT
run() {   # name, NQ, STEPS, counters...
    local name=$1 nq=$2 steps=$3; shift 3
    export NQ=$nq STEPS=$steps WARMUP=1
    timeout -s KILL 170 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -- python3 scripts/tune_scan.py > $OUT/$name.log 2>&1
    echo "pass $name done"
}
run b4096_sq 4096 2 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE
run b4096_fetch 4096 2 FETCH_SIZE
run b4096_write 4096 2 WRITE_SIZE
run b1_fetch 1 10 FETCH_SIZE
run b1_write 1 10 WRITE_SIZE
python3 scripts/pmc_summary.py "batch 4096" $OUT/b4096_sq $OUT/b4096_fetch $OUT/b4096_write > $OUT/summary_b4096.csv
python3 scripts/pmc_summary.py "batch 1" $OUT/b1_fetch $OUT/b1_write > $OUT/summary_b1.csv
cat $OUT/summary_b4096.csv $OUT/summary_b1.csv
