#!/bin/bash
# Counter passes for the IVF leg of bench.py (12.5M x 1024, nlist 16384, nprobe 64): HBM traffic and MFMA activity of the
# list-scan kernels, one rocprofv3 --pmc run per counter group.  Results under gpurun_out/pmc_ivf/.
set -e
set -o pipefail
export TMPDIR=/tmp
OUT=gpurun_out/pmc_ivf
mkdir -p $OUT
run() {   # name, counters...
    local name=$1; shift
    timeout -s KILL 170 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -- python3 scripts/bench_ivf_share.py > $OUT/$name.log 2>&1
    echo "pass $name done"
}
run fetch FETCH_SIZE
run write WRITE_SIZE
run sq SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE
python3 scripts/pmc_summary.py "ivf share" $OUT/fetch $OUT/write $OUT/sq > $OUT/summary.csv
grep -E "ivf_lists_kernel|flat_scan_kernel" $OUT/summary.csv | cut -c1-220
# and the kernel trace of the same program
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 scripts/bench_ivf_share.py > $OUT/traced.json 2> $OUT/trace.err
f=$(ls $OUT/trace/*/*kernel_stats.csv | head -1)
(head -1 $f; grep "rag::" $f) > $OUT/kernel_stats_rag.csv
head -12 $OUT/kernel_stats_rag.csv | cut -c1-200
