#!/bin/bash
# Round evidence: the bench line, the rocprofv3 kernel-trace summary of the same command, the PMC passes.
set -e
set -o pipefail
export TMPDIR=/tmp
OUT=gpurun_out/prof
mkdir -p $OUT
timeout -k 10 400 python3 bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err
tail -c 600 $OUT/bench.json; echo
# the same command under the kernel trace
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --steps 20 --warmup 5 > $OUT/bench_traced.json 2> $OUT/trace.err
f=$(ls $OUT/trace/*/*kernel_stats.csv | head -1)
(head -1 $f; grep "rag::" $f) > $OUT/kernel_stats_rag.csv
cat $OUT/kernel_stats_rag.csv
bash scripts/run_pmc.sh > $OUT/pmc.log 2>&1 || { tail -20 $OUT/pmc.log; exit 1; }
grep -E "rag::" gpurun_out/pmc/summary_b4096.csv gpurun_out/pmc/summary_b1.csv | cut -c1-200
