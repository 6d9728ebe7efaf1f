#!/bin/bash
# kernel statistics of the segment walk (CASES, default 256 queries at k = 200)
set -e
export TMPDIR=/tmp
export CASES=${CASES:-256:200}
OUT=gpurun_out/prof_large_k
mkdir -p $OUT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 scripts/bench_large_k.py > $OUT/run.txt 2> $OUT/run.err
f=$(ls $OUT/trace/*/*kernel_stats.csv | head -1)
(head -1 $f; grep "rag::" $f) | cut -c1-260 > $OUT/kernel_stats_rag.csv
cat $OUT/run.txt $OUT/kernel_stats_rag.csv
