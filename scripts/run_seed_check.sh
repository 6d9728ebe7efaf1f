#!/bin/bash
# k above 64 in the selecting kernel with seeded thresholds: parity cases, then timings with the seed on and off
set -e
set -o pipefail
OUT=gpurun_out/seed
mkdir -p $OUT
timeout -k 10 600 python -u -m pytest tests/test_gemm_gpu.py -x -q --timeout 300 --timeout-method thread -k "segments_path or seeded" > $OUT/tests.log 2>&1 || { tail -30 $OUT/tests.log; exit 1; }
tail -3 $OUT/tests.log
CASES=${CASES:-256:200,512:200,1024:200,1024:100,2048:200,4096:200}
CASES=$CASES timeout -k 10 300 python3 scripts/bench_large_k.py > $OUT/seed_on.txt 2>&1 || { tail $OUT/seed_on.txt; exit 1; }
cat $OUT/seed_on.txt
CASES=256:200,1024:200 RAG_GEMM_SEED=0 timeout -k 10 300 python3 scripts/bench_large_k.py > $OUT/seed_off.txt 2>&1 || { tail $OUT/seed_off.txt; exit 1; }
cat $OUT/seed_off.txt
CASES=256:200,1024:200 RAG_GEMM_SEED_SEGS=4 timeout -k 10 300 python3 scripts/bench_large_k.py > $OUT/seed_4.txt 2>&1 || { tail $OUT/seed_4.txt; exit 1; }
cat $OUT/seed_4.txt
CASES=256:200,1024:200 RAG_GEMM_SEED_SEGS=16 timeout -k 10 300 python3 scripts/bench_large_k.py > $OUT/seed_16.txt 2>&1 || { tail $OUT/seed_16.txt; exit 1; }
cat $OUT/seed_16.txt
