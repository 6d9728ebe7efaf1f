#!/bin/bash
# thresholds seeded from the first rows of a short corpus under a large batch: parity, then timings per seed size
set -e
set -o pipefail
OUT=gpurun_out/seed
mkdir -p $OUT
timeout -k 10 600 python -u -m pytest tests/test_gemm_gpu.py tests/test_fuzz_gpu.py -x -q --timeout 300 --timeout-method thread > $OUT/tests_short.log 2>&1 || { tail -30 $OUT/tests_short.log; exit 1; }
tail -3 $OUT/tests_short.log
export ROWS=100000 DIM=384 STEPS=20 WARMUP=3
run() { echo -n "$1 NQ=$NQ ROWS=$ROWS: "; env $1 timeout -k 10 100 python3 scripts/tune_scan.py 2>&1 | grep -E "ms" | tail -2 | tr '\n' ' '; echo; }
for nq in 4096 1024; do
  export NQ=$nq
  run RAG_GEMM_SEED_SHORT_ROWS=0
  run RAG_GEMM_SEED_SHORT_ROWS=1024
  run RAG_GEMM_SEED_SHORT_ROWS=2048
  run RAG_GEMM_SEED_SHORT_ROWS=4096
  run RAG_GEMM_SEED_SHORT_ROWS=8192
done
export ROWS=400000 NQ=4096
run RAG_GEMM_SEED_SHORT_ROWS=0
run RAG_GEMM_SEED_SHORT_ROWS=4096
