#!/bin/bash
# does the seeding step pay at k <= 64 on longer corpora?
set -e
set -o pipefail
export DIM=768 STEPS=5 WARMUP=2 NQ=4096
run() { echo -n "$1 NQ=$NQ ROWS=$ROWS: "; env $1 timeout -k 10 200 python3 scripts/tune_scan.py 2>&1 | grep -E "ms" | tail -1; }
for rows in 1000000 10000000; do
  export ROWS=$rows
  run X=0
  run RAG_GEMM_SEED_SHORT_MAX_N=2000000000
  RAG_GEMM_SEED_SHORT_ROWS=16384 run RAG_GEMM_SEED_SHORT_MAX_N=2000000000
done
export NQ=1024 ROWS=10000000
run X=0
run RAG_GEMM_SEED_SHORT_MAX_N=2000000000
