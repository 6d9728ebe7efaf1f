#!/bin/bash
# k <= 64 batches of the selecting matrix-core kernel with seeded thresholds: parity, then step times with the seeding
# off (RAG_GEMM_SEED_ROWS=0), by the rule, and with forced row counts
set -e
set -o pipefail
OUT=gpurun_out/seedk64
mkdir -p $OUT
timeout -k 10 900 python -u -m pytest tests/test_gemm_gpu.py tests/test_fuzz_gpu.py tests/test_flat_gpu.py tests/test_facade_gpu.py -x -q --timeout 300 --timeout-method thread > $OUT/tests.log 2>&1 || { tail -30 $OUT/tests.log; exit 1; }
tail -2 $OUT/tests.log
export STEPS=${STEPS:-10} WARMUP=3
run() { echo -n "$ROWS x $DIM nq=$NQ k=${K:-10} $1: "; env $1 timeout -k 10 200 python3 scripts/tune_scan.py 2>&1 | grep kernel_ms | sed -e "s/.*path=/path=/" -e "s/GB.s=[0-9]* //" -e "s/launches.*//"; }
for cfg in "100000 384" "1000000 768" "10000000 768"; do
  set -- $cfg; export ROWS=$1 DIM=$2
  for nq in 40 64 128 256 1024 4096; do
    export NQ=$nq
    run RAG_GEMM_SEED_ROWS=0
    run RAG_GEMM_SEED_ROWS=-1
  done
done
export ROWS=10000000 DIM=768 NQ=256 K=64
run RAG_GEMM_SEED_ROWS=0
run RAG_GEMM_SEED_ROWS=-1
