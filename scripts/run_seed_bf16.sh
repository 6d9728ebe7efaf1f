#!/bin/bash
# the bf16 first pass with seeded thresholds: parity, then step times with the seeding off and on
set -e
set -o pipefail
OUT=gpurun_out/seedbf16
mkdir -p $OUT
timeout -k 10 900 python -u -m pytest tests/test_prefilter_gpu.py tests/test_fullsize_gpu.py -x -q --timeout 300 --timeout-method thread -k "prefilter or bf16" > $OUT/tests.log 2>&1 || { tail -30 $OUT/tests.log; exit 1; }
tail -2 $OUT/tests.log
timeout -k 10 400 python3 -u tests/soak/soak_narrow_bf16.py > $OUT/soak.log 2>&1 || { tail -5 $OUT/soak.log; exit 1; }
tail -1 $OUT/soak.log
export STEPS=${STEPS:-10} WARMUP=3 PREFILTER=1 ROWS=10000000 DIM=768
run() { echo -n "bf16 first pass nq=$NQ k=${K:-10} $1: "; env $1 timeout -k 10 200 python3 scripts/tune_scan.py 2>&1 | grep kernel_ms | sed -e "s/.*path=/path=/" -e "s/GB.s=[0-9]* //" -e "s/launches.*//"; }
for nq in ${BATCHES:-8 32 64 100 256 512 1024 2048 4096}; do
  export NQ=$nq
  run RAG_GEMM_SEED_ROWS=0
  run RAG_GEMM_SEED_ROWS=-1
done
export NQ=4096 K=32
run RAG_GEMM_SEED_ROWS=0
run RAG_GEMM_SEED_ROWS=-1
