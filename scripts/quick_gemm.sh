#!/bin/bash
# kernel time of the GEMM path at a few batch sizes, fp32 and with the bf16 first pass (10M x 768)
set -e
for nq in ${NQS:-32 128 512 4096}; do
  export NQ=$nq STEPS=$(( nq < 1024 ? 10 : 3 )) WARMUP=2
  PREFILTER=0 timeout -k 10 120 python3 scripts/tune_scan.py 2>&1 | grep kernel_ms | sed -e 's/blocks=.*path=/path=/'
done
for nq in ${NQS16:-256 4096}; do
  export NQ=$nq STEPS=5 WARMUP=2
  PREFILTER=1 timeout -k 10 120 python3 scripts/tune_scan.py 2>&1 | grep kernel_ms | sed -e 's/blocks=.*path=/path=/'
done
