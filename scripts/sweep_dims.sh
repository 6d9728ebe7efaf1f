#!/bin/bash
# scan and GEMM kernel rates over the embedding sizes the reference's models produce
set -e
for cfg in "20000000 384" "12000000 512" "10000000 768" "7500000 1024" "5000000 1536"; do
  set -- $cfg
  for nq in 1 4096; do
    export ROWS=$1 DIM=$2 NQ=$nq STEPS=$(( nq == 1 ? 20 : 2 )) WARMUP=2
    timeout -k 10 200 python3 scripts/tune_scan.py 2>&1 | grep kernel_ms | sed -e 's/blocks=.*path=/path=/'
  done
done
