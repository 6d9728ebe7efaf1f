#!/bin/bash
# queries/sec over batch size at 10M x 768 (one process per batch size)
set -e
for nq in 1 2 3 4 8 12 16 32 64 100 128 256 512 1024; do
  export NQ=$nq STEPS=$(( nq < 16 ? 20 : 5 )) WARMUP=2
  timeout -k 10 120 python3 scripts/tune_scan.py 2>&1 | grep kernel_ms | sed -e 's/blocks=.*path=/path=/'
done
