#!/bin/bash
# A/B on one box: the narrow kernel's previous start (a selecting launch over the leading 32768 rows) against the score launch + bound
set -e
set -o pipefail
export STEPS=30 WARMUP=5
run() { echo -n "$1 NQ=$NQ: "; env $1 timeout -k 10 200 python3 scripts/tune_scan.py 2>&1 | grep -E "ms" | tail -1 | sed -e 's/.*path=/path=/'; }
for rep in 1 2; do
for nq in 16 32 8 24; do
  export NQ=$nq
  run RAG_HIP_LIB=$PWD/build/variants/librag_hip_oldnarrow.so
  run RAG_NARROW_SEED_ROWS=8192
  run RAG_NARROW_SEED_ROWS=32768
done
done
