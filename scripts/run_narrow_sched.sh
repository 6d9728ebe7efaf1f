#!/bin/bash
# A/B on one box: issue-order variants and timing-only ablations of narrow_topk_kernel (built by scripts/build_narrow_variants.sh)
set -e
set -o pipefail
export STEPS=30 WARMUP=5
run() { echo -n "$1 NQ=$NQ: "; RAG_HIP_LIB=$PWD/build/variants/librag_hip_$1.so timeout -k 10 200 python3 scripts/tune_scan.py 2>&1 | grep -E "kernel_ms" | tail -1 | sed -e 's/.*path=/path=/'; }
base() { echo -n "base NQ=$NQ: "; timeout -k 10 200 python3 scripts/tune_scan.py 2>&1 | grep -E "kernel_ms" | tail -1 | sed -e 's/.*path=/path=/'; }
for nq in ${NQS:-32 16}; do
  export NQ=$nq
  base
  for v in ${VARIANTS}; do run $v; done
  base
done
