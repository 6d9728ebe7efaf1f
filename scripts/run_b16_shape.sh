#!/bin/bash
# timing-only: the bf16 first pass without its epilogue, 32x32x16 MFMAs (g1) against 16x16x32 MFMAs on the same fragments (g2049)
set -e
set -o pipefail
export STEPS=5 WARMUP=2 NQ=4096 PREFILTER=1
for rep in 1 2; do
  for v in g1 g2049; do
    echo -n "$v: "; RAG_HIP_LIB=$PWD/build/variants/librag_hip_$v.so timeout -k 10 200 python3 scripts/tune_scan.py 2>&1 | grep -E "kernel_ms" | tail -1 | sed -e 's/.*path=/path=/'
  done
done
