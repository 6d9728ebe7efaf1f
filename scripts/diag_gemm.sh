#!/bin/bash
# timing-only ablations of the GEMM kernel (scripts/build_variant.sh diagN "-DRAG_GEMM_DIAG=N")
set -e
for nq in ${NQS:-4096}; do
for v in "" ${VARIANTS:-diag1 diag2}; do
  export NQ=$nq STEPS=3 WARMUP=1
  if [ -n "$v" ]; then export RAG_HIP_LIB=$PWD/build/variants/librag_hip_$v.so; else unset RAG_HIP_LIB; fi
  echo -n "variant ${v:-base} "; timeout -k 10 120 python3 scripts/tune_scan.py 2>&1 | grep kernel_ms | sed -e 's/blocks=.*path=/path=/'
done
done
