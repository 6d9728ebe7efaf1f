#!/bin/bash
# A/B of two builds of the library on the same box: usage ab_gemm.sh "NQ list" libA.so libB.so
set -e
for nq in $1; do
  for lib in $2 $3 $2 $3; do
    export NQ=$nq STEPS=3 WARMUP=1 RAG_HIP_LIB=$PWD/$lib
    echo -n "$lib  "; timeout -k 10 120 python3 scripts/tune_scan.py 2>&1 | grep kernel_ms | sed -e 's/blocks=.*path=/path=/'
  done
done
