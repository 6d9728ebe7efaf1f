#!/bin/bash
# the narrow kernel with one 2 KiB transposer image per wave (both row groups in turn) against two: rows of 768 floats (same
# shapes as before) and rows of 1024 floats (17..32 queries then fit LDS and leave the 256 x 32 GEMM shape)
set -e
set -o pipefail
export STEPS=20 WARMUP=3
one() { echo -n "$1 DIM=$DIM NQ=$NQ: "; if [ "$1" = base ]; then timeout -k 10 200 python3 scripts/tune_scan.py 2>&1 | grep kernel_ms | sed -e 's/.*path=/path=/'; else RAG_HIP_LIB=$PWD/build/variants/librag_hip_$1.so timeout -k 10 200 python3 scripts/tune_scan.py 2>&1 | grep kernel_ms | sed -e 's/.*path=/path=/'; fi; }
for dim in 768 1024; do for nq in 16 24 32; do export DIM=$dim NQ=$nq; one base; one tph; done; done
