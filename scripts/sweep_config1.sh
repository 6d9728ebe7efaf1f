#!/bin/bash
# The 100k x 384 corpus under 4096 queries: kernel time of the selecting matrix-core kernel per library variant
# (build/variants/librag_hip_<name>.so, diagnostic builds: their answers are wrong) and per knob.
export ROWS=${ROWS:-100000} DIM=${DIM:-384} NQ=${NQ:-4096} STEPS=10 WARMUP=3
run() { echo -n "$1: "; env $1 timeout -k 10 100 python3 scripts/tune_scan.py 2>&1 | grep kernel_ms | sed -e 's/.*path=/path=/'; }
run X=0
for v in "$@"; do run RAG_HIP_LIB=$PWD/build/variants/librag_hip_$v.so; done
