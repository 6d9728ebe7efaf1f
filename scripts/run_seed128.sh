#!/bin/bash
# k 65..200 in the 128 x 128 shape with seeded thresholds: parity, then timings against the segment walk
set -e
set -o pipefail
OUT=gpurun_out/seed128
mkdir -p $OUT
timeout -k 10 900 python -u -m pytest tests/test_gemm_gpu.py tests/test_fuzz_gpu.py -x -q --timeout 300 --timeout-method thread > $OUT/tests.log 2>&1 || { tail -30 $OUT/tests.log; exit 1; }
tail -2 $OUT/tests.log
CASES=${CASES:-65:100,100:100,128:200,300:200,600:200,900:150} timeout -k 10 300 python3 scripts/bench_large_k.py 2>&1 | grep -v amdgpu.ids | tee $OUT/seeded.txt
CASES=${CASES:-65:100,100:100,128:200,300:200,600:200,900:150} RAG_GEMM_SEED=0 timeout -k 10 300 python3 scripts/bench_large_k.py 2>&1 | grep -v amdgpu.ids | tee $OUT/walk.txt
CASES=100:10,300:10,600:64 timeout -k 10 300 python3 scripts/bench_large_k.py 2>&1 | grep -v amdgpu.ids | tee $OUT/k10.txt
