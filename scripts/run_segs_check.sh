#!/bin/bash
# the segment walk (k above 64 in a batch): the GPU suite, then ms per batch at 10M x 768 and on small corpora with the walk on and off
set -e
mkdir -p gpurun_out/segs
timeout -k 10 900 python3 -u -m pytest tests -x -q -m gpu --timeout 120 --timeout-method thread > gpurun_out/segs/tests.txt 2>&1
tail -3 gpurun_out/segs/tests.txt
CASES=2:100,4:100,6:200,8:100,16:100,24:200,32:100,40:100,64:100,64:64,100:200,128:200,256:200,300:200,384:200,512:200,700:200,768:200,1024:200,1500:200,4096:200,256:1024 timeout -k 10 400 python3 -u scripts/bench_large_k.py > gpurun_out/segs/large_k.txt 2>&1
cat gpurun_out/segs/large_k.txt
# up to 32 queries with the score matrix written by the matrix-core GEMM shapes instead of the register-streaming kernel
RAG_NARROW_SCORES=0 CASES=2:100,8:100,16:100,32:100 timeout -k 10 200 python3 -u scripts/bench_large_k.py > gpurun_out/segs/narrow_off.txt 2>&1
cat gpurun_out/segs/narrow_off.txt
for shape in "100000 384" "1000000 768"; do
  set -- $shape
  ROWS=$1 DIM=$2 CASES=2:100,5:100,7:200,8:100,300:100 timeout -k 10 200 python3 -u scripts/bench_large_k.py > gpurun_out/segs/small_$1.txt 2>&1
  cat gpurun_out/segs/small_$1.txt
  ROWS=$1 DIM=$2 RAG_GEMM_SEGMENTS_MIN_NQ=0 CASES=2:100,5:100,7:200,8:100 timeout -k 10 200 python3 -u scripts/bench_large_k.py > gpurun_out/segs/small_$1_scan.txt 2>&1
  cat gpurun_out/segs/small_$1_scan.txt
done
