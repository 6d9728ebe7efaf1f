#!/bin/bash
# (query, segment) pairs per launch of the segment walk (RAG_GEMM_SEGMENTS_PAIRS): ms per batch at 10M x 768, k = 200
set -e
mkdir -p gpurun_out/seg_pairs
for pairs in ${PAIRS:-1024 2048 4096 8192}; do
  echo "pairs $pairs"
  RAG_GEMM_SEGMENTS_PAIRS=$pairs CASES=${CASES:-128:200,256:200,512:200,1024:200,1500:200} timeout -k 10 200 python3 -u scripts/bench_large_k.py > gpurun_out/seg_pairs/p$pairs.txt 2>&1
  grep "^1" gpurun_out/seg_pairs/p$pairs.txt
done
