#!/bin/bash
# the bf16 first pass at 4096 queries under the placements of query blocks over the XCDs (RAG_GEMM_GQ): how many workgroups of an
# XCD share a row range decides how often its rows come from that XCD's L2
set -e
set -o pipefail
export STEPS=5 WARMUP=2 NQ=4096 PREFILTER=1
for gq in -1 8 4 2 1 0; do
  echo -n "RAG_GEMM_GQ=$gq: "; RAG_GEMM_GQ=$gq timeout -k 10 200 python3 scripts/tune_scan.py 2>&1 | grep -E "kernel_ms" | tail -1 | sed -e 's/.*path=/path=/'
done
