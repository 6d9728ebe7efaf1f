set -e
for gq in 8 4 2 1; do export RAG_GEMM_GQ=$gq PREFILTER=1 NQ=4096 STEPS=3 WARMUP=2; echo -n "gq=$gq "; timeout -k 10 120 python3 scripts/tune_scan.py 2>&1 | grep kernel_ms | sed -e 's/blocks=.*path=/path=/'; done
