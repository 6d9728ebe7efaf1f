set -e
for w in 512 768 1024; do export RAG_GEMM_WGS=$w
for nq in 32 64; do export NQ=$nq STEPS=5 WARMUP=2; echo -n "wgs=$w "; timeout -k 10 120 python3 scripts/tune_scan.py 2>&1 | grep kernel_ms | sed -e 's/blocks=.*path=/path=/'; done; done
