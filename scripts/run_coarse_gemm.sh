export ROWS=16384 DIM=1024 NQ=4096 K=64 STEPS=20 WARMUP=3
echo -n "base: "; timeout -k 10 100 python3 scripts/tune_scan.py 2>&1 | grep kernel_ms | sed -e 's/.*path=/path=/'
for v in g1 g4; do echo -n "$v: "; RAG_HIP_LIB=$PWD/build/variants/librag_hip_$v.so timeout -k 10 100 python3 scripts/tune_scan.py 2>&1 | grep kernel_ms | sed -e 's/.*path=/path=/'; done
echo -n "base: "; timeout -k 10 100 python3 scripts/tune_scan.py 2>&1 | grep kernel_ms | sed -e 's/.*path=/path=/'
