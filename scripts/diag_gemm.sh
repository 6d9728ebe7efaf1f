set -e
export NQ=4096 STEPS=3 WARMUP=1
for v in "" diag1 diag2 diag4 diag6 diag7; do
  if [ -n "$v" ]; then export RAG_HIP_LIB=$PWD/build/variants/librag_hip_$v.so; fi
  echo "variant ${v:-base}"; timeout -k 10 120 python3 scripts/tune_scan.py 2>&1 | grep kernel_ms
done
