#!/bin/bash
# starting thresholds of the IVF list scan: parity, then the 12.5M x 1024 share with the seed off / 32 / 64 rows
set -e
set -o pipefail
OUT=gpurun_out/ivfseed
mkdir -p $OUT
timeout -k 10 600 python -u -m pytest tests/test_ivf_gpu.py tests/test_ivf_config5_gpu.py tests/test_fuzz_gpu.py -x -q --timeout 300 --timeout-method thread > $OUT/tests.log 2>&1 || { tail -30 $OUT/tests.log; exit 1; }
tail -2 $OUT/tests.log
for r in ${SEEDS:-0 16 32}; do
  RAG_IVF_SEED_ROWS=$r timeout -k 10 300 python3 scripts/bench_ivf_share.py > $OUT/seed_$r.json 2> $OUT/seed_$r.err || { tail $OUT/seed_$r.err; exit 1; }
  python3 - $OUT/seed_$r.json $r <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
b = d["batch_4096"]
print("seed rows", sys.argv[2], "batch 4096:", round(b["ms_per_step"], 3), "ms/step, list scan", round(b["list_scan_kernel_ms"], 3), "ms, verified", d.get("verified"), "| batch 1", round(d["batch_1"]["ms_per_step"] * 1e3, 1), "us")
PY
done
