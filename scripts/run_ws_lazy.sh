#!/bin/bash
# does the event a search records behind its last launch cost time on the device?  By default it is recorded only when a search on
# another stream needs it (the Python classes promise that their streams live on); RAG_WS_EAGER_EVENTS=1 records it after every search.  100k x 384 batch 1 (no profiling events) and the IVF share's batch 1.
set -e
set -o pipefail
small() { timeout -k 10 200 python3 - <<'PY'
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from bench import make_corpus, make_queries
from rag_amd import DeviceIndex
dev = torch.device("cuda", 0)
X = make_corpus(torch, 0, 100_000, 384, dev); Q = make_queries(torch, 64, 384, dev)
idx = DeviceIndex(384, metric="ip"); idx.adopt(X)
for rep in range(3):
    for i in range(200): idx.search(Q[i % 64:i % 64 + 1], 10)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(2000): idx.search(Q[i % 64:i % 64 + 1], 10)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"  100k x 384 batch 1: {dt / 2000 * 1e6:.2f} us per query", flush=True)
PY
}
ivf() { timeout -k 10 300 python3 scripts/bench_ivf_share.py 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print('  ivf share batch 1: %.1f us per query; batch 4096: %.3f ms' % (d['batch_1']['ms_per_step']*1e3, d['batch_4096']['ms_per_step']))"; }
for eager in 1 0 1 0; do
  echo "RAG_WS_EAGER_EVENTS=$eager"
  export RAG_WS_EAGER_EVENTS=$eager
  small
  ivf
done
