#!/bin/bash
# A/B of list-scan kernel variants (build/variants/librag_hip_<name>.so, see scripts/build_variant.sh) on the IVF
# leg of bench.py: prints the batch-4096 list-scan kernel time per variant.  usage: scripts/sweep_ivf_lists.sh name...
cd "$(dirname "$0")/.."
for v in "$@"; do
  RAG_HIP_LIB=$PWD/build/variants/librag_hip_$v.so timeout -k 10 200 python scripts/bench_ivf_share.py 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
b=d['batch_4096']
print('$v', 'scan_ms=%.3f' % b['list_scan_kernel_ms'], 'step_ms=%.3f' % b['ms_per_step'], 'batch1_us=%.1f' % (d['batch_1']['ms_per_step']*1e3), 'verified=', d.get('verified'))
" || exit 1
done
