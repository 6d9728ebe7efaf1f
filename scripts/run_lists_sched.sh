#!/bin/bash
# A/B on one box: issue-order variants of ivf_lists_kernel (built by scripts/build_lists_variants.sh) on the IVF leg of the bench
set -e
set -o pipefail
pick() { python3 -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1])
b=d['batch_4096']; o=d['batch_1']
print('batch 4096: step %.3f ms, list scan %.3f ms; batch 1: %.1f us, scan %.1f us; verified %s' % (b['ms_per_step'], b['list_scan_kernel_ms'], o['ms_per_step']*1e3, o['list_scan_kernel_ms']*1e3, d.get('verified')))"; }
base() { echo -n "base: "; timeout -k 10 300 python3 scripts/bench_ivf_share.py 2>/dev/null | pick; }
run() { echo -n "$1: "; RAG_HIP_LIB=$PWD/build/variants/librag_hip_$1.so timeout -k 10 300 python3 scripts/bench_ivf_share.py 2>/dev/null | pick; }
base
for v in ${VARIANTS}; do run $v; done
base
