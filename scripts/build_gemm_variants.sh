#!/bin/bash
# usage: scripts/build_gemm_variants.sh NAME "-DFLAG=1 ..." [NAME "-D..."]...  -> build/variants/librag_hip_NAME.so
# (only gemm_topk.hip is rebuilt; the other objects come from rag_amd/csrc as built by the Makefile)
set -e
cd "$(dirname "$0")/.."
mkdir -p build/variants
while [ $# -ge 2 ]; do
  name=$1; flags=$2; shift 2
  (
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off $flags -c rag_amd/csrc/gemm_topk.hip -o build/variants/gemm_$name.o
    objs=$(ls rag_amd/csrc/*.o | grep -v gemm_topk.o)
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs build/variants/gemm_$name.o -o build/variants/librag_hip_$name.so
    echo built $name
  ) &
done
wait
