#!/bin/bash
# usage: scripts/build_variant.sh NAME "-DFLAG=1 ..."   -> build/variants/librag_hip_NAME.so
set -e
cd "$(dirname "$0")/.."
mkdir -p build/variants/$1
for f in rag_amd/csrc/*.hip; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off $2 -c $f -o build/variants/$1/$(basename $f .hip).o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC build/variants/$1/*.o -o build/variants/librag_hip_$1.so
echo built build/variants/librag_hip_$1.so
