#!/bin/bash
# what the driver runs at the end of a round, in its order: the GPU tests, the smoke call, the default bench line
set -e
mkdir -p gpurun_out/final
timeout -k 10 900 python3 -u -m pytest tests -x -q -m gpu --timeout 120 --timeout-method thread > gpurun_out/final/tests.txt 2>&1
tail -3 gpurun_out/final/tests.txt
timeout -k 10 300 python3 -u -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/final/smoke.txt 2>&1
tail -2 gpurun_out/final/smoke.txt
timeout -k 10 500 python3 bench.py > gpurun_out/final/bench.json 2> gpurun_out/final/bench.err
tail -c 400 gpurun_out/final/bench.json
