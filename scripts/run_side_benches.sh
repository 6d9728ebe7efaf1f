#!/bin/bash
# the side measurements kept under profiles/: pipeline, fan-out, IVF, small corpora
set -e
mkdir -p gpurun_out/side
timeout -k 10 300 python3 -u scripts/bench_pipeline.py > gpurun_out/side/pipeline.json 2> gpurun_out/side/pipeline.err
timeout -k 10 300 python3 -u scripts/bench_fanout.py > gpurun_out/side/fanout.json 2> gpurun_out/side/fanout.err
timeout -k 10 400 python3 -u scripts/bench_ivf.py > gpurun_out/side/ivf.json 2> gpurun_out/side/ivf.err
timeout -k 10 300 python3 -u scripts/bench_small.py > gpurun_out/side/small.txt 2> gpurun_out/side/small.err
cat gpurun_out/side/pipeline.json gpurun_out/side/fanout.json gpurun_out/side/ivf.json gpurun_out/side/small.txt
