#!/bin/bash
# the side measurements kept under profiles/: pipeline, fan-out, IVF, small corpora, short corpora under large batches, large k, L2, graph replay
set -e
mkdir -p gpurun_out/side
timeout -k 10 300 python3 -u scripts/bench_pipeline.py > gpurun_out/side/pipeline.json 2> gpurun_out/side/pipeline.err
timeout -k 10 300 python3 -u scripts/bench_fanout.py > gpurun_out/side/fanout.json 2> gpurun_out/side/fanout.err
timeout -k 10 400 python3 -u scripts/bench_ivf.py > gpurun_out/side/ivf.json 2> gpurun_out/side/ivf.err
timeout -k 10 300 python3 -u scripts/bench_small.py > gpurun_out/side/small.txt 2> gpurun_out/side/small.err
# large batches over short corpora, k above 64 in a batch, the L2 metric, graph replay (each step under its own time limit)
timeout -k 10 200 python3 -u scripts/bench_coarse.py > gpurun_out/side/coarse.txt 2> gpurun_out/side/coarse.err
timeout -k 10 300 python3 -u scripts/bench_large_k.py > gpurun_out/side/large_k.txt 2> gpurun_out/side/large_k.err
timeout -k 10 400 python3 -u scripts/bench_l2.py > gpurun_out/side/l2.txt 2> gpurun_out/side/l2.err
timeout -k 10 300 python3 -u scripts/bench_graph.py > gpurun_out/side/graph.txt 2> gpurun_out/side/graph.err
cat gpurun_out/side/coarse.txt gpurun_out/side/large_k.txt gpurun_out/side/l2.txt gpurun_out/side/graph.txt
cat gpurun_out/side/pipeline.json gpurun_out/side/fanout.json gpurun_out/side/ivf.json gpurun_out/side/small.txt
