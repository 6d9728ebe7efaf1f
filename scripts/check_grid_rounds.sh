#!/bin/bash
# batch sizes whose query-block count is not a power of two: ms per batch at 10M x 768 (k = 10: the selecting kernel; k = 200:
# the segment walk below 2048 queries) and the short-corpus path
set -e
mkdir -p gpurun_out/grid_rounds
CASES=${CASES:-1280:10,2048:10,2304:10,3000:10,3500:10,4096:10,4352:10,700:10,300:10,300:200,768:200,1500:200,2304:200,3000:200} timeout -k 10 400 python3 -u scripts/bench_large_k.py > gpurun_out/grid_rounds/large_k.txt 2>&1
grep "^1" gpurun_out/grid_rounds/large_k.txt
SHAPES=16384:1024:512:64,16384:1024:700:64,16384:1024:384:64,16384:1024:1024:64,16384:1024:2304:64,16384:1024:4096:64,32768:768:900:10 timeout -k 10 200 python3 -u scripts/bench_coarse.py > gpurun_out/grid_rounds/coarse.txt 2>&1
cat gpurun_out/grid_rounds/coarse.txt
