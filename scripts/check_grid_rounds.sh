#!/bin/bash
# batch sizes whose query-block count does not divide the workgroup target (3, 5, 6, 7 blocks): the score-writing launches of
# the segment walk and of the short-corpus path must stay within one round of workgroups
set -e
mkdir -p gpurun_out/grid_rounds
CASES=300:200,384:200,700:200,768:200,1500:200,1024:200 timeout -k 10 300 python3 -u scripts/bench_large_k.py > gpurun_out/grid_rounds/large_k.txt 2>&1
grep "^1" gpurun_out/grid_rounds/large_k.txt
SHAPES=16384:1024:512:64,16384:1024:700:64,16384:1024:384:64,16384:1024:1024:64,32768:768:900:10 timeout -k 10 200 python3 -u scripts/bench_coarse.py > gpurun_out/grid_rounds/coarse.txt 2>&1
cat gpurun_out/grid_rounds/coarse.txt
