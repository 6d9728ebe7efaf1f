set -e
for mt in 16 8 4 2; do export RAG_SCAN_MIN_TILES=$mt; echo "min_tiles=$mt"; timeout -k 10 200 python3 scripts/bench_small.py 2>&1 | grep -v amdgpu | head -2; done
