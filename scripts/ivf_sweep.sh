set -o pipefail
for cfg in "32 2048 0" "16 2048 0" "8 2048 0" "8 4096 0" "4 4096 0" "32 2048 8" "8 2048 8" "8 2048 4"; do
  set -- $cfg
  echo "slice_tiles=$1 blocks=$2 min_tiles=$3"
  RAG_IVF_SLICE_TILES=$1 RAG_IVF_SCAN_BLOCKS=$2 RAG_SCAN_MIN_TILES=$3 ROWS=12500000 timeout -k 10 120 python3 scripts/timeline_ivf.py || exit 1
done
