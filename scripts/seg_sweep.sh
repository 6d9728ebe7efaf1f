cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export NDOCS=100 LIMIT=10
for st in 32 16 8 4; do
  export RAG_SEG_SLICE_TILES=$st
  timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/seg_$st -- python3 scripts/fanout_trace.py > gpurun_out/seg_$st.log 2>&1 || exit 1
  f=$(ls gpurun_out/seg_$st/*/*kernel_stats.csv | head -n 1)
  echo "SLICE_TILES=$st"; python3 - "$f" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'rag::' in r['Name'] and 'ingest' not in r['Name']: print(f"  {r['Name'][:50]:50s} {float(r['AverageNs'])/1e3:8.2f}")
PY
done
