cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for v in "$@"; do
  echo "== experiment $v"
  rm -rf gpurun_out/elkprof
  SHEPSEG_LIBPATH=$GRAFT_REPO_ROOT/tools/ab/libexp$v.so timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/elkprof -- python3 tools/perf_fit.py 40000 6 > gpurun_out/elkprof.log 2>&1
  f=$(find gpurun_out/elkprof -name "*kernel_stats.csv" | head -1)
  python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'sum_lists' in r['Name'] or 'estep' in r['Name']: print(r['Name'][:30], r['Calls'], float(r['AverageNs'])/1e3)
PY
  grep "kmeans fit" gpurun_out/elkprof.log | tail -1
done
