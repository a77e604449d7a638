# kernel statistics of the fit's Elkan path on the benchmark sample
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/r3_elkprof
SHEPSEG_FIT_TIMING=1 timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3_elkprof -- python3 tools/perf_fit.py 40000 6 > gpurun_out/r3_elkprof.log 2>&1
f=$(find gpurun_out/r3_elkprof -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    print('%-60s calls %6s  avg %9.1f us  total %8.1f ms' % (r['Name'][:60], r['Calls'], float(r['AverageNs']) / 1e3, float(r['TotalDurationNs']) / 1e6))
PY
grep "kmeans fit" gpurun_out/r3_elkprof.log | tail -2
rm -f gpurun_out/r3_elkprof/*/*kernel_trace.csv
timeout -k 10 300 python -m pytest tests/test_fit_elkan.py tests/test_gpu_tile.py -x -q -m gpu -k "fit or kmeans or elkan" 2>&1 | tail -2
