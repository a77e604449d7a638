# diagnostic: per-kernel median durations inside one bench step, with and without the walkers
R=$PWD; cd /tmp && export TMPDIR=/tmp; cd $R
i=0
for spec in ${LAT_SPECS:-"A=0" "SHEPSEG_DBG_SKIP_DFS=1,SHEPSEG_DBG_SKIP_SMALL=1" "SHEPSEG_FILL_MAX=1" "SHEPSEG_FILL_MAX=2"}; do
  i=$((i+1))
  for kv in $(echo "$spec" | tr ',' ' '); do export $kv; done
  rm -rf gpurun_out/tl
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl -o run -- python bench.py --steps 1 --warmup 1 --cpu-sample 0 > gpurun_out/tl.log 2>&1 || { tail -5 gpurun_out/tl.log; exit 1; }
  for kv in $(echo "$spec" | tr ',' ' '); do unset ${kv%%=*}; done
  f=$(ls gpurun_out/tl/*kernel_trace.csv gpurun_out/tl/*/*kernel_trace.csv 2>/dev/null | head -1)
  echo "== $spec" >> gpurun_out/r2_lat.txt
  grep "^{" gpurun_out/tl.log | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['config']['step_s'])" >> gpurun_out/r2_lat.txt
  python - "$f" >> gpurun_out/r2_lat.txt <<'PY'
import csv, sys, collections
d = collections.defaultdict(list)
with open(sys.argv[1]) as fh:
    for r in csv.DictReader(fh):
        d[r['Kernel_Name'].split('(')[0][:40]].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    v.sort()
    print('%-42s n %5d sum %8.1f ms  med %7.1f  p10 %7.1f p90 %8.1f' % (k, len(v), sum(v) / 1e3, v[len(v) // 2], v[len(v) // 10], v[int(len(v) * .9)]))
PY
done
rm -rf gpurun_out/tl
grep -A1 "^==" gpurun_out/r2_lat.txt
