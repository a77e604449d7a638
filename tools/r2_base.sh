# round-2 baseline: single-tile stage timings with the pass-loop phase breakdown, then a kernel trace of one default step
R=$PWD; cd /tmp && export TMPDIR=/tmp; cd $R
SHEPSEG_SMALL_TIMING=1 timeout -k 10 300 python tools/perf_tile.py 4096 > gpurun_out/r2_tile.log 2>&1; tail -12 gpurun_out/r2_tile.log
rm -rf gpurun_out/tl
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl -o run -- python bench.py --steps 1 --warmup 1 --cpu-sample 0 > gpurun_out/tl.log 2>&1 &&
f=$(ls gpurun_out/tl/*kernel_trace.csv gpurun_out/tl/*/*kernel_trace.csv 2>/dev/null | head -1) &&
python tools/timeline.py $f > gpurun_out/timeline.txt 2>&1 &&
python - "$f" <<'PY'
import csv, sys
# compact copy of the trace: start, end, short name, queue, workgroups
with open(sys.argv[1]) as fh, open('gpurun_out/r2_trace_compact.csv', 'w') as out:
    for r in csv.DictReader(fh):
        g = int(r.get('Grid_Size_X', 0) or 0) * max(int(r.get('Grid_Size_Y', 1) or 1), 1)
        w = max(int(r.get('Workgroup_Size_X', 1) or 1) * max(int(r.get('Workgroup_Size_Y', 1) or 1), 1), 1)
        out.write('%s,%s,%s,%s,%d,%s\n' % (r['Start_Timestamp'], r['End_Timestamp'], r['Kernel_Name'].split('(')[0][:40], r.get('Queue_Id', '0'), g // w, r.get('LDS_Block_Size', r.get('LDS_Block_Size_v', ''))))
PY
rm -rf gpurun_out/tl
tail -1 gpurun_out/tl.log | cut -c1-300
