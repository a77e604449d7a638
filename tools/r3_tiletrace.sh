# kernel trace of ONE tile run alone (no contention): per-kernel duration and the gaps between launches
R=$PWD; cd /tmp && export TMPDIR=/tmp; cd $R
rm -rf gpurun_out/tt
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tt -o run -- python tools/perf_tile.py 4096 > gpurun_out/tt.log 2>&1 &&
f=$(ls gpurun_out/tt/*kernel_trace.csv gpurun_out/tt/*/*kernel_trace.csv 2>/dev/null | head -1) &&
python - "$f" <<'PY'
import csv, sys
rows = []
with open(sys.argv[1]) as fh:
    for r in csv.DictReader(fh):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0][:48]))
rows.sort()
# the last doShepherdSegmentation call: from the last k_assign launch on
last = max(i for i, r in enumerate(rows) if r[2].startswith('void k_assign') or r[2].startswith('k_assign'))
sel = rows[last:]
t0 = sel[0][0]
with open('gpurun_out/r3_tiletrace.txt', 'w') as out:
    prev = None
    tot = 0
    for (s, e, n) in sel:
        gap = (s - prev) / 1e3 if prev else 0.0
        out.write('%9.1f  dur %8.1f  gap %7.1f  %s\n' % ((s - t0) / 1e3, (e - s) / 1e3, gap, n))
        prev = e
        tot += e - s
    out.write('launches %d  sum of durations %.1f us  span %.1f us\n' % (len(sel), tot / 1e3, (sel[-1][1] - t0) / 1e3))
PY
rm -rf gpurun_out/tt
tail -3 gpurun_out/r3_tiletrace.txt
