R=$PWD; cd /tmp && export TMPDIR=/tmp; cd $R
rm -rf gpurun_out/prof_w1
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_w1 -o run -- python bench.py --size 12288 --steps 2 --warmup 1 --cpu-sample 0 --workers 1 > gpurun_out/prof_w1.log 2>&1
python - <<'PY'
import csv, glob, collections
f = glob.glob('gpurun_out/prof_w1/**/*kernel_trace.csv', recursive=True)[0]
acc = collections.defaultdict(list)
rows = [r for r in csv.DictReader(open(f)) if r['Kernel_Name'].startswith('k_dfs_split')]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
for i, r in enumerate(rows):
    acc['first(B: 64K/global)' if i % 2 == 0 else 'second(A: 24K)'].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6)
for k, v in acc.items():
    print('k_dfs_split LDS', k, 'launches', len(v), 'avg ms %.2f' % (sum(v) / len(v)), 'max %.2f' % max(v), 'min %.2f' % min(v))
PY
rm -f gpurun_out/prof_w1/*kernel_trace.csv gpurun_out/prof_w1/*/*kernel_trace.csv
