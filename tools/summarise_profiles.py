"""Condense the rocprofv3 outputs of tools/refresh_profiles.sh into the small files kept under
profiles/: the per-kernel stats table, the bench JSON line of the profiled run, and per-kernel
FETCH_SIZE / WRITE_SIZE averages (KB per launch)."""
import csv, glob, json, os, sys, collections
tag = sys.argv[1]
out = 'gpurun_out/summary_%s' % tag
os.makedirs(out, exist_ok=True)

def one(pattern):
    g = glob.glob(pattern, recursive=True)
    return g[0] if g else None

st = one('gpurun_out/prof_%s_stats/**/*kernel_stats.csv' % tag)
if st:
    open(os.path.join(out, 'kernel_stats_c3_default.csv'), 'w').write(open(st).read())
for wl in ('c4', 'c5'):
    st2 = one('gpurun_out/prof_%s_%s/**/*kernel_stats.csv' % (tag, wl))
    if st2:
        open(os.path.join(out, 'kernel_stats_%s.csv' % wl), 'w').write(open(st2).read())
    lg = 'gpurun_out/prof_%s_%s.log' % (tag, wl)
    if os.path.exists(lg):
        for ln in open(lg):
            if ln.startswith('{"metric"'):
                open(os.path.join(out, 'bench_line_%s_under_rocprof.json' % wl), 'w').write(ln)
log = 'gpurun_out/prof_%s_stats.log' % tag
if os.path.exists(log):
    for ln in open(log):
        if ln.startswith('{"metric"'):
            open(os.path.join(out, 'bench_line_under_rocprof.json'), 'w').write(ln)

def short(name):
    n = name.split('(')[0]
    if n.startswith('void '): n = n[5:]
    return n.split('<')[0]

agg = {}
for key, d in (('FETCH_SIZE', 'pmcf'), ('WRITE_SIZE', 'pmcw')):
    f = one('gpurun_out/prof_%s_%s/**/*counter_collection.csv' % (tag, d))
    if not f: continue
    acc = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        if r['Counter_Name'] != key: continue
        a = acc[short(r['Kernel_Name'])]
        a[0] += 1; a[1] += float(r['Counter_Value'])
    for k, (n, v) in acc.items():
        e = agg.setdefault(k, {'launches': n})
        e[key + '_KB_per_launch'] = round(v / n, 1)
if agg:
    rows = sorted(agg.items(), key=lambda kv: -(kv[1].get('FETCH_SIZE_KB_per_launch', 0) + kv[1].get('WRITE_SIZE_KB_per_launch', 0)) * kv[1]['launches'])
    with open(os.path.join(out, 'pmc_fetch_write_per_kernel.csv'), 'w') as fh:
        fh.write('kernel,launches,FETCH_SIZE_KB_per_launch,WRITE_SIZE_KB_per_launch\n')
        for k, e in rows:
            fh.write('%s,%d,%s,%s\n' % (k, e['launches'], e.get('FETCH_SIZE_KB_per_launch', ''), e.get('WRITE_SIZE_KB_per_launch', '')))
    json.dump({"note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only) of "
               "`python bench.py --steps 1 --warmup 1 --cpu-sample 0`; raw counter values in KB per launch, "
               "averaged over all launches. gfx950: FETCH_SIZE counts 1/2 of wide coalesced reads (k_window: "
               "~107 MB raw vs 219 MB read), exact for 16-B streaming writes; uncalibrated for the scattered "
               "4-byte accesses of k_small_loop / k_dfs_pool.",
               "kernels": dict(rows)}, open(os.path.join(out, 'pmc_summary.json'), 'w'), indent=1)
print('summaries in', out, os.listdir(out))
