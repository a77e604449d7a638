"""Per-tile kernel time table from a rocprofv3 kernel_stats.csv: python tools/kstats.py DIR NTILES [TOP]"""
import csv, glob, sys
d, nt = sys.argv[1], int(sys.argv[2]); top = int(sys.argv[3]) if len(sys.argv) > 3 else 30
f = glob.glob(d + '/**/*kernel_stats.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = 0
for r in rows[:top]:
    n = int(r['Calls']); t = float(r['TotalDurationNs']) / 1e6
    print('%-44s calls %5d  per-tile %6.3f ms  avg %8.1f us' % (r['Name'][:44], n, t / nt, float(r['AverageNs']) / 1e3))
skip = ('k_dfs_split', 'k_small_loop', 'k_fit', 'k_synthimg')
print('sum per tile excluding dfs/small_loop/fit/synth: %.2f ms' % (sum(float(r['TotalDurationNs']) / 1e6 for r in rows if not r['Name'].startswith(skip) and 'k_fit' not in r['Name']) / nt))
