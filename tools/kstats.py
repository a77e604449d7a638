"""rocprofv3 CSV digests.
    python tools/kstats.py DIR NTILES [TOP]        per-tile kernel time table from DIR/**/kernel_stats.csv
    python tools/kstats.py --top N FILE            the N largest rows of a kernel_stats.csv
    python tools/kstats.py --sequence TRACE.csv    kernel sequence (start, duration, gap) of the LAST tile call in a
                                                   kernel trace of tools/perf_tile.py (one tile run alone)
    python tools/kstats.py --calls NAME TRACE.csv [EVERY]   durations (us) of a kernel's calls in launch order, EVERY-th printed"""
import csv
import glob
import sys


def short(name):
    n = name.split('(')[0]
    return n[5:] if n.startswith('void ') else n


def main():
    if sys.argv[1] == '--top':
        rows = list(csv.DictReader(open(sys.argv[3])))
        for r in rows[:int(sys.argv[2])]:
            print('%-48s calls %6s  avg %9.1f us  total %8.1f ms' % (short(r['Name'])[:48], r['Calls'],
                  float(r['AverageNs']) / 1e3, float(r['TotalDurationNs']) / 1e6))
        return
    if sys.argv[1] == '--calls':
        rows = []
        for r in csv.DictReader(open(sys.argv[3])):
            if short(r['Kernel_Name']).startswith(sys.argv[2]):
                rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp'])))
        rows.sort()
        every = int(sys.argv[4]) if len(sys.argv) > 4 else 10
        print('%d calls of %s; every %d-th (index: us): ' % (len(rows), sys.argv[2], every) +
              ' '.join('%d:%.0f' % (i, (e - s) / 1e3) for i, (s, e) in enumerate(rows) if i % every == 0))
        return
    if sys.argv[1] == '--sequence':
        rows = []
        for r in csv.DictReader(open(sys.argv[2])):
            rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), short(r['Kernel_Name'])[:48]))
        rows.sort()
        last = max(i for i, r in enumerate(rows) if r[2].startswith('k_assign'))
        sel = rows[last:]
        t0 = sel[0][0]
        prev, tot = None, 0
        for (s, e, n) in sel:
            print('%9.1f  dur %8.1f  gap %7.1f  %s' % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev) / 1e3 if prev else 0.0, n))
            prev = e
            tot += e - s
        print('launches %d  sum of durations %.1f us  span %.1f us' % (len(sel), tot / 1e3, (sel[-1][1] - t0) / 1e3))
        return
    d, nt = sys.argv[1], int(sys.argv[2])
    top = int(sys.argv[3]) if len(sys.argv) > 3 else 30
    f = glob.glob(d + '/**/*kernel_stats.csv', recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    for r in rows[:top]:
        n = int(r['Calls'])
        t = float(r['TotalDurationNs']) / 1e6
        print('%-44s calls %5d  per-tile %6.3f ms  avg %8.1f us' % (short(r['Name'])[:44], n, t / nt, float(r['AverageNs']) / 1e3))
    skip = ('k_dfs_pool', 'k_small_loop', 'k_fit', 'k_elk', 'k_synthimg')
    print('sum per tile excluding replay / pass loop / fit / synth: %.2f ms' % (
        sum(float(r['TotalDurationNs']) / 1e6 for r in rows if not short(r['Name']).startswith(skip)) / nt))


if __name__ == '__main__':
    main()
