"""Timeline of the last bench step in a rocprofv3 kernel_trace.csv: per 20-ms bin, the number of
streams with a kernel in flight, the share of the bin covered by GPU-filling kernels (>= 1024
workgroups) and by the latency-bound ones (k_dfs_split, k_small_loop).  python tools/timeline.py TRACE.csv"""
import csv, sys, collections
rows = []
with open(sys.argv[1]) as fh:
    for r in csv.DictReader(fh):
        g = int(r.get('Grid_Size_X', 0) or 0) * max(int(r.get('Grid_Size_Y', 1) or 1), 1)
        w = max(int(r.get('Workgroup_Size_X', 1) or 1) * max(int(r.get('Workgroup_Size_Y', 1) or 1), 1), 1)
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r.get('Queue_Id', '0'), g // w))
rows.sort()
# the last step starts at the last k_subsample / first k_fit after a gap: take the last k_synthimg-free span
fits = [s for s, e, n, q, b in rows if n.startswith('k_subsample')]
t0 = fits[-1] if fits else rows[0][0]
rows = [r for r in rows if r[0] >= t0]
t1 = max(r[1] for r in rows)
BIN = 20e6
nb = int((t1 - t0) / BIN) + 1
fill = [0.0] * nb; lat = [0.0] * nb; qs = [set() for _ in range(nb)]
def cover(iv, lo, hi):
    iv = sorted((max(s, lo), min(e, hi)) for s, e in iv if e > lo and s < hi)
    tot = 0; cur = None
    for s, e in iv:
        if cur is None or s > cur[1]:
            if cur: tot += cur[1] - cur[0]
            cur = [s, e]
        else: cur[1] = max(cur[1], e)
    if cur: tot += cur[1] - cur[0]
    return tot
F = [(s, e) for s, e, n, q, b in rows if b >= 1024 and not n.startswith(('k_dfs_split', 'k_dfs_pool', 'k_small_loop'))]
Lt = [(s, e) for s, e, n, q, b in rows if n.startswith(('k_dfs_split', 'k_dfs_pool', 'k_small_loop'))]
print('step span %.1f ms; sum of GPU-filling kernel durations %.1f ms; union %.1f ms' % (
    (t1 - t0) / 1e6, sum(e - s for s, e in F) / 1e6, cover(F, t0, t1) / 1e6))
for i in range(nb):
    lo, hi = t0 + i * BIN, min(t0 + (i + 1) * BIN, t1)
    nq = len({q for s, e, n, q, b in rows if e > lo and s < hi})
    nl = sum(1 for s, e in Lt if e > lo and s < hi)
    print('%5.0f ms  queues %2d  latency-bound kernels in flight %2d  filling-union %.2f  sum-of-filling %.2f' % (
        (lo - t0) / 1e6, nq, nl, cover(F, lo, hi) / (hi - lo), sum(min(e, hi) - max(s, lo) for s, e in F if e > lo and s < hi) / (hi - lo)))
# what runs at the very end of the step (after the last latency-bound kernel has finished)
lastwalk = max([e for s, e in Lt] or [t0])
tail = collections.defaultdict(lambda: [0, 0.0])
for s, e, n, q, b in rows:
    if s >= lastwalk:
        k = tail[n.split('(')[0][:44]]
        k[0] += 1; k[1] += (e - s) / 1e6
print('after the last walker (%.1f ms before the end):' % ((t1 - lastwalk) / 1e6))
for n, (c, ms) in sorted(tail.items(), key=lambda kv: -kv[1][1])[:12]:
    print('   %-46s %4d launches %8.2f ms' % (n, c, ms))
