"""Summarise a rocprofv3 kernel_trace.csv: GPU busy fraction, concurrency, per-stream gaps."""
import csv, sys, glob, collections
f = sys.argv[1]
rows = []
with open(f) as fh:
    rd = csv.DictReader(fh)
    cols = rd.fieldnames
    for r in rd:
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'][:40], r.get('Queue_Id', '0'),
                     int(r.get('Grid_Size_X', 0) or 0), int(r.get('Workgroup_Size_X', 0) or 0)))
print('columns', cols)
rows.sort()
t0 = rows[0][0]; t1 = max(r[1] for r in rows)
# restrict to the last 40% of the span (timed steps)
cut = t0 + int((t1 - t0) * 0.6)
rows = [r for r in rows if r[0] >= cut]
t0 = rows[0][0]; t1 = max(r[1] for r in rows)
ev = []
for s, e, *_ in rows:
    ev.append((s, 1)); ev.append((e, -1))
ev.sort()
busy = 0; conc_time = collections.Counter(); cur = 0; last = ev[0][0]
for t, d in ev:
    if cur > 0: busy += t - last
    conc_time[min(cur, 20)] += t - last
    cur += d; last = t
span = t1 - t0
print('span ms %.1f busy frac %.3f' % (span / 1e6, busy / span))
print('time share by #concurrent kernels:', {k: round(v / span, 3) for k, v in sorted(conc_time.items())})
# "large" kernels (>= 2048 workgroups) union
big = [(s, e) for s, e, n, q, g, w in rows if w and g // max(w, 1) >= 2048]
ev = sorted([(s, 1) for s, e in big] + [(e, -1) for s, e in big])
cur = 0; last = ev[0][0] if ev else 0; bb = 0
for t, d in ev:
    if cur > 0: bb += t - last
    cur += d; last = t
print('union of kernels with >=2048 workgroups: %.3f of span; sum of their durations %.1f ms' % (bb / span, sum(e - s for s, e in big) / 1e6))
# per-queue gaps
byq = collections.defaultdict(list)
for s, e, n, q, g, w in rows: byq[q].append((s, e, n))
gaps = []
for q, lst in byq.items():
    lst.sort()
    for a, b in zip(lst, lst[1:]):
        gaps.append(b[0] - a[1])
gaps.sort()
import statistics
print('queues', len(byq), 'gap between consecutive kernels of a queue: median %.1f us p90 %.1f us mean %.1f us sum %.1f ms' % (
    gaps[len(gaps)//2] / 1e3, gaps[int(len(gaps)*0.9)] / 1e3, statistics.mean(gaps) / 1e3, sum(g for g in gaps if g > 0) / 1e6))

# per-queue busy fraction
for q, lst in sorted(byq.items()):
    b = sum(e - s for s, e, n in lst)
    print('queue', q, 'kernels', len(lst), 'busy %.3f' % (b / span), 'first %.1f ms last %.1f ms' % ((lst[0][0]-t0)/1e6, (max(e for s,e,n in lst)-t0)/1e6))
# time-weighted kernel mix
mix = collections.Counter()
for s, e, n, q, g, w in rows: mix[n.split('(')[0]] += e - s
tot = sum(mix.values())
print('kernel-time share:', [(k, round(v / tot, 3)) for k, v in mix.most_common(8)], 'sum/span = %.2f' % (tot / span))
