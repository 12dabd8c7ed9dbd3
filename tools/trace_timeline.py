"""Per-frame timeline from a rocprofv3 --kernel-trace CSV: for the last few frames print each kernel's
start (relative to the frame's first k_iter start), duration and stream/queue; plus the busy fraction."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows = [r for r in rows]
for r in rows:
    r['s'] = int(r['Start_Timestamp']); r['e'] = int(r['End_Timestamp'])
rows.sort(key=lambda r: r['s'])
iters = [i for i, r in enumerate(rows) if 'k_iter' in r['Kernel_Name']]
# take the window of frames [-6, -2)
a, b = iters[-6], iters[-2]
t0 = rows[a]['s']
busy = 0; last = t0
for r in rows[a:b]:
    name = r['Kernel_Name'].split('(')[0][:28]
    print('%9.1f us  +%7.1f us  q%-3s %s' % ((r['s'] - t0) / 1e3, (r['e'] - r['s']) / 1e3, r.get('Queue_Id', '?'), name))
    s = max(r['s'], last)
    if r['e'] > s:
        busy += r['e'] - s; last = r['e']
print('window %.1f us, busy %.1f us (%.1f %%), %.3f ms per frame' % ((rows[b]['s'] - t0) / 1e3, busy / 1e3, 100.0 * busy / (rows[b]['s'] - t0), (rows[b]['s'] - t0) / 4e6))
