#!/bin/bash
# GPU busy fraction and per-kernel time under the real two-lane pipeline: tools/trace_busy.sh <tag>
export TMPDIR=/tmp
tag=$1
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace_$tag -o t -- python3 bench.py --steps 12 --warmup 2 --cpu-seconds 0 --preheat-seconds 1 > gpurun_out/trace_$tag.log 2>&1
python3 - <<PY
import csv, collections
rows = list(csv.DictReader(open("gpurun_out/trace_$tag/t_kernel_trace.csv")))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:40]) for r in rows)
# take the last 40 % of the run (timed region + kernel-level context excluded roughly): use a window of steady state
t0, t1 = ev[0][0], ev[-1][1]
lo, hi = t0 + 0.45 * (t1 - t0), t0 + 0.75 * (t1 - t0)
win = [e for e in ev if e[0] >= lo and e[1] <= hi]
busy, cur_s, cur_e = 0, None, None
for s, e, _ in win:
    if cur_e is None or s > cur_e:
        if cur_e is not None: busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
span = win[-1][1] - win[0][0]
tot = collections.Counter(); n = collections.Counter()
for s, e, k in win: tot[k] += e - s; n[k] += 1
niter = n[[k for k in n if "k_iter" in k][0]]
print("window %.1f ms, %d frames, %.3f ms/frame, GPU busy %.1f %%, sum of kernel time / span = %.2f" % (span / 1e6, niter, span / 1e6 / niter, 100.0 * busy / span, sum(tot.values()) / span))
for k, v in tot.most_common(12): print("  %-42s %7.1f us/frame (%d launches/frame)" % (k, v / 1e3 / niter, round(n[k] / niter)))
PY
