# Reads what tools/de_slot_budget.sh collected and prints the DE's vector-ALU slot budget (see there).
import csv, collections, glob, sys, re
root = sys.argv[1]
TRANS_SLOTS = 3.2        # a v_exp_f32 / v_log_f32 / v_rcp_f32 occupies the vector ALU 3.4 ns against 1.05 ns for a v_fma_f32 (profiles/r03_valu_lds_microbench.txt, r04_vgpr_bank_bench.txt)
def direction(name):
    m = re.search(r'k_de_dir<(\d)', name)
    return int(m.group(1)) if m else None
def durations(b):
    out = collections.defaultdict(list)
    for r in csv.DictReader(open('%s/sb_%s_t/b_kernel_stats.csv' % (root, b))):
        d = direction(r['Name'])
        if d is not None: out[d].append((float(r['AverageNs']) / 1e3, int(r['Calls'])))
    return {d: sum(a * n for a, n in v) / sum(n for a, n in v) for d, v in out.items()}
def counters(b):
    per = collections.defaultdict(lambda: collections.defaultdict(lambda: collections.defaultdict(float)))
    for f in sorted(glob.glob('%s/sb_%s_c*/b_counter_collection.csv' % (root, b))):
        for r in csv.DictReader(open(f)):
            d = direction(r['Kernel_Name'])
            if d is not None: per[d][r['Counter_Name']][r['Dispatch_Id']] += float(r['Counter_Value'])
    out = {}
    for d, cs in per.items():
        out[d] = {}
        for c, disp in cs.items():
            v = sorted(disp.values()); out[d][c] = v[len(v) // 2]
    return out
dur = {b: durations(b) for b in ('full', 'stage', 'taps')}
cnt = {b: counters(b) for b in ('full', 'stage', 'taps')}
import hashlib, os
print('# lib_sha256: %s' % hashlib.sha256(open(os.environ.get('DE_FULL_LIB', 'cuburn_amd/_lib/libflame_hip.so'), 'rb').read()).hexdigest())      # the "full" build: bench.py quotes this budget for that library only
print('# DE slot budget (tools/de_slot_budget.sh): vector-ALU instructions per OUTPUT PIXEL = per launched lane (SQ_INSTS_VALU / waves), by build')
print('# slot-equivalents = VALU + %.1f * transcendental (a transcendental holds the ALU %.1f slots).  stage = until the planes are staged,' % (TRANS_SLOTS - 1, TRANS_SLOTS))
print('# taps = tap loop + epilogue on unstaged LDS; full = the shipped kernel.  us = rocprofv3 average duration.')
print('%-4s %-6s %8s %8s %8s %8s %8s %10s %8s %10s' % ('dir', 'build', 'us', 'VALU/px', 'trans/px', 'int/px', 'LDS/px', 'slot-eq/px', 'SALU/px', 'ns/slot-eq'))
tot = collections.defaultdict(float)
rows = {}
for d in range(8):
    for b in ('full', 'stage', 'taps'):
        c = cnt[b].get(d, {}); w = c.get('SQ_WAVES', 0)
        if not w or d not in dur[b]: continue
        valu = c['SQ_INSTS_VALU'] / w; tr = c['SQ_INSTS_VALU_TRANS_F32'] / w; it = c['SQ_INSTS_VALU_INT32'] / w; lds = c['SQ_INSTS_LDS'] / w; salu = c['SQ_INSTS_SALU'] / w
        se = valu + (TRANS_SLOTS - 1) * tr
        nss = dur[b][d] * 1e3 * 1024 / (w * se)          # ns per slot-equivalent per SIMD
        rows[(d, b)] = (dur[b][d], valu, tr, it, lds, se, salu, nss, w)
        print('%-4d %-6s %8.1f %8.1f %8.1f %8.1f %8.1f %10.1f %8.1f %10.3f' % (d, b, dur[b][d], valu, tr, it, lds, se, salu, nss))
        tot[b] += dur[b][d]
print('sum of eight directions: full %.1f us, stage-only %.1f us, taps-only %.1f us' % (tot['full'], tot['stage'], tot['taps']))
print()
print('# per direction: slot-equivalents by phase (staging = stage build; taps + epilogue = full - stage), and what ONE rate explains')
ses = [(d, rows[(d, 'full')][5], rows[(d, 'full')][0], rows[(d, 'full')][8]) for d in range(8) if (d, 'full') in rows]
num = sum(t * (se * w / 1024e3) for d, se, t, w in ses); den = sum((se * w / 1024e3) ** 2 for d, se, t, w in ses)
rate = num / den
print('fitted rate: %.3f ns per slot-equivalent per SIMD (independent v_fma_f32: 1.05; the tap loop in isolation: 19.6 ns per tap / 16.2 slot-eq = 1.21)' % rate)
print('%-4s %10s %10s %10s %10s %10s %8s' % ('dir', 'staging', 'taps+epi', 'total', 'model us', 'measured', 'error'))
for d, se, t, w in ses:
    st = rows[(d, 'stage')][5] if (d, 'stage') in rows else float('nan')
    model = rate * se * w / 1024e3
    print('%-4d %10.1f %10.1f %10.1f %10.1f %10.1f %+7.1f%%' % (d, st, se - st, se, model, t, 100 * (model - t) / t))
# the same with a fixed cost per launch (ramp-up and tail: every workgroup stages at once at the start, few are left at the end)
xs = [se * w / 1024e3 for d, se, t, w in ses]; ts = [t for d, se, t, w in ses]
n = float(len(xs)); sx, sy = sum(xs), sum(ts); sxx = sum(x * x for x in xs); sxy = sum(x * y for x, y in zip(xs, ts))
a2 = (n * sxy - sx * sy) / (n * sxx - sx * sx); c2 = (sy - a2 * sx) / n
print()
print('with a fixed cost per launch: %.3f ns per slot-equivalent per SIMD + %.1f us per launch' % (a2, c2))
print('%-4s %10s %10s %8s' % ('dir', 'model us', 'measured', 'error'))
for (d, se, t, w), x in zip(ses, xs):
    m = a2 * x + c2
    print('%-4d %10.1f %10.1f %+7.1f%%' % (d, m, t, 100 * (m - t) / t))
