#!/usr/bin/env python3
"""Summarise a rocprofv3 kernel_trace.csv: per-kernel count / avg / min / max duration (us), VGPRs, LDS, grid.

usage: summarize_trace.py kernel_trace.csv [--steady MARKER N | --timeline FIRST LAST [PAD]]
  --timeline FIRST LAST [PAD]   the dispatch sequence of the LAST training step from the launch whose name contains FIRST to the one whose name contains
                      LAST, PAD (default 3) dispatches either side: start offset, duration, gap to the previous dispatch's end, stream (queue) —
                      what actually sits on the critical path between the networks' forward and their backward.
  --steady MARKER N   keep only the dispatches of the last N training steps, delimited by the launches of the kernel whose
                      name contains MARKER (one launch per step), so that MIOpen's first-use solver search during warm-up
                      does not drown the steady state.  Totals are then also printed per step.
"""
import csv, sys, collections
args = sys.argv[1:]
rows = list(csv.DictReader(open(args[0])))
steps = None
if len(args) >= 4 and args[1] == '--timeline':
    first, last, pad = args[2], args[3], int(args[4]) if len(args) > 4 else 3
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    i1 = max(i for i, r in enumerate(rows) if last in r['Kernel_Name'])
    i0 = max(i for i, r in enumerate(rows[:i1]) if first in r['Kernel_Name'])
    sel = rows[max(i0 - pad, 0):i1 + pad + 1]
    t0, prev_end = int(rows[i0]['Start_Timestamp']), None
    print(f'{"start_us":>10s} {"dur_us":>9s} {"gap_us":>8s} {"queue":>6s}  kernel   (t = 0: start of the first "{first}" launch of the last step)')
    for r in sel:
        st, en = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        gap = '' if prev_end is None else f'{(st - prev_end)/1e3:8.2f}'
        print(f'{(st - t0)/1e3:10.2f} {(en - st)/1e3:9.2f} {gap:>8s} {r.get("Queue_Id", ""):>6s}  {r["Kernel_Name"][:110]}')
        prev_end = en if prev_end is None else max(prev_end, en)
    span = (int(rows[i1]['End_Timestamp']) - t0)/1e3
    print(f'span from the start of "{first}" to the end of "{last}": {span:.2f} us')
    sys.exit(0)
if len(args) >= 4 and args[1] == '--steady':
    marker, n = args[2], int(args[3])
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    marks = [int(r['Start_Timestamp']) for r in rows if marker in r['Kernel_Name']]
    if len(marks) > n:
        lo, hi = marks[-n - 1], marks[-1]
        rows = [r for r in rows if lo <= int(r['Start_Timestamp']) < hi]
        steps = n
        print(f'steady-state window: last {n} steps delimited by "{marker}", wall {(hi - lo)/1e6/n:.3f} ms/step')
agg = collections.OrderedDict()
for r in rows:
    k = r['Kernel_Name']
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp']))/1e3
    a = agg.setdefault(k, {'n': 0, 't': 0.0, 'min': 1e30, 'max': 0.0, 'vgpr': r.get('VGPR_Count', r.get('Arch_VGPR_Count', '')),
                           'accum': r.get('Accum_VGPR_Count', ''), 'sgpr': r.get('SGPR_Count', ''), 'lds': r.get('LDS_Block_Size', ''),
                           'grid': (r.get('Grid_Size_X', ''), r.get('Grid_Size_Y', ''), r.get('Grid_Size_Z', '')), 'wg': r.get('Workgroup_Size_X', '')})
    a['n'] += 1; a['t'] += d; a['min'] = min(a['min'], d); a['max'] = max(a['max'], d)
tot = sum(a['t'] for a in agg.values())
# how much of the wall time at least one kernel was running (union of the dispatch intervals over all streams)
iv = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in rows)
busy, cur_s, cur_e = 0, None, None
for s_, e_ in iv:
    if cur_e is None or s_ > cur_e:
        if cur_e is not None: busy += cur_e - cur_s
        cur_s, cur_e = s_, e_
    else: cur_e = max(cur_e, e_)
if cur_e is not None: busy += cur_e - cur_s
if iv: print(f'GPU busy (union of kernel intervals): {busy/1e6:.3f} ms of {(max(e for _, e in iv) - iv[0][0])/1e6:.3f} ms spanned' + (f'  ({busy/1e6/steps:.3f} ms per step)' if steps else ''))
print(f'total kernel time {tot/1e3:.3f} ms over {len(rows)} dispatches' + (f'  ({tot/1e3/steps:.3f} ms, {len(rows)//steps} dispatches per step)' if steps else ''))
smd = sum(a['t'] for k, a in agg.items() if 'smd::' in k)
print(f'smd:: kernels (this library): {smd/1e3:.3f} ms = {100*smd/max(tot, 1e-9):.2f} % of kernel time' + (f'  ({smd/steps:.1f} us per step)' if steps else ''))
print(f'{"kernel":90s} {"n":>6s} {"total_us":>11s} {"%":>6s} {"avg_us":>9s} {"min_us":>9s} {"max_us":>9s} vgpr sgpr lds grid wg')
for k, a in sorted(agg.items(), key=lambda kv: -kv[1]['t']):
    print(f'{k[:90]:90s} {a["n"]:6d} {a["t"]:11.1f} {100*a["t"]/tot:6.2f} {a["t"]/a["n"]:9.2f} {a["min"]:9.2f} {a["max"]:9.2f} '
          f'{a["vgpr"]} {a["sgpr"]} {a["lds"]} {"x".join(a["grid"])} {a["wg"]}')
