#!/usr/bin/env python3
"""Summarise a rocprofv3 kernel_trace.csv: per-kernel count / avg / min / max duration (us), VGPRs, LDS, grid."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.OrderedDict()
for r in rows:
    k = r['Kernel_Name']
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp']))/1e3
    a = agg.setdefault(k, {'n': 0, 't': 0.0, 'min': 1e30, 'max': 0.0, 'vgpr': r.get('VGPR_Count', r.get('Arch_VGPR_Count', '')),
                           'accum': r.get('Accum_VGPR_Count', ''), 'sgpr': r.get('SGPR_Count', ''), 'lds': r.get('LDS_Block_Size', ''),
                           'grid': (r.get('Grid_Size_X', ''), r.get('Grid_Size_Y', ''), r.get('Grid_Size_Z', '')), 'wg': r.get('Workgroup_Size_X', '')})
    a['n'] += 1; a['t'] += d; a['min'] = min(a['min'], d); a['max'] = max(a['max'], d)
tot = sum(a['t'] for a in agg.values())
print(f'total kernel time {tot/1e3:.3f} ms over {len(rows)} dispatches')
print(f'{"kernel":90s} {"n":>6s} {"total_us":>11s} {"%":>6s} {"avg_us":>9s} {"min_us":>9s} {"max_us":>9s} vgpr sgpr lds grid wg')
for k, a in sorted(agg.items(), key=lambda kv: -kv[1]['t']):
    print(f'{k[:90]:90s} {a["n"]:6d} {a["t"]:11.1f} {100*a["t"]/tot:6.2f} {a["t"]/a["n"]:9.2f} {a["min"]:9.2f} {a["max"]:9.2f} '
          f'{a["vgpr"]} {a["sgpr"]} {a["lds"]} {"x".join(a["grid"])} {a["wg"]}')
