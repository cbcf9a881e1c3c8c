#!/usr/bin/env python3
"""gpurun_out/pmc_<tag>_bench_<wl>/summary.txt -> the JSON bench.py reads as profiles/traffic.json (scripts/pmc_traffic.sh).

Counter units and the gfx950 correction follow /opt/skills/guides/MI355X_MICROARCH.md: FETCH_SIZE / WRITE_SIZE are in KiB, and
FETCH_SIZE counts 64 B for each 128-byte read request on gfx950, so reads are doubled (calibrated in round 1 on a kernel of known
traffic, profiles/r01_pmc_microbench_final_kernels.txt)."""
import json, re, sys
from pathlib import Path

root = Path(__file__).resolve().parents[1]
tag, wls = sys.argv[1], sys.argv[2:]
out = {'_source': f'scripts/pmc_traffic.sh {tag}: rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE | TCC_HIT_sum TCC_MISS_sum (three separate passes) over '
                  '`python bench.py --workload <wl> --steps 3 --warmup 2 --no-cpu-baseline`, per-launch averages; KiB units, FETCH_SIZE x2 on gfx950',
       '_summaries': [f'profiles/{tag}_pmc_bench_{wl}.txt' for wl in wls]}
for wl in wls:
    kernels, cur = {}, None
    for line in (root/'gpurun_out'/f'pmc_{tag}_bench_{wl}'/'summary.txt').read_text().splitlines():
        m = re.match(r'\s+(\S+)\s+avg/dispatch\s+([0-9.]+)\s+\(n=(\d+)\)', line)
        if m and cur is not None: kernels[cur][m.group(1)] = (float(m.group(2)), int(m.group(3)))
        elif line and not line.startswith(' '): cur = line.strip(); kernels[cur] = {}
    ent = {}
    for key, pat in (('recon_fwd', 'k_recon_main<'), ('recon_prep', 'k_recon_prep<'), ('recon_bwd', 'k_recon_bwd<')):
        ks = [k for k in kernels if pat in k and 'FETCH_SIZE' in kernels[k]]
        if not ks: continue
        k = max(ks, key=lambda k: kernels[k]['FETCH_SIZE'][1])          # the instantiation the run actually used most
        c = kernels[k]
        rd, wr = c['FETCH_SIZE'][0]*1024*2, c['WRITE_SIZE'][0]*1024
        ent[f'{key}_kernel'] = k
        ent[f'{key}_read_bytes_corrected'] = int(rd); ent[f'{key}_write_bytes'] = int(wr); ent[f'{key}_bytes'] = int(rd + wr)
        ent[f'{key}_launches_counted'] = c['FETCH_SIZE'][1]
        if 'TCC_HIT_sum' in c and 'TCC_MISS_sum' in c:
            ent[f'l2_hit_rate_{key}'] = round(c['TCC_HIT_sum'][0]/max(c['TCC_HIT_sum'][0] + c['TCC_MISS_sum'][0], 1.0), 3)
    out[wl] = ent
print(json.dumps(out, indent=2))
