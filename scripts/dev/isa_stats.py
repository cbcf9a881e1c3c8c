#!/usr/bin/env python3
"""Vector-instruction statistics of the row loops of the two fused kernels, from the ISA hipcc generates (no GPU needed).
usage: python scripts/dev/isa_stats.py > profiles/rNN_instruction_counts.txt
Costs per class are the issue rates measured on gfx950 with >= 2 waves per SIMD (profiles/r02_valu_rate2.txt)."""
import collections, re, subprocess, sys, tempfile
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
CSRC = ROOT/'slowtv_monodepth_amd'/'csrc'
FULL = {'v_fma_f32', 'v_fmac_f32', 'v_add_f32', 'v_sub_f32', 'v_subrev_f32', 'v_mul_f32', 'v_mov_b32', 'v_add_u32', 'v_fmaak_f32', 'v_fmamk_f32',
        'v_sub_u32', 'v_and_b32', 'v_or_b32', 'v_xor_b32', 'v_cndmask_b32'}

def isa(src):
    out = Path(tempfile.mkdtemp())/'k.s'
    subprocess.run(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fno-slp-vectorize', f'-I{ROOT}/include', '-S', '--cuda-device-only',
                    str(CSRC/src), '-o', str(out)], check=True, stderr=subprocess.DEVNULL)
    return out.read_text().split('\n')

def loop_stats(lines, key, rows_per_iter, what, nested=False):
    found = [i for i, l in enumerate(lines) if l.startswith(key)]
    if not found:
        print(f'{what}\n  (not in this build: experiments-only kernels need -DSMD_EXPERIMENTS)\n'); return
    start = found[0]
    end = [i for i, l in enumerate(lines) if i > start and 's_endpgm' in l][0]
    body = lines[start:end]
    labels = {m.group(1): i for i, l in enumerate(body) if (m := re.match(r'^(\.LBB\d+_\d+):', l))}
    loops = []
    for i, l in enumerate(body):
        m = re.search(r's_c?branch\w*\s+(\.LBB\d+_\d+)', l)
        if m and m.group(1) in labels and labels[m.group(1)] < i: loops.append((labels[m.group(1)], i))
    # the row loop: the largest loop (forward), or the largest loop nested inside another one (backward: inside the loop over supports)
    loops = sorted(set(loops), key=lambda ab: ab[0] - ab[1])
    # backward: the row loop is the LARGEST loop that sits inside another one (the loop over this wave's supports) and itself
    # encloses no loop of comparable size (the wave-uniform skip branches of SKIP=2 are forward jumps, not loops)
    if nested: loops = [ab for ab in loops if any(o[0] <= ab[0] and o[1] >= ab[1] and o != ab and (o[1] - o[0]) > 1.1*(ab[1] - ab[0]) for o in loops)]
    if not nested: loops = [ab for ab in loops if (ab[1] - ab[0]) < 0.5*len(body)] or loops   # (round 5: the guest blocks' code wraps the kernel in one more back edge)
    a, b = loops[0]
    if nested:   # SKIP=2: the skip branches rotate the loop, leaving several overlapping back edges — take their union
        changed = True
        while changed:
            changed = False
            for (x, y) in loops:
                if x <= b and y >= a and (x < a or y > b) and (max(y, b) - min(x, a)) < 1.6*(b - a): a, b, changed = min(a, x), max(b, y), True
    seg = [l.strip() for l in body[a:b + 1] if l.strip() and not l.strip().startswith(('.', ';'))]
    c, cost, vmem, lds, salu = collections.Counter(), 0.0, 0, 0, 0
    for l in seg:
        op = l.split()[0]
        if op.startswith(('buffer_', 'global_', 'scratch_')): vmem += 1
        elif op.startswith('ds_'): lds += 1
        elif op.startswith('s_'): salu += 1
        if not op.startswith('v_'): continue
        base = re.sub(r'_(e32|e64|dpp|sdwa)$', '', op)
        if 'dpp' in l: k, w = 'DPP', 4.5
        elif base.startswith(('v_rcp', 'v_sqrt', 'v_rsq', 'v_exp', 'v_log', 'v_cos', 'v_sin')): k, w = 'transcendental', 8.4
        elif base in FULL or base.startswith('v_cmp'): k, w = 'full rate (fma/add/mul/mov/cndmask/cmp)', 2.65
        else: k, w = 'half rate (min/max/med3/floor/cvt/bfi/int mul/lane)', 4.4
        c[k] += 1; cost += w
    n = sum(c.values())
    print(f'{what}\n  row loop = {rows_per_iter} row steps per iteration: {n/rows_per_iter:.0f} vector instructions per row step '
          f'({vmem/rows_per_iter:.1f} vector memory, {lds/rows_per_iter:.1f} LDS, {salu/rows_per_iter:.0f} scalar); estimated issue cycles per row step {cost/rows_per_iter:.0f} '
          f'(average {cost/n:.2f} per instruction)')
    for k, v in c.most_common(): print(f'    {v/rows_per_iter:7.1f}  {k}')
    for l in lines[end:]:
        pass

def regs(lines, key):
    hit = [k for k, l in enumerate(lines) if '.name:' in l and key in l]
    if not hit: return '(not in this build)'
    i = hit[0]
    blk = '\n'.join(lines[i - 30:i + 30])
    v = re.search(r'\.vgpr_count:\s+(\d+)', '\n'.join(lines[i:i + 30])); s = re.search(r'\.vgpr_spill_count:\s+(\d+)', '\n'.join(lines[i:i + 30]))
    return f'{v.group(1)} VGPRs, {s.group(1)} spilled'

if __name__ == '__main__':
    f, bw = isa('smd_recon_fwd.hip'), isa('smd_recon_bwd.hip')
    print('Instruction statistics of the fused kernels\' row loops (hipcc ROCm 7.2, gfx950), produced by scripts/dev/isa_stats.py\n')
    for key, what in (('_ZN3smd12k_recon_mainILi2ELb1ELb1ELb0ELb1ELi1ELb1EEE', 'k_recon_main<2, true, true, false, true, 1, true>  (two supports, K0 fused, shared target ring: the bench\'s forward kernel; the static count includes the once-per-four-rows epoch block)'),
                      ('_ZN3smd12k_recon_mainILi2ELb1ELb1ELb0ELb1ELi1ELb0EEE', 'k_recon_main<2, true, true, false, true, 1, false> (the same without the shared ring: SMD_FWD_SHARE=0, or a pyramid that is not four scales)'),
                      ('_ZN3smd12k_recon_mainILi2ELb1ELb1ELb0ELb0ELi1ELb0EEE', 'k_recon_main<2, true, true, false, false, 1, false> (two supports, depth read from a K0 launch)'),
                      ('_ZN3smd12k_recon_mainILi4ELb1ELb1ELb0ELb1ELi1ELb1EEE', 'k_recon_main<4, true, true, false, true, 1, true>  (four supports, cfg 5)')):
        loop_stats(f, key, 2, what); print('  ' + regs(f, key) + '\n')
    for key, what in (('_ZN3smd11k_recon_bwdILb1ELi0ELi2ELb1ELb0EEE', 'k_recon_bwd<true, 0, 2, true, false> (one support per wave, plain row loop: the steady-state body of the peeled pipeline, every row does the full adjoint)'),
                      ('_ZN3smd11k_recon_bwdILb1ELi2ELi2ELb1ELb0EEE', 'k_recon_bwd<true, 2, 2, true, false> (one support per wave, liveness-gated row loop; the static count includes the clear / dead-row paths)'),
                      ('_ZN3smd11k_recon_bwdILb1ELi0ELi4ELb1ELb0EEE', 'k_recon_bwd<true, 0, 4, true, false> (cfg 5: four supports, one per wave)')):
        loop_stats(bw, key, 3, what, nested=True); print('  ' + regs(bw, key) + '\n')
    key = '_ZN3smd16k_recon_bwd_pairILb1ELi1EEE'
    loop_stats(bw, key, 3, 'k_recon_bwd_pair<true, 1> (experiment: TWO supports per wave, per row step and PAIR; "vector memory" includes the scratch instructions of its spills)', nested=False)
    print('  ' + regs(bw, key) + '\n')
    print('History (same method): round 1 forward 425 per row for two supports; backward 421 per support row step at the start of round 2 (73 of them v_mov),\n'
          '328 after the (row mod 3) slot rewrite, 300 / 313 at the end of round 2; round 3 left the backward\'s row loop as it was (301 / 313: the launch structure changed).\n'
          'Forward with the shared target ring: of the 16 vector-memory instructions counted per row step, 3 are the LDS-DMA pieces of the epoch block, which runs once\n'
          'every four row steps: a row step executes 13 + 0.75 (without the ring: 16).\n'
          'The SKIP=2 build keeps its rarely taken clear paths inside the loop body: its static count is an upper bound of what a live row executes.\n'
          'Round 4: backward 302 plain (peeled pipeline, re-added window sums: 126 -> 111 VGPRs) / 333 gated (static); two supports per wave: 539 per PAIR of row steps against 2 x 302.\n'
          'End of round 4: no divergent control flow left inside a row step.  Backward 302 -> 298 vector / 84 -> 59 scalar (branch-free row reflection, the g_in and\n'
          'several-supports-per-wave paths compiled out of the common instantiation); forward 352 -> 358 with the stores outside `if (interior)` (out-of-range offsets drop\n'
          'them), which removed the one spilled VGPR (128 -> 126) and took <4,...> from 168 to 158 VGPRs (r04_bwd_variants.txt, box 3: backward -3 %, cfg 5 forward -3.4 %).\n'
          'Round 5: forward 358 -> 362 (one v_cmp per support and row for the liveness table + its scalar or; 126 -> 127 VGPRs); the guest blocks (smoothness sweep) sit\n'
          'outside the row loop.  Backward row loops unchanged (298 / 329); the liveness probe (one vector load + ballot before the start-up barrier) takes the plain\n'
          'instantiations from 112 to 116 VGPRs, nothing spilled; the gated loop does not probe (128 VGPRs, at its limit).')
