#!/usr/bin/env python3
"""Instruction mix of the decoder convolutions' inner loops, from the ISA hipcc generates (no GPU needed): MFMAs against LDS reads / writes, vector-memory
instructions and other vector instructions per K chunk (forward forms) / per row (weight gradients).  VERDICT r5 item 2 asked for the MFMA : LDS-read ratio.
usage: python scripts/dev/conv_isa_stats.py >> profiles/rNN_instruction_counts.txt"""
import re, subprocess, tempfile
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
CSRC = ROOT/'slowtv_monodepth_amd'/'csrc'

def isa(src):
    out = Path(tempfile.mkdtemp())/'k.s'
    subprocess.run(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fno-slp-vectorize', f'-I{ROOT}/include', '-S', '--cuda-device-only',
                    str(CSRC/src), '-o', str(out)], check=True, stderr=subprocess.DEVNULL)
    return out.read_text().split('\n')

def kernel(lines, key):
    start = [i for i, l in enumerate(lines) if l.startswith(key) and l.rstrip().endswith(':') is False and ':' in l][0]
    end = [i for i, l in enumerate(lines) if i > start and 's_endpgm' in l][0]
    body = lines[start:end]
    meta = {k: next((re.search(r'(\d+)', l.split(k)[1]).group(1) for l in lines[end:end + 400] if k in l), '?') for k in ('.vgpr_count', '.agpr_count', 'NumVgprs', 'NumAgprs', 'ScratchSize', 'LDSByteSize')}
    return body, meta

def count(body):
    c = {'mfma': 0, 'ds_read': 0, 'ds_write': 0, 'vmem_load': 0, 'vmem_store': 0, 'valu': 0, 'salu': 0, 'barrier': 0, 'waitcnt': 0}
    for l in body:
        t = l.strip().split(' ')[0] if l.strip() else ''
        if t.startswith('v_mfma'): c['mfma'] += 1
        elif t.startswith('ds_read') or t.startswith('ds_load'): c['ds_read'] += 1
        elif t.startswith('ds_write') or t.startswith('ds_store'): c['ds_write'] += 1
        elif t.startswith('global_load') or t.startswith('buffer_load'): c['vmem_load'] += 1
        elif t.startswith('global_store') or t.startswith('buffer_store'): c['vmem_store'] += 1
        elif t.startswith('v_'): c['valu'] += 1
        elif t == 's_barrier': c['barrier'] += 1
        elif t == 's_waitcnt': c['waitcnt'] += 1
        elif t.startswith('s_'): c['salu'] += 1
    return c

def main_loop(body):
    labels = {m.group(1): i for i, l in enumerate(body) if (m := re.match(r'^(\.LBB\d+_\d+):', l))}
    loops = []
    for i, l in enumerate(body):
        m = re.search(r's_c?branch\w*\s+(\.LBB\d+_\d+)', l)
        if m and m.group(1) in labels and labels[m.group(1)] < i: loops.append((labels[m.group(1)], i))
    best = max(loops, key=lambda ab: sum(1 for l in body[ab[0]:ab[1]] if 'v_mfma' in l)) if loops else (0, len(body))
    return body[best[0]:best[1]]

def report(lines, key, what, unit):
    body, meta = kernel(lines, key)
    whole, loop = count(body), count(main_loop(body))
    if loop['mfma'] == 0: loop = whole                     # (no loop around the MFMAs: the kernel is one chunk)
    print(f'{what}')
    print(f'  main loop ({unit}): {loop["mfma"]} MFMA, {loop["ds_read"]} LDS reads, {loop["ds_write"]} LDS writes, {loop["vmem_load"]} vector-memory loads, {loop["valu"]} other vector, '
          f'{loop["salu"]} scalar, {loop["barrier"]} barriers, {loop["waitcnt"]} waits  ->  MFMA : LDS read = {loop["mfma"]/max(loop["ds_read"], 1):.2f}, other vector per MFMA = {loop["valu"]/max(loop["mfma"], 1):.2f}')
    print(f'  whole kernel: {whole["mfma"]} MFMA, {whole["vmem_load"]} loads, {whole["vmem_store"]} stores; registers: see the resource table of the build (make EXTRA=-Rpass-analysis=kernel-resource-usage)\n')

print('\nDecoder convolutions (round 6), produced by scripts/dev/conv_isa_stats.py\n')
L = isa('smd_conv_mfma.hip')
report(L, '_ZN3smd11k_conv_mfmaILi64ELi3ELb0EffLi1EE', 'k_conv_mfma<64, 3, false, float, float>  (wide forward, fp32 tensors, three pieces: 4 rows x 64 columns x 32 output channels per block; a chunk = 16 channels x 9 taps)', 'one K chunk of one wave: 9 taps x 2 pixel tiles x 6 products = 108 MFMA')
report(L, '_ZN3smd11k_conv_mfmaILi64ELi3ELb1EffLi1EE', 'k_conv_mfma<64, 3, true, float, float>   (wide data gradient)', 'one K chunk')
report(L, '_ZN3smd11k_conv_mfmaILi64ELi1ELb0E14__hip_bfloat16S1_Li1EE', 'k_conv_mfma<64, 1, false, bf16, bf16>    (wide forward, bf16 tensors, one piece)', 'one K chunk: 18 MFMA')
report(L, '_ZN3smd13k_conv16_mfmaILi1ELi3ELb0EffEE', 'k_conv16_mfma<1, 3, false, float, float> (thin forward 16 -> 16: 16x16x32 MFMA, two taps per K step, weights in registers; no loop — one chunk)', 'the whole tile: 5 K steps x 4 rows x 6 products = 120 MFMA')
report(L, '_ZN3smd16k_conv_wgrad_dmaILi3EEE', 'k_conv_wgrad_dma<3>                      (wide weight gradient, fp32 tensors: 32 output x 64 input channels per block, loop over INPUT rows, ring of four RAW rows by LDS-DMA, split at fragment read; the loop body is three rows)', 'three input rows of one wave (its K step of 16 columns, one tile of 32 input channels): 3 x 54 MFMA')
report(L, '_ZN3smd18k_conv16_wgrad_dmaILi1ELi3EEE', 'k_conv16_wgrad_dma<1, 3>                 (thin weight gradient 16 -> 16, fp32 tensors: the same on 16x16x32, strips of 64 columns)', 'three input rows of one wave')
report(L, '_ZN3smd18k_conv16_wgrad_dmaILi2ELi3EEE', 'k_conv16_wgrad_dma<2, 3>                 (thin weight gradient 32 -> 16)', 'three input rows of one wave')
report(L, '_ZN3smd17k_conv_wgrad_mfmaILi1E14__hip_bfloat16EE', 'k_conv_wgrad_mfma<1, bf16>               (wide weight gradient, bf16 tensors: rows staged through registers, split + filed after the row\'s MFMAs)', 'one input row of one wave: 9 MFMA')
T = isa('smd_conv_thin.hip')
report(T, '_ZN3smd11k_thin_mfmaILi16ELi1ELb0EEE', 'k_thin_mfma<16, 1, false>                (round 5, f32 MFMA 16x16x4: the comparison)', 'the whole tile')
