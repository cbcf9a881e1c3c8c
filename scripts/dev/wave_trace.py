"""Where and when did every wave of k_recon_main run?  (GPU box; library built with EXTRA=-DSMD_TRACE_WAVES)
usage: python scripts/dev/wave_trace.py [cfg2]   -> occupancy statistics per SIMD over the launch."""
import ctypes as C, sys
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from slowtv_monodepth_amd import _lib
import runpy

def main():
    which = 'bwd' if 'bwd' in sys.argv[2:] else 'fwd'      # usage: wave_trace.py [cfg2] [bwd]
    sys.argv = [sys.argv[0], sys.argv[1] if len(sys.argv) > 1 else 'cfg2', '3']
    runpy.run_path(str(Path(__file__).resolve().parent/'microbench.py'), run_name='__main__')   # leaves the trace of its last forward launch
    torch.cuda.synchronize()
    lib = _lib.lib
    fn = lib.smd_debug_wave_trace_bwd if which == 'bwd' else lib.smd_debug_wave_trace
    fn.restype = C.c_int
    n = 1 << 16
    buf = np.zeros((n, 3), dtype=np.uint64)
    rc = fn(buf.ctypes.data_as(C.c_void_p), n)
    print(f'[{which}: {"k_recon_bwd, entry to end of the row loop" if which == "bwd" else "k_recon_main"}]')
    assert rc == 0, rc
    live = buf[:, 1] > 0
    t0, t1, hw = buf[live, 0].astype(np.int64), buf[live, 1].astype(np.int64), buf[live, 2]
    base = t0.min(); t0 -= base; t1 -= base
    tick = 1e-2  # us per s_memrealtime tick (100 MHz)
    xcc = (hw >> np.uint64(32)).astype(np.int64) & 0xf
    hwid = (hw & np.uint64(0xffffffff)).astype(np.int64)
    simd, cu, sh, se = (hwid >> 4) & 3, (hwid >> 8) & 0xf, (hwid >> 12) & 1, (hwid >> 13) & 7
    key = ((xcc*8 + se)*2 + sh)*16*4 + cu*4 + simd
    print(f'waves {live.sum()}  launch span {t1.max()*tick:.1f} us  wave life: mean {(t1 - t0).mean()*tick:.1f} us, p10 {np.percentile(t1 - t0, 10)*tick:.1f}, p90 {np.percentile(t1 - t0, 90)*tick:.1f}')
    print(f'distinct SIMDs used {len(np.unique(key))}  waves per SIMD: min {np.bincount(np.unique(key, return_inverse=True)[1]).min()} max {np.bincount(np.unique(key, return_inverse=True)[1]).max()}')
    span = t1.max()
    grid = np.zeros(span + 1, dtype=np.int64)
    np.add.at(grid, t0, 1); np.add.at(grid, t1, -1)
    conc = np.cumsum(grid)[:-1]
    nsimd = len(np.unique(key))
    print('resident waves per SIMD over time (10 slices):', ' '.join(f'{conc[i*span//10:(i + 1)*span//10].mean()/nsimd:.2f}' for i in range(10)))
    print(f'time-averaged resident waves per SIMD: {conc.mean()/nsimd:.2f}; first wave start spread: {np.percentile(t0, 48)*tick:.1f} us for the first 48 % of the waves')
    st = np.sort(t0)
    print('start time of the k-th wave (us):', ' '.join(f'{k}:{st[min(k, len(st) - 1)]*tick:.1f}' for k in (0, 1024, 2048, 4095, 4096, 5000, 6000, 7000, 8000, len(st) - 1)))

    # which waves are slow?  decode (scale, sample, strip) like smd_kernels.h: decode_wave (4 waves per block) — the layout WITHOUT the shared ring
    # (SMD_FWD_SHARE=0) and without a taper; the occupancy statistics above do not depend on it
    import os
    by_hw = lambda name, keyarr: print(f'mean life by {name}:', ' '.join(f'{k}:{((t1 - t0)*tick)[keyarr == k].mean():.1f}' for k in np.unique(keyarr)[:24]))
    by_hw('XCC', xcc); by_hw('SIMD', simd); by_hw('start decile', np.minimum(t0*10//max(t0.max(), 1), 9))
    if os.environ.get('WAVE_DECODE') != '1': return
    b, h, w, S = 12, 192, 640, 4
    rh = int(os.environ.get('SMD_FWD_RH', '12'))
    nsx, nsy = -(-w//62), -(-h//rh)
    nstrips = nsx*nsy; nbx = -(-nstrips//4)
    widx = np.nonzero(live)[0]
    pblk, wid = widx//4, widx % 4
    nq = nbx*b; full = nq & ~7
    slot = pblk >> 3
    q = np.where(pblk < full*S, (slot//S)*8 + (pblk & 7), full + (pblk - full*S)//S)
    sc = np.where(pblk < full*S, slot % S, (pblk - full*S) % S)
    bi = q//nbx; strip = (q - bi*nbx)*4 + wid
    ok = strip < nstrips
    life = (t1 - t0)*tick
    sx, sy = strip % nsx, strip//nsx
    def by(name, keyarr):
        ks = np.unique(keyarr[ok])
        print(f'mean life by {name}:', ' '.join(f'{k}:{life[ok & (keyarr == k)].mean():.1f}' for k in ks[:24]))
    by('scale', sc); by('strip column', sx); by('strip row', sy); by('sample', bi); by('XCC', xcc); by('SE', se); by('CU', cu); by('SIMD', simd)
    by('start decile', np.minimum(t0*10//max(t0.max(), 1), 9))


if __name__ == '__main__':
    main()
