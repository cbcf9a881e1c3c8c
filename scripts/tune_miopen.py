#!/usr/bin/env python3
"""One-off MIOpen solver search for the convolutions of a bench workload (run on the GPU box through gpurun).

PyTorch's default (cudnn.benchmark = False) asks MIOpen for its heuristic pick ("immediate mode"); with a user find-db
present MIOpen answers from the measured entries instead.  This script runs two training steps with benchmark = True so
that MIOpen measures every applicable solver for every (layer, direction), and leaves the resulting user find-db in
`$MIOPEN_USER_DB_PATH` (default gpurun_out/miopen_db) for inspection / check-in under slowtv_monodepth_amd/miopen_db/.

    MIOPEN_USER_DB_PATH=$PWD/gpurun_out/miopen_db python scripts/tune_miopen.py [--workload cfg2]
"""
import argparse, os, sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
os.environ.setdefault('MIOPEN_USER_DB_PATH', str(ROOT/'gpurun_out'/'miopen_db'))
os.environ.setdefault('MIOPEN_CUSTOM_CACHE_DIR', '/tmp/miopen_kcache')
os.environ.setdefault('MIOPEN_FIND_MODE', '1')          # NORMAL: measure all applicable solvers
os.environ['SMD_NO_MIOPEN_DB'] = '1'                       # the package must not redirect MIOPEN_USER_DB_PATH to its private copy
Path(os.environ['MIOPEN_USER_DB_PATH']).mkdir(parents=True, exist_ok=True)
for f in (ROOT/'slowtv_monodepth_amd'/'miopen_db').glob('*.txt'):   # start from the entries already shipped: results accumulate
    dst = Path(os.environ['MIOPEN_USER_DB_PATH'])/f.name
    if not dst.exists(): dst.write_bytes(f.read_bytes())
import torch
import bench
from slowtv_monodepth_amd.synthetic import make_batch
from slowtv_monodepth_amd.train import StepModule, train_steps
from slowtv_monodepth_amd.trainer import MonoDepthModule

ap = argparse.ArgumentParser(); ap.add_argument('--workload', nargs='+', default=['cfg2']); args = ap.parse_args()
torch.backends.cudnn.benchmark = True
dev = torch.device('cuda:0')
for name in args.workload:
    wl = bench.WORKLOADS[name]
    module = MonoDepthModule(bench.make_cfg(wl, False)).to(dev)
    opt = module.configure_optimizers()['optimizer']
    batch = make_batch(wl['b'], wl['h'], wl['w'], wl['supp'], seed=42, device=dev)
    model = StepModule(module)
    t0 = time.perf_counter()
    for i in range(2):
        train_steps(model, opt, lambda it: batch, 1); torch.cuda.synchronize()
        print(f'[tune {name} +{time.perf_counter() - t0:7.1f}s] step {i} done', flush=True)
    t1 = time.perf_counter()
    train_steps(model, opt, lambda it: batch, 10); torch.cuda.synchronize()
    print(f'[tune {name}] steady state with searched solvers: {(time.perf_counter() - t1)/10*1e3:.2f} ms/step = {wl["b"]*10/(time.perf_counter() - t1):.1f} img/s', flush=True)
    del module, opt, model, batch
    torch.cuda.empty_cache()
for p in sorted(Path(os.environ['MIOPEN_USER_DB_PATH']).rglob('*')):
    if p.is_file(): print(f'[tune] {p}  {p.stat().st_size} bytes')
