import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
print('cpu_count', os.cpu_count(), 'affinity', len(os.sched_getaffinity(0)), 'torch threads', torch.get_num_threads(), flush=True)
wl = bench.WORKLOADS['cfg2']
for thr in (8, 16, 32, 64):
    if thr > len(os.sched_getaffinity(0)): break
    os.environ['SMD_CPU_THREADS'] = str(thr)
    t = time.time(); r = bench.cpu_baseline(wl, sample_b=2, steps=2); print(thr, 'threads ->', r['value'], 'img/s', f'{time.time()-t:.1f}s', flush=True)
