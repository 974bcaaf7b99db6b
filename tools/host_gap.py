"""Host time of one 999-ray product step (Simulation2D.cast + heatmap) split at the launch of the cast: what the GPU
waits for between the read that ends one step and the cast of the next.  python tools/host_gap.py"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, rt2d_loader, bench
rt = rt2d_loader.load()
from ray_tracer_amd import _lib
L = _lib.lib()
env = bench.make_env(rt)
src = np.array([0., 0.])
angles = -(np.arange(999) + 1) / 1000.0
marks = {}
orig = L.rt2d_cast_write
def wrapped(*a):
    marks['pre'] = time.perf_counter()
    rc = orig(*a)
    marks['post'] = time.perf_counter()
    return rc
L.rt2d_cast_write = wrapped
for _ in range(10):
    bench.product_step(rt, env, src, angles)
torch.cuda.synchronize()
pre, launch, rest = [], [], []
for _ in range(200):
    t0 = time.perf_counter()
    bench.product_step(rt, env, src, angles)
    t1 = time.perf_counter()
    pre.append(marks['pre'] - t0); launch.append(marks['post'] - marks['pre']); rest.append(t1 - marks['post'])
print(f'per step: host before the cast launch {1e6 * np.median(pre):.1f} us, the launch call {1e6 * np.median(launch):.1f} us, '
      f'rest of the step (queueing + waiting for the GPU) {1e6 * np.median(rest):.1f} us; step {1e6 * (np.median(pre) + np.median(launch) + np.median(rest)):.1f} us')
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(200):
    bench.product_step(rt, env, src, angles)
pr.disable()
pstats.Stats(pr).sort_stats('tottime').print_stats(22)
