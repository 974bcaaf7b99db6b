"""Kernel timeline of one product step of the 999-ray headline (run under rocprofv3 --kernel-trace, read the database with
the snippet in DESIGN.md section 6), and its wall time."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, rt2d_loader, bench
rt = rt2d_loader.load()
env = bench.make_env(rt)
source = np.array([0., 0.])
angles = -(np.arange(999) + 1) / 1000.
for _ in range(20):
    fan, hm = bench.product_step(rt, env, source, angles)
torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(200):
        fan, hm = bench.product_step(rt, env, source, angles)
    torch.cuda.synchronize()
    print('ms per step', (time.perf_counter() - t0) / 200 * 1e3)
