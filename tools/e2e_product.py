"""Wall time of the product API itself (Simulation2D.cast + .heatmap), as a user of the reference would call it, for the
999-ray C2 fan and a 1M-ray fan -- bench.py spells the same pipeline out stage by stage; this checks that the Python
mirror adds no hidden per-ray cost.  Run on the GPU box: python tools/e2e_product.py [n_large]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rt2d_loader
import bench
rt = rt2d_loader.load()
env = bench.make_env(rt)
n_large = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
for name, angles in (('999', np.array([i / 1000 for i in range(-1, -1000, -1)])),
                     (str(n_large), -(np.arange(n_large) + 1) * (0.999 / n_large))):
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        sim = rt.Simulation2D(env, np.array([0., 0.]))
        sim.cast(angles, **bench.PROP)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        hm = sim.heatmap(resolution=[50, 25], cutoff=.02, reduction_power=.5)
        torch.cuda.synchronize(); t2 = time.perf_counter()
        print(f'{name:>8s} rays, run {rep}: cast {1e3 * (t1 - t0):8.2f} ms, heatmap {1e3 * (t2 - t1):8.2f} ms, map {hm.shape}', flush=True)
        del sim
