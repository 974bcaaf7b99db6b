"""Wall time of BASELINE config 4 through the product API: Simulation2D.heatmaps(freqs) -- one gain launch for all
frequencies, then one binning + normalisation per frequency -- on the 999-ray C2 fan and a 65 536-ray fan.
Run on the GPU box: python tools/e2e_multifreq.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rt2d_loader
import bench
rt = rt2d_loader.load()
env = bench.make_env(rt)
for n in (999, 1 << 16):
    angles = -(np.arange(n) + 1) * (0.999 / n)
    for F in (8, 64):
        freqs = list(np.linspace(1, 20000, F))
        for rep in range(2):
            sim = rt.Simulation2D(env, np.array([0., 0.]))
            sim.cast(angles, **bench.PROP)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            maps = sim.heatmaps(freqs, resolution=[50, 25], cutoff=.5, reduction_power=.5)
            torch.cuda.synchronize(); t1 = time.perf_counter()
        print(f'{n:6d} rays, {F:3d} frequencies: {1e3 * (t1 - t0):9.2f} ms for {len(maps)} maps '
              f'({1e3 * (t1 - t0) / F:.3f} ms per map)', flush=True)
