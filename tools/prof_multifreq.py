import os, sys, time, cProfile, pstats
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, rt2d_loader, bench
rt = rt2d_loader.load()
env = bench.make_env(rt)
n = 1 << 16
angles = -(np.arange(n) + 1) * (0.999 / n)
freqs = list(np.linspace(1, 20000, 8))
for rep in range(2):
    sim = rt.Simulation2D(env, np.array([0., 0.]))
    sim.cast(angles, **bench.PROP)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    sim.raypacks['main'].populate(*freqs)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    pr = cProfile.Profile(); pr.enable()
    for f in freqs:
        sim.heatmap(freq=f, resolution=[50, 25], cutoff=.5, reduction_power=.5)
    torch.cuda.synchronize(); pr.disable(); t2 = time.perf_counter()
    print(f'populate {1e3*(t1-t0):.2f} ms, 8 heatmaps {1e3*(t2-t1):.2f} ms')
pstats.Stats(pr).sort_stats('cumulative').print_stats(18)
