"""Development aid: host-side time of the product API calls (how long until the kernels are queued), 999-ray fan."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rt2d_loader
import bench
rt = rt2d_loader.load()
env = bench.make_env(rt)
angles = np.array([i / 1000 for i in range(-1, -1000, -1)])
src = np.array([0., 0.])
for rep in range(6):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    sim = rt.Simulation2D(env, src)
    t1 = time.perf_counter()
    fan = sim.cast(angles, **bench.PROP)
    t2 = time.perf_counter()
    hm = sim.heatmap(resolution=[50, 25], cutoff=.5, as_tensor=True)
    t3 = time.perf_counter()
    torch.cuda.synchronize()
    t4 = time.perf_counter()
    print(f'run {rep}: Simulation2D() {1e6 * (t1 - t0):6.0f} us, cast() returns after {1e6 * (t2 - t1):6.0f} us, heatmap() {1e6 * (t3 - t2):6.0f} us, '
          f'final sync {1e6 * (t4 - t3):5.0f} us, total {1e6 * (t4 - t0):6.0f} us')
import cProfile, pstats
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(20):
    sim = rt.Simulation2D(env, src)
    fan = sim.cast(angles, **bench.PROP)
    hm = sim.heatmap(resolution=[50, 25], cutoff=.5, as_tensor=True)
pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(28)
