"""Development aid: where the time of Simulation2D.heatmaps(64 frequencies) on 65 536 rays goes."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rt2d_loader
import bench
rt = rt2d_loader.load()
env = bench.make_env(rt)
n, F = 1 << 16, 64
angles = -(np.arange(n) + 1) * (0.999 / n)
freqs = list(np.linspace(1, 20000, F))
def t():
    torch.cuda.synchronize(); return time.perf_counter()
for rep in range(2):
    sim = rt.Simulation2D(env, np.array([0., 0.]))
    fan = sim.cast(angles, **bench.PROP)
    t0 = t()
    fan._ensure_compact(); t1 = t()
    fan.populate(*freqs); t2 = t()
    maps = sim.heatmaps(freqs, resolution=[50, 25], cutoff=.5, as_tensor=True); t3 = t()
    host = torch.stack([maps[f] for f in freqs]).cpu().numpy(); t4 = t()
    print(f'compact {1e3*(t1-t0):.2f} ms, gain {1e3*(t2-t1):.2f} ms, bin+finalize {1e3*(t3-t2):.2f} ms, copy {1e3*(t4-t3):.2f} ms; points {fan.ray_steps()}')
