"""Where the time of Simulation2D.heatmaps goes: F frequencies on one fan (gain for all frequencies, binning of all maps,
finalize + copy).  python tools/multifreq_stages.py [rays] [freqs]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, rt2d_loader, bench
rt = rt2d_loader.load()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
F = int(sys.argv[2]) if len(sys.argv) > 2 else 64
env = bench.make_env(rt)
source = np.array([0., 0.])
angles = -(np.arange(n) + 1) * (0.999 / n)
freqs = np.geomspace(10., 20000., F)
sim = rt.Simulation2D(env, source)
sim.cast(angles, **bench.PROP)
fan = sim.raypacks[sim.pack_default].segments()[0][0]
torch.cuda.synchronize()
for rep in range(3):
    fan.G.clear(); fan.G_end.clear()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    fan.populate(*freqs)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    hms = sim.heatmaps(freqs, resolution=[50, 25])
    torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f'{n} rays x {F} freqs: gain {1e3 * (t1 - t0):.2f} ms, maps (bin + finalize + copy) {1e3 * (t2 - t1):.2f} ms, '
          f'{1e3 * (t2 - t0) / F:.3f} ms per map')
for rep in range(3):
    fan.G.clear(); fan.G_end.clear()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    hms = sim.heatmaps(freqs, resolution=[50, 25])
    torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f'{n} rays x {F} freqs, heatmaps() from scratch: {1e3 * (t2 - t0):.2f} ms, {1e3 * (t2 - t0) / F:.3f} ms per map')
