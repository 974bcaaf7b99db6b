"""The 1 M-ray fan over the bathymetry cut of bench.py's large_fan_bathymetry block through the product API, a few steps:
for kernel traces (bash tools/trace_script.sh TAG tools/product_large_bathy.py).  python tools/product_large_bathy.py [rays] [steps]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, rt2d_loader, bench
rt = rt2d_loader.load()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
bmap = rt.Map.from_asc(os.path.join(ROOT, 'tests', 'golden', 'synth_bathy.asc'))
b0, b1 = (3.05, 41.7), (3.9, 41.75)
length = float(bmap.cut_length(b0, b1))
env = rt.Environment2D(bmap.floor_table(b0, b1, npoints=751), (np.array([0., length]), np.array([0., 0.])),
                       np.array([0., -2501.]), np.array([length, 1.]))
src = np.array([0., -50.])
angles = torch.from_numpy(-(np.arange(n) + 1) * (0.999 / n)).cuda()
a = bench.product_step(rt, env, src, angles); b = bench.product_step(rt, env, src, angles); del a, b
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(steps):
    fan, hm = bench.product_step(rt, env, src, angles)
torch.cuda.synchronize()
print(f'{n} rays over the bathymetry cut: {1e3 * (time.perf_counter() - t0) / steps:.2f} ms per cast + heat map ({fan.passes}), '
      f'{fan.last_slow_rays} side rays')
