"""The 1 M-ray C2 fan through the product API (Simulation2D.cast + heatmap), a few steps: for kernel traces
(bash tools/trace_script.sh TAG tools/product_large.py) and timelines.  python tools/product_large.py [rays] [steps]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, rt2d_loader, bench
rt = rt2d_loader.load()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
env = bench.make_env(rt)
src = np.array([0., 0.])
angles = torch.from_numpy(-(np.arange(n) + 1) * (0.999 / n)).cuda()
a = bench.product_step(rt, env, src, angles); b = bench.product_step(rt, env, src, angles); del a, b
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(steps):
    fan, hm = bench.product_step(rt, env, src, angles)
torch.cuda.synchronize()
print(f'{n} rays: {1e3 * (time.perf_counter() - t0) / steps:.2f} ms per cast + heat map ({fan.passes})')
