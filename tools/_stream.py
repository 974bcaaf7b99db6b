import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, rt2d_loader, bench
rt = rt2d_loader.load()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
env = bench.make_env(rt)
dev = torch.device('cuda', 0)
angles = torch.from_numpy(-(np.arange(n) + 1) * (0.999 / (1 << 20))).to(dev) if len(sys.argv) > 2 else torch.from_numpy(-(np.arange(n) + 1) * (0.999 / n)).to(dev)
shape = (1503, 62)
H = torch.zeros(shape[0] * shape[1], dtype=torch.float64, device=dev)
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    fan = rt.RayFan(env, np.array([0., 0.]), angles, store=True, **bench.PROP)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    fan.heatmap_accumulate_stream(H, bench.RES, shape, freq=1.0)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f'{n} rays: cast {1e3*(t1-t0):.2f} ms ({fan.passes}), stream gain+bin {1e3*(t2-t1):.2f} ms, slow rays {fan.last_slow_rays}, total pts {fan.total}')
    del fan
