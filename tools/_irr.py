import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, rt2d_loader, bench
rt = rt2d_loader.load()
n = 1 << 20
env = bench.make_env(rt)
dev = torch.device('cuda', 0)
angles = torch.from_numpy(-(np.arange(n) + 1) * (0.999 / n)).to(dev)
fan = rt.RayFan(env, np.array([0., 0.]), angles, store=False, **bench.PROP)
shape = (1503, 62)
m = fan.fused_mask(bench.RES, shape).cpu().numpy()
st = fan.status.cpu().numpy(); mono = fan.monotone.cpu().numpy(); sm = fan.slope_max.cpu().numpy(); npt = fan.n_pts.cpu().numpy()
stop = fan.stop_code.cpu().numpy(); nreb = fan.n_rebounds.cpu().numpy()
print('not fused', (m == 0).sum(), 'status!=0', (st != 0).sum(), 'non-mono', (mono == 0).sum(), 'slope>', (np.floor(sm * 2) + 2 > 8).sum())
idx = np.nonzero(m == 0)[0]
print('index range of irregular', idx.min(), idx.max(), 'contiguous runs', (np.diff(idx) > 1).sum() + 1)
print('n_pts all: mean', npt.mean(), 'max', npt.max(), ' irregular: mean', npt[idx].mean(), 'max', npt[idx].max())
print('stop codes irregular', np.bincount(stop[idx]), 'all', np.bincount(stop))
print('rebounds irregular mean', nreb[idx].mean(), 'all', nreb.mean())
h = np.histogram(idx, bins=16, range=(0, n))[0]
print('hist of irregular over the fan', h)
# time the side path alone
H = torch.zeros(shape[0] * shape[1], dtype=torch.float64, device=dev)
part = torch.from_numpy(idx).to(dev)
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    side = rt.RayFan(env, np.array([0., 0.]), angles[part], store=True, device=dev, **bench.PROP)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    side.heatmap_accumulate(H, bench.RES, shape, freq=1.0, fast=False)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f'side fan {len(idx)} rays: cast {1e3*(t1-t0):.2f} ms ({side.passes}), gain+bin {1e3*(t2-t1):.2f} ms, total pts {side.total}')
    del side
