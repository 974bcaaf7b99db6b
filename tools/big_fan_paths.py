"""Product API on a big C2 fan: cast + heat map, stored fan (auto path) and store=False, against the three-kernel path."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, rt2d_loader, bench
rt = rt2d_loader.load()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
env = bench.make_env(rt)
source = np.array([0., 0.])
angles = -(np.arange(n) + 1) * (0.999 / n)
def run(**kw):
    hk = {k: kw.pop(k) for k in list(kw) if k in ('stream', 'fast')}
    sim = rt.Simulation2D(env, source)
    fan = sim.cast(angles, **bench.PROP, **kw)
    hm = sim.heatmap(resolution=list(bench.RES), cutoff=bench.CUTOFF, reduction_power=.5, as_tensor=True, **hk)
    return fan, hm
ref = None
variants = (('stored, three kernels', dict(stream=False)), ('stored, auto', dict()), ('store=False', dict(store=False)))
if os.environ.get('BIG_ONLY'):
    variants = [variants[int(os.environ['BIG_ONLY'])]]
for name, kw in variants:
    run(**kw); torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        fan, hm = run(**kw)
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    if ref is None: ref = hm
    d = (hm - ref).abs()
    rel = 0.0 if hm is ref else (d / ref.abs().clamp_min(1e-300))[torch.isfinite(ref) & (ref != 0)].max().item()
    print(f'{n} rays, {name}: {1e3 * min(ts):.2f} ms per cast+heat map (best of 3), slow rays {fan.last_slow_rays}, max rel diff vs three kernels {rel:.2e}, shape {tuple(hm.shape)}')
    del fan, hm
