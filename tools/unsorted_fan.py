"""A fan whose angles are NOT a monotonic sweep (shuffled): Simulation2D.cast sorts it before casting; against the same
angles cast in the order given (sort=False)."""
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
shuffled = np.random.default_rng(1).permutation(angles)
def run(a, **kw):
    sim = rt.Simulation2D(env, source)
    fan = sim.cast(a, **bench.PROP, **kw)
    hm = sim.heatmap(resolution=list(bench.RES), cutoff=bench.CUTOFF, reduction_power=.5, as_tensor=True)
    return sim, fan, hm
ref = None
for name, a, kw in (('in order', angles, {}), ('shuffled, cast as given', shuffled, dict(sort=False)), ('shuffled, sorted by cast()', shuffled, {})):
    run(a, **kw); torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        sim, fan, hm = run(a, **kw)
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    if ref is None: ref = hm
    rel = ((hm - ref).abs() / ref.abs().clamp_min(1e-300))[torch.isfinite(ref) & (ref != 0)].max().item()
    print(f'{n} rays, {name}: {1e3 * min(ts):.2f} ms per cast + heat map, max rel diff of the map {rel:.1e}')
    del sim, fan, hm
