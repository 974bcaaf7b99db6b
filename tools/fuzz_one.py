"""Details of one seed of tools/fuzz_maps.py: python tools/fuzz_one.py SEED"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import rt2d_loader
import test_gpu_random as t
rt = rt2d_loader.load()
seed = int(sys.argv[1])
rng = np.random.default_rng(2000 + seed)
env, source, target, angles, kw = t._random_case(rt, rng)
kw['n_steps_max'] = min(kw['n_steps_max'], 3000)
res = [int(v) for v in rng.choice([[50, 25], [20, 10], [100, 50], [10, 5]])]
freq = float(rng.choice([1., 200., 3000.]))
print('kw', kw, 'res', res, 'freq', freq, 'rays', len(angles))
sims = {}
for name, store in (('stored', True), ('fused', False)):
    sims[name] = rt.Simulation2D(env, source)
    sims[name].cast(angles, store=store, **kw)
fan = sims['stored'].raypacks[sims['stored'].pack_default].segments()[0][0]
print('status', np.unique(fan.host('status'), return_counts=True), 'stop', np.unique(fan.host('stop_code'), return_counts=True))
H = {'stored': sims['stored'].heatmap_full(resolution=res, freq=freq, stream=False)[0].cpu().numpy(),
     'general': sims['stored'].heatmap_full(resolution=res, freq=freq, stream=False, fast=False)[0].cpu().numpy(),
     'stream': sims['stored'].heatmap_full(resolution=res, freq=freq, stream=True)[0].cpu().numpy(),
     'fused': sims['fused'].heatmap_full(resolution=res, freq=freq)[0].cpu().numpy()}
ref = H['stored']
for name in ('general', 'stream', 'fused'):
    h = H[name]
    fin_diff = np.argwhere(np.isfinite(h) != np.isfinite(ref))
    zero_diff = np.argwhere((h == 0) != (ref == 0))
    m = np.isfinite(ref) & (ref != 0) & np.isfinite(h)
    rel = (np.abs(h - ref)[m] / np.abs(ref[m])).max(initial=0)
    print(name, 'finite pattern differs at', fin_diff[:5].tolist(), len(fin_diff), 'zero pattern differs at', zero_diff[:5].tolist(), len(zero_diff), 'max rel', rel)
    for c in fin_diff[:3].tolist() + zero_diff[:3].tolist():
        print('   cell', c, 'ref', ref[tuple(c)], name, h[tuple(c)])
    if rel > 1e-12:
        w = np.argwhere(m & (np.abs(h - ref) / np.maximum(np.abs(ref), 1e-300) > 1e-12))
        for c in w[:5].tolist():
            print('   cell', c, 'ref', ref[tuple(c)], name, h[tuple(c)])
G = fan.G[freq].cpu().numpy()
print('G range', np.nanmin(G), np.nanmax(G), 'inf count', np.isinf(G).sum(), 'nan', np.isnan(G).sum())
# which points fall into the cells that differ
bad = np.argwhere((H['stream'] == 0) != (ref == 0))
status = fan.host('status'); npts = fan.host('n_pts'); stop = fan.host('stop_code')
for c in bad[:3].tolist():
    for i in range(fan.n):
        X = fan.ray_slice(fan.X, i).cpu().numpy()[:npts[i]]; Z = fan.ray_slice(fan.Z, i).cpu().numpy()[:npts[i]]
        ix = np.trunc(X).astype(np.int64) // res[0]; iz = np.trunc(Z).astype(np.int64) // (-res[1])
        hit = np.nonzero((ix == c[0]) & (iz == c[1]))[0]
        if len(hit):
            print('cell', c, 'ray', i, 'angle', angles[i], 'points', hit.tolist(), 'of', npts[i], 'stop', stop[i], 'status', status[i])
            lo = max(0, hit[0] - 3); hi = min(npts[i], hit[-1] + 4)
            for s in range(lo, hi):
                print('    s', s, 'x', X[s], 'z', Z[s], 'ix', ix[s], 'iz', iz[s])
m = fan.fused_mask(res, H['stored'].shape)
print('fused_mask', np.unique(m.cpu().numpy(), return_counts=True))
