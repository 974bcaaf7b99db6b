"""The product's heat map -- Simulation2D.cast + heatmap_full / heatmap with random cell sizes, frequencies, cut-offs and
powers -- against the oracle's (test infrastructure: simulation.py:108-138 restated) over the GPU's own points, on random
environments: shape, dropped-point count, every cell, the normalised map.  python tools/fuzz_product.py [first] [last]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import rt2d_loader, common
import test_gpu_random as t
from oracle import oracle as orc
orc.compile()
rt = rt2d_loader.load()
first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
last = int(sys.argv[2]) if len(sys.argv) > 2 else 100
bad = []
for seed in range(first, last):
    rng = np.random.default_rng(5000 + seed)
    env, source, target, angles, kw = t._random_case(rt, rng)
    kw['n_steps_max'] = min(kw['n_steps_max'], 2500)
    res = [int(v) for v in rng.choice([[50, 25], [20, 10], [100, 50], [10, 5], [7, 3], [33, 60]])]
    freq = float(rng.choice([1., 200., 3000., 15000.]))
    cutoff, power = float(rng.choice([.02, .1, .5])), float(rng.choice([.5, 1., .25]))
    store = bool(rng.integers(0, 2))
    try:
        oenv = common.oracle_env_from_product(orc, env)
        sim = rt.Simulation2D(env, source)
        sim.cast(angles, store=store, **kw)
        H, dropped = sim.heatmap_full(resolution=res, freq=freq)
        H = H.cpu().numpy()
        out = sim.heatmap(resolution=res, freq=freq, cutoff=cutoff, reduction_power=power)
        ref = rt.RayFan(env, source, angles, **kw)                 # (stored: the oracle walks the GPU's own points)
        rays = [r for r in common.oracle_rays_from_fan(orc, rt, ref) if r.status == 0]
        for r in rays:
            orc.post_ray(oenv, r)
        Href, plot_ref, drop_ref = orc.heatmap(oenv, rays, res=tuple(res), reduction_power=power, cutoff=cutoff, freq=freq)
        assert H.shape == Href.shape, (H.shape, Href.shape)
        assert int(dropped.item()) == drop_ref, (int(dropped.item()), drop_ref)
        assert np.array_equal(H == 0, Href == 0) and np.array_equal(np.isfinite(H), np.isfinite(Href)), 'occupancy'
        # (cells above 1e30 hold points whose 10^(G/10) is a SUBNORMAL double -- gains between -3000 and -3240 dB, ray.py:317
        # evaluated as written: the two pows -- where one unit in the last place is 2^-32 relative and the device's pow and
        # libm's may round differently: 2e-10 in 10^(G/10), 2e-11 in the value)
        m = np.isfinite(Href) & (Href != 0) & (np.abs(Href) < 1e29)
        rel = float((np.abs(H - Href)[m] / np.abs(Href[m])).max(initial=0))
        assert rel < 1e-11, rel
        m2 = np.isfinite(Href) & (np.abs(Href) >= 1e29)
        rel2 = float((np.abs(H - Href)[m2] / np.abs(Href[m2])).max(initial=0))
        assert rel2 < 1e-9, rel2
        assert out.shape == plot_ref.shape
        assert np.array_equal(np.isnan(out), np.isnan(plot_ref)), 'nan pattern of the normalised map'
        mm = ~np.isnan(plot_ref)
        d = float(np.abs(out - plot_ref)[mm].max(initial=0))
        assert d < 1e-9, d
        print(f'seed {seed}: store={store}, res {res}, freq {freq}, map {H.shape}, dropped {drop_ref}, cells to {rel:.1e}, normalised map to {d:.1e}', flush=True)
    except AssertionError as e:
        bad.append(seed); print(f'seed {seed}: MISMATCH {str(e)[:200]}', flush=True)
    except Exception as e:
        print(f'seed {seed}: {type(e).__name__}: {str(e)[:200]}', flush=True)
print(f'{last - first} seeds, {len(bad)} mismatches {bad}')
