"""tools/fuzz_maps.py at the reference's default step sizes (dx_max 10, dz_max 0.1 ... 2, no rebound limit, up to 40 000
steps): stored / one-pass / fused maps of random environments must agree.  python tools/fuzz_maps_long.py [first] [last]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import rt2d_loader
import test_gpu_random as t
rt = rt2d_loader.load()
first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
last = int(sys.argv[2]) if len(sys.argv) > 2 else 30
bad = []
for seed in range(first, last):
    rng = np.random.default_rng(6000 + seed)
    env, source, target, angles, kw = t._random_case(rt, rng)
    kw.update(dx_max=10., dz_max=float(rng.choice([.1, .5, 2.])), n_rebounds_max=-1, n_steps_max=int(rng.choice([10000, 40000])))
    res = [int(v) for v in rng.choice([[50, 25], [20, 10], [10, 5]])]
    freq = float(rng.choice([1., 200., 3000.]))
    try:
        sims = {}
        for name, store in (('stored', True), ('fused', False)):
            sims[name] = rt.Simulation2D(env, source)
            fan = sims[name].cast(angles, store=store, **kw)
        H = {'stored': sims['stored'].heatmap_full(resolution=res, freq=freq, stream=False)[0].cpu().numpy(),
             'stream': sims['stored'].heatmap_full(resolution=res, freq=freq, stream=True)[0].cpu().numpy(),
             'fused': sims['fused'].heatmap_full(resolution=res, freq=freq)[0].cpu().numpy()}
        ref = H['stored']
        worst = 0.0
        for name in ('stream', 'fused'):
            assert H[name].shape == ref.shape
            assert np.array_equal(np.isfinite(H[name]), np.isfinite(ref)) and np.array_equal(H[name] == 0, ref == 0), name
            m = np.isfinite(ref) & (ref != 0)
            worst = max(worst, float((np.abs(H[name] - ref)[m] / np.abs(ref[m])).max(initial=0)))
        assert worst < 1e-11, worst
        print(f'seed {seed}: {kw["dz_max"]} m steps, {fan.ray_steps()} ray-steps, res {res}, redone the exact way: {bool(fan.__dict__.get("_exact_binning"))}, worst {worst:.1e}', flush=True)
    except AssertionError as e:
        bad.append(seed); print(f'seed {seed}: MISMATCH {str(e)[:200]}', flush=True)
    except Exception as e:
        print(f'seed {seed}: {type(e).__name__}: {str(e)[:200]}', flush=True)
print(f'{last - first} seeds, {len(bad)} mismatches {bad}')
