"""Packs built from several casts (ray.py:441-472: a pack keeps the FIRST ray of every angle, in the order given; its plot
range folds every ray in) on random environments: two overlapping, unsorted angle sets cast one after the other into one
pack must give the heat map of ONE cast of the distinct angles.  python tools/fuzz_packs.py [first] [last]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import rt2d_loader
import test_gpu_random as t
rt = rt2d_loader.load()
first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
last = int(sys.argv[2]) if len(sys.argv) > 2 else 100
bad = []
for seed in range(first, last):
    rng = np.random.default_rng(8000 + seed)
    env, source, target, angles, kw = t._random_case(rt, rng)
    kw['n_steps_max'] = min(kw['n_steps_max'], 2000)
    angles = np.asarray(angles, dtype=np.float64)
    fine = np.unique(np.concatenate([angles, np.linspace(angles.min(), angles.max(), int(rng.integers(70, 400)))]))
    A = rng.permutation(fine)[:int(len(fine) * rng.uniform(.3, .9))]
    B = rng.permutation(fine)[:int(len(fine) * rng.uniform(.3, .9))]
    if rng.random() < .5:
        A, B = np.sort(A), np.sort(B)[::-1]
    res = [int(v) for v in rng.choice([[50, 25], [20, 10], [10, 5]])]
    freq = float(rng.choice([1., 500.]))
    try:
        two = rt.Simulation2D(env, source)
        two.cast(A, **kw); two.cast(B, **kw)
        seen, distinct = set(), []
        for a in np.concatenate([A, B]):
            if a not in seen:
                seen.add(a); distinct.append(a)
        one = rt.Simulation2D(env, source)
        one.cast(np.array(distinct), **kw)
        p2, p1 = two.raypacks[two.pack_default], one.raypacks[one.pack_default]
        assert p2.n_rays == p1.n_rays, (p2.n_rays, p1.n_rays)
        assert np.array_equal(p2.range_min_plot, p1.range_min_plot) and np.array_equal(p2.range_max_plot, p1.range_max_plot), 'plot range'
        H2 = two.heatmap_full(resolution=res, freq=freq)[0].cpu().numpy()
        H1 = one.heatmap_full(resolution=res, freq=freq)[0].cpu().numpy()
        assert H2.shape == H1.shape and np.array_equal(H2 == 0, H1 == 0) and np.array_equal(np.isfinite(H2), np.isfinite(H1)), 'occupancy'
        m = np.isfinite(H1) & (H1 != 0)
        rel = float((np.abs(H2 - H1)[m] / np.abs(H1[m])).max(initial=0))
        assert rel < 1e-11, rel
        n2, n1 = two.heatmap(resolution=res, freq=freq), one.heatmap(resolution=res, freq=freq)
        assert np.array_equal(np.isnan(n2), np.isnan(n1)) and float(np.nanmax(np.abs(n2 - n1), initial=0)) < 1e-9
        print(f'seed {seed}: {len(A)} + {len(B)} angles, {p1.n_rays} rays in the pack, map {H1.shape}, cells to {rel:.1e}', flush=True)
    except AssertionError as e:
        bad.append(seed); print(f'seed {seed}: MISMATCH {str(e)[:200]}', flush=True)
    except Exception as e:
        print(f'seed {seed}: {type(e).__name__}: {str(e)[:200]}', flush=True)
print(f'{last - first} seeds, {len(bad)} mismatches {bad}')
