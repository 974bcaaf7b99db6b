"""Distance to a target two ways on random environments (ray.py:282-287): carried along by the stepping kernel (a fan cast
WITH the target) and from the stored trajectories afterwards (RayFan.dist_to, rt2d_dist_kernel) -- the same doubles, for
the cast's own target and for another one.  python tools/fuzz_dist.py [first] [last]"""
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
    rng = np.random.default_rng(9000 + seed)
    env, source, target, angles, kw = t._random_case(rt, rng)
    kw['n_steps_max'] = min(kw['n_steps_max'], 2500)
    lo, hi = np.asarray(env.range_min, dtype=float), np.asarray(env.range_max, dtype=float)
    t1 = np.array([rng.uniform(lo[0], hi[0]), rng.uniform(max(lo[1], -3000.), 0.)])
    t2 = np.array([rng.uniform(lo[0], hi[0]), rng.uniform(max(lo[1], -3000.), 0.)])
    try:
        f1 = rt.RayFan(env, source, angles, target=t1, **kw)
        f2 = rt.RayFan(env, source, angles, target=t2, store=False, **kw)
        ok = f1.host('status') == 0
        d11 = f1.dist_to(t1).cpu().numpy(); d12 = f1.dist_to(t2).cpu().numpy()
        assert np.array_equal(d11[ok], f1.host('dist')[ok], equal_nan=True), 'own target'
        assert np.array_equal(d12[ok], f2.host('dist')[ok], equal_nan=True), 'other target'
        print(f'seed {seed}: {int(ok.sum())} rays, {int(np.isnan(d11[ok]).sum())} / {int(np.isnan(d12[ok]).sum())} never reach the abscissa of the target', flush=True)
    except AssertionError as e:
        bad.append(seed); print(f'seed {seed}: MISMATCH {str(e)[:200]}', flush=True)
    except Exception as e:
        print(f'seed {seed}: {type(e).__name__}: {str(e)[:200]}', flush=True)
print(f'{last - first} seeds, {len(bad)} mismatches {bad}')
