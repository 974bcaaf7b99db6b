"""Several frequencies at once against one map per frequency on random environments (tests/test_gpu_random.py's generator):
python tools/fuzz_multi.py [first] [last].  Simulation2D.heatmaps(freqs) -- gains of all frequencies from one launch, all
maps binned in one pass (map-minor scratch) -- must give the maps of heatmap(freq=f), to the order of the atomics."""
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
    rng = np.random.default_rng(3000 + seed)
    env, source, target, angles, kw = t._random_case(rt, rng)
    kw['n_steps_max'] = min(kw['n_steps_max'], 3000)
    res = [int(v) for v in rng.choice([[50, 25], [20, 10], [100, 50], [10, 5]])]
    nf = int(rng.choice([2, 3, 4, 5, 8, 9, 16]))
    freqs = [float(f) for f in np.geomspace(rng.uniform(1., 50.), rng.uniform(500., 20000.), nf)]
    try:
        a = rt.Simulation2D(env, source); a.cast(angles, **kw)
        b = rt.Simulation2D(env, source); b.cast(angles, **kw)
        many = a.heatmaps(freqs, resolution=res)
        worst = 0.0
        for f in freqs:
            one = b.heatmap(freq=f, resolution=res, stream=False)
            m = many[f]
            assert m.shape == one.shape
            assert np.array_equal(np.isfinite(m), np.isfinite(one)) and np.array_equal(m == 0, one == 0)
            fin = np.isfinite(one) & (one != 0)
            worst = max(worst, float((np.abs(m - one)[fin] / np.abs(one[fin])).max(initial=0)))
        assert worst < 1e-11, worst
        print(f'seed {seed}: {nf} maps {one.shape}, res {res}, worst relative difference {worst:.1e}', flush=True)
    except AssertionError as e:
        bad.append(seed); print(f'seed {seed}: MISMATCH {str(e)[:200]}', flush=True)
    except Exception as e:
        print(f'seed {seed}: {type(e).__name__}: {str(e)[:160]}', flush=True)
print(f'{last - first} seeds, {len(bad)} mismatches {bad}')
