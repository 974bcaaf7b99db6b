"""Gain of random fans at random sets of frequencies against the oracle (test infrastructure) evaluating the same points
and the same rebound log: python tools/fuzz_gain.py [first] [last].  Every gain kernel the launcher can pick is met (1, 2,
4 or 8 frequencies per lane, full and partial groups, two-phase and one-pass form)."""
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
    rng = np.random.default_rng(4000 + seed)
    env, source, target, angles, kw = t._random_case(rt, rng)
    kw['n_steps_max'] = min(kw['n_steps_max'], 2000)
    nf = int(rng.choice([1, 2, 3, 4, 5, 8, 11, 16]))
    freqs = [float(f) for f in np.geomspace(rng.uniform(1., 50.), rng.uniform(500., 20000.), nf)] if nf > 1 else [float(rng.choice([1., 1000.]))]
    two_phase = int(rng.integers(0, 2))
    os.environ['RT2D_GAIN_TWO_PHASE_MAX_RAYS'] = '1000000' if two_phase else '0'
    try:
        oenv = common.oracle_env_from_product(orc, env)
        fan = rt.RayFan(env, source, angles, **kw)
        fan.populate(*freqs)
        rays = common.oracle_rays_from_fan(orc, rt, fan)
        worst = 0.0
        for i in range(0, fan.n, 3):
            r = rays[i]
            if r.status != 0:
                continue
            orc.post_ray(oenv, r)
            for f in freqs:
                Go = orc.gain(oenv, r, f)
                Gh = fan.ray_slice(fan.G[f], i).cpu().numpy()[:r.steps]
                assert np.array_equal(np.isfinite(Gh), np.isfinite(Go)), (i, f)
                m = np.isfinite(Go)
                scale = max(1.0, float(np.abs(Go[m]).max(initial=0)))
                worst = max(worst, float(np.abs(Gh - Go)[m].max(initial=0)) / scale)
                ge = float(fan.G_end[f][i].item())
                assert (ge == Gh[-1]) or (np.isnan(ge) and np.isnan(Gh[-1])), (i, f, ge, Gh[-1])
        assert worst < 1e-13, worst
        print(f'seed {seed}: {nf} frequencies, {"two-phase" if two_phase else "one-pass"}, {fan.n} rays, worst error {worst:.1e} (relative to max |G|)', flush=True)
    except AssertionError as e:
        bad.append(seed); print(f'seed {seed}: MISMATCH {str(e)[:200]}', flush=True)
    except Exception as e:
        print(f'seed {seed}: {type(e).__name__}: {str(e)[:160]}', flush=True)
print(f'{last - first} seeds, {len(bad)} mismatches {bad}')
