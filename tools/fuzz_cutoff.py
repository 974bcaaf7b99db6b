"""The opt-in power cut-off (not in the reference; the rule is the oracle's: truncate the un-truncated ray after the first
point with 10^(G/10) < cutoff) on random environments, cut-offs and frequencies: point counts, stop codes, rebound counts,
status and every stored point of every ray whose decision is not within 1e-9 dB of the threshold.
python tools/fuzz_cutoff.py [first] [last]"""
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
    rng = np.random.default_rng(7000 + seed)
    env, source, target, angles, kw = t._random_case(rt, rng)
    kw['n_steps_max'] = min(kw['n_steps_max'], 2500)
    cutoff = float(rng.choice([.5, .1, 1e-3, 1e-9]))
    freq = float(rng.choice([1., 1000., 5000., 20000.]))
    try:
        oenv = common.oracle_env_from_product(orc, env)
        init = common.host_init(float(env.penv.calc_c(0.)), angles)
        fan = rt.RayFan(env, source, angles, target=target, init=init, power_cutoff=cutoff, cutoff_freq=freq, **kw)
        lean = rt.RayFan(env, source, angles, init=init, store=False, power_cutoff=cutoff, cutoff_freq=freq, **kw)
        fan.validate()
        assert np.array_equal(lean.host('n_pts'), fan.host('n_pts')) and np.array_equal(lean.host('stop_code'), fan.host('stop_code'))
        p = orc.make_params(target=target, rebound_cap=max(64, int(fan.rebound_cap), 4096), **kw)
        n_cut = n_checked = 0
        for i, a in enumerate(angles):
            r0 = orc.cast_ray(oenv, source, a, p, init=(init[0][i], init[1][i]))
            r, j, db = orc.apply_power_cutoff(oenv, r0, cutoff, freq, target=target)
            if not db > 1e-9:
                continue
            n_checked += 1
            n_cut += r.stop_code == 8
            assert fan.host('n_pts')[i] == r.steps and fan.host('stop_code')[i] == r.stop_code, f'ray {i}: {fan.host("n_pts")[i]} / {r.steps} points, stop {fan.host("stop_code")[i]} / {r.stop_code}'
            assert fan.host('n_rebounds')[i] == r.n_rebounds and fan.host('status')[i] == r.status, f'ray {i}: rebounds / status'
            ray = rt.Ray2D._from_fan(fan, i)
            assert np.array_equal(ray.XZ, r.XZ), f'ray {i}: points'
        print(f'seed {seed}: cutoff {cutoff} at {freq} Hz: {n_cut} of {n_checked} checked rays end on the power cut-off ({fan.n} rays)', flush=True)
    except AssertionError as e:
        bad.append(seed); print(f'seed {seed}: MISMATCH {str(e)[:200]}', flush=True)
    except Exception as e:
        print(f'seed {seed}: {type(e).__name__}: {str(e)[:200]}', flush=True)
print(f'{last - first} seeds, {len(bad)} mismatches {bad}')
