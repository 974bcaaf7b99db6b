"""EigenraySim2D on random environments: the sweep's selection against the reference's rule applied on the host to the same
distances (simulation.py:241-249: the rays within the n smallest DISTINCT distances, sorted by distance), and the batched
target products (rt2d_target_products) against the per-ray host loop Ray2D.gen_filter (ray.py:382-405).
python tools/fuzz_eigen.py [first] [last]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import rt2d_loader
import test_gpu_random as t
rt = rt2d_loader.load()
first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
last = int(sys.argv[2]) if len(sys.argv) > 2 else 60
bad = []
for seed in range(first, last):
    rng = np.random.default_rng(10000 + seed)
    env, source, target, angles, kw = t._random_case(rt, rng)
    lo, hi = np.asarray(env.range_min, dtype=float), np.asarray(env.range_max, dtype=float)
    tgt = np.array([rng.uniform(lo[0] + .3 * (hi[0] - lo[0]), hi[0]), rng.uniform(max(lo[1], -1500.) * .5, -5.)])
    n_keep, n_scan = int(rng.choice([5, 20, 60])), int(rng.choice([300, 1000, 3000]))
    nrm = int(rng.choice([3, 10, 20]))
    try:
        sim = rt.EigenraySim2D(env, source, tgt, n_keep, nrm, n_rays_scan=n_scan, scan_angle_min=-1.2, scan_angle_max=1.0,
                               dx_max=50., dz_max=25.)
        d = sim.scan_fan.host('dist'); a = sim.scan_fan._angles_host if sim.scan_fan._angles_host is not None else sim.scan_fan.angles_dev.cpu().numpy()
        okr = (sim.scan_fan.host('status') == 0) & ~np.isnan(d)
        dd = np.unique(d[okr])[:n_keep]
        sel = np.nonzero(okr & (d <= dd[-1]))[0] if len(dd) else np.array([], dtype=int)
        sel = sel[np.argsort(d[sel], kind='stable')]
        got = np.array([r.angle for r in sim.raypacks['scan'].rays])
        assert np.array_equal(got, a[sel]), (len(got), len(sel))
        if len(got):
            freqs = [1., 300., 8000.]
            a1, T1, M1 = sim.target_products(*freqs, batched=True)
            a0, T0, M0 = sim.target_products(*freqs, batched=False)
            assert np.array_equal(a1, a0) and M1.shape == M0.shape
            assert np.abs(T1 - T0).max() <= 1e-13 * np.abs(T0).max(), 'travel times'
            assert np.array_equal(M1 == 0, M0 == 0)
            nz = (M0 != 0) & np.isfinite(M0)
            assert (np.abs(M1[nz] - M0[nz]) / np.abs(M0[nz])).max(initial=0) < 1e-10, 'transmittances'
        print(f'seed {seed}: {n_scan} scanned, {len(got)} kept (asked {n_keep}), target {tgt.round(1).tolist()}', flush=True)
    except AssertionError as e:
        bad.append(seed); print(f'seed {seed}: MISMATCH {str(e)[:200]}', flush=True)
    except Exception as e:
        print(f'seed {seed}: {type(e).__name__}: {str(e)[:200]}', flush=True)
print(f'{last - first} seeds, {len(bad)} mismatches {bad}')
