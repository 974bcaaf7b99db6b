"""GPU: sweep + refine (EigenraySim2D scan / refine) against the reference's own run (tests/golden/c5_eigen.npz)."""

import numpy as np
import pytest

import common

pytestmark = pytest.mark.gpu


def test_scan_and_refine_vs_reference(rt):
    g = common.load('c5_eigen')
    env = common.product_env(rt, g)
    sim = rt.EigenraySim2D(env, g['source'], g['target'], int(g['n_rays']), int(g['n_rebounds_max']),
                           n_rays_scan=int(g['n_rays_scan']), scan_angle_min=float(g['scan_angle_min']),
                           scan_angle_max=float(g['scan_angle_max']), dx_max=float(g['dx_max']), dz_max=float(g['dz_max']))
    # the sweep: every scanning ray's distance to the target
    d = sim.scan_fan.host('dist')
    ref = g['scan_dist']
    assert np.array_equal(np.isnan(d), np.isnan(ref))
    m = ~np.isnan(ref)
    knife = sim.scan_fan.host('margin') <= common.KNIFE_EDGE_M
    assert np.max(np.abs(d - ref)[m & ~knife]) < 1e-6
    # selection: same angles, same order (sorted by distance)
    sel = np.array([r.angle for r in sim.raypacks['scan'].rays])
    assert np.array_equal(sel, g['scan_selected_angles'])
    assert abs(sim.dist_avg - float(g['scan_dist_avg'])) < 1e-6
    sim.refine(iterations=2, n_rays=5)
    assert sim.n_refines == int(g['n_refines'])
    for it in range(1, sim.n_refines + 1):
        got = np.array([r.angle for r in sim.raypacks[f'refine-{it}'].rays])
        assert np.allclose(got, g[f'refine{it}_angles'], rtol=0, atol=1e-15), it
        gd = np.array([r.dist_to_target for r in sim.raypacks[f'refine-{it}'].rays])
        assert np.max(np.abs(gd - g[f'refine{it}_dist'])) < 1e-6
    assert abs(sim.dist_avg - float(g['final_dist_avg'])) < 1e-6


def test_eigenray_filter_vs_reference(rt):
    """per-ray target products (Ray2D.gen_filter) and the filtered signal of get_filter_NEW against the
    reference's own run on a synthetic signal (tests/golden/f5_filter.npz, made by oracle/gen_golden.py f5)."""
    g = common.load('f5_filter')
    env = common.product_env(rt, g)
    sim = rt.EigenraySim2D(env, g['source'], g['target'], int(g['n_rays']), int(g['n_rebounds_max']),
                           n_rays_scan=int(g['n_rays_scan']), scan_angle_min=float(g['scan_angle_min']),
                           scan_angle_max=float(g['scan_angle_max']), dx_max=float(g['dx_max']), dz_max=float(g['dz_max']))
    freqs = [float(f) for f in g['freqs']]
    angles, T_t, Tm_t = sim.target_products(*freqs)
    assert np.array_equal(angles, g['angles'])
    assert np.abs(T_t - g['T_target']).max() < 1e-9
    ref = g['Tmult_target']
    assert np.array_equal(Tm_t == 0, ref == 0)                       # underflow at the high frequencies, as in the reference
    assert (np.abs(Tm_t - ref)[ref != 0] / ref[ref != 0]).max() < 1e-9
    filt = sim.get_filter_NEW(*freqs)
    sr, out = filt(float(g['sample_rate']), g['signal'])
    assert sr == float(g['sample_rate']) and out.shape == g['filtered'].shape
    assert np.abs(out - g['filtered']).max() < 1e-9 * np.abs(g['filtered']).max()


@pytest.mark.gpu
def test_target_products_batched_equals_the_per_ray_loop(rt):
    """rt2d_target_products (one wave per ray: closest stored point, residual time and gain to the target)
    against the host loop over Ray2D.gen_filter, which is the reference's own sequence (ray.py:382-405) and is pinned by
    the tests_2.ipynb numbers: same closest steps, same travel times, transmittances to 1e-11."""
    g = common.load('c2_fan999')
    env = common.product_env(rt, g)
    target = np.array([30000., -400.])
    sim = rt.EigenraySim2D(env, g['fan_source'], target, 40, 20, n_rays_scan=4000, scan_angle_min=-0.9, scan_angle_max=-0.01,
                           dx_max=50., dz_max=25.)
    freqs = [1., 250., 5000., 20000.]
    a1, T1, M1 = sim.target_products(*freqs, batched=True)
    a0, T0, M0 = sim.target_products(*freqs, batched=False)
    assert len(a1) >= 40 and np.array_equal(a1, a0)
    assert M1.shape == M0.shape == (len(a1), len(freqs))
    assert np.abs(T1 - T0).max() <= 1e-13 * np.abs(T0).max()
    assert np.array_equal(M1 == 0, M0 == 0)
    nz = M0 != 0
    assert (np.abs(M1[nz] - M0[nz]) / M0[nz]).max() < 1e-11
    # the closest step itself, for a few rays
    pack = sim.raypacks[sim.pack_default]
    for ray in list(pack.rays)[:5]:
        s = ray.closest_step_to_point(target)
        T_host, _ = ray.gen_filter(target)
        k = int(np.nonzero(a1 == ray.angle)[0][0])
        assert abs(T1[k] - T_host) <= 1e-13 * abs(T_host), (s, T1[k], T_host)


@pytest.mark.gpu
def test_sweep_of_a_million_rays_selects_like_a_host_sort(rt):
    """BASELINE config 5 at its stated size: n_rays_scan = 1 000 000 over the C2 environment, 1000 rays kept.  The device
    selection (distinct-distance threshold + stable sort, simulation.py:241-249's rule) against NumPy on the same
    distances, ties included; then one refine iteration keeps every parent or a ray of its cone that is closer."""
    g = common.load('c2_fan999')
    env = common.product_env(rt, g)
    target = np.array([40000., -600.])
    n_scan, n_keep = 1_000_000, 1000
    sim = rt.EigenraySim2D(env, g['fan_source'], target, n_keep, 20, n_rays_scan=n_scan, scan_angle_min=-0.999,
                           scan_angle_max=-0.001, dx_max=50., dz_max=25.)
    assert sim.scan_fan.n == n_scan
    d = sim.scan_fan.dist.cpu().numpy().copy()
    d[sim.scan_fan.host('status') != 0] = np.nan
    angles = np.linspace(-0.999, -0.001, n_scan)
    kth = np.unique(d[~np.isnan(d)])[n_keep - 1]
    sel = np.nonzero(d <= kth)[0]
    sel = sel[np.argsort(d[sel], kind='stable')]
    got = np.array([r.angle for r in sim.raypacks['scan'].rays])
    assert len(got) >= n_keep and np.array_equal(got, angles[sel])
    got_d = np.array([r.dist_to_target for r in sim.raypacks['scan'].rays])
    assert np.array_equal(got_d, d[sel])                       # the re-cast with trajectories reproduces the sweep's rays
    # one refine iteration (simulation.py:255-314): every kept ray comes from the cone of a parent (or is the parent), at most
    # one ray per angle; like the reference, refine casts WITHOUT the rebound limit, so a kept ray may be farther from the
    # target than its parent was.  The winners carry scalars only until their trajectories are asked for.
    cone = sim._angular_precision
    sim.refine(iterations=1, n_rays=5, force_refine=True)
    pack = sim.raypacks[sim.pack_default]
    kept = np.array([r.angle for r in pack.rays])
    assert 0 < len(kept) <= len(got) and len(np.unique(kept)) == len(kept)
    assert np.all(np.min(np.abs(kept[:, None] - got[None, :]), axis=1) <= cone * (1 + 1e-6) + 1e-15)
    ray = next(r for r in pack.rays if r._fan.passes == 'subset')
    d_before = ray.dist_to_target
    xz = ray.XZ                                                # casts the winners' trajectories now
    assert xz.shape == (ray.steps, 2) and ray._fan.store and ray._fan.passes == 'count+store'
    assert float(ray._fan.host('dist')[ray._i]) == d_before    # the second cast of the angle is the same ray
