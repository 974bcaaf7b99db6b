"""GPU: randomised parity.  Seeded random environments (sea floors with relief, flat or rippled surfaces), sources,
targets and propagation parameters; the HIP kernels against the oracle reading the SAME tables and start values: every
integer and every stored double must be identical.  This is the net under the fast paths of the stepping kernel (constant
surface instance, skipped step division, two-FMA quotient, one-predicate common step), which only exist because they are
claimed to change no bit."""

import numpy as np
import pytest

import common

pytestmark = pytest.mark.gpu


def _random_case(rt, rng):
    length = float(rng.choice([8000., 20000., 60000.]))
    depth = float(rng.uniform(300., 3000.))
    nx = int(rng.integers(40, 400))
    xs = np.linspace(0., length, nx)
    relief = rng.uniform(0., 0.35) * depth
    floor = -depth + relief * np.sin(xs / rng.uniform(500., 5000.) + rng.uniform(0, 6.28)) * rng.uniform(0.2, 1.) \
        + rng.normal(scale=rng.uniform(0., 3.), size=nx)
    if rng.random() < 0.5:
        ceil_x, ceil_z = np.array([0., length]), np.array([0., 0.])                     # flat sea surface (CEILC kernels)
    else:
        ceil_x = np.linspace(0., length, int(rng.integers(20, 200)))
        ceil_z = -1.0 + rng.uniform(0.1, 1.) * np.cos(ceil_x / rng.uniform(50., 800.))  # swell (general kernels)
    zmin = float(np.floor(floor.min())) - 1.
    env = rt.Environment2D((xs, floor), (ceil_x, ceil_z), np.array([0., zmin]), np.array([length, 1.]),
                           res_x=int(rng.choice([50, 100, 200])), res_z=int(rng.choice([5, 10, 20])))
    source = np.array([float(rng.uniform(0., 0.2 * length)), float(rng.uniform(0.9 * floor.max(), -2.))])
    kw = dict(dx_max=float(rng.choice([10., 25., 50., 80.])), dz_max=float(rng.choice([0.5, 5., 25., 40.])),
              n_rebounds_max=int(rng.choice([-1, 0, 3, 12])), backprop=bool(rng.random() < 0.7),
              n_steps_max=int(rng.choice([300, 3000, 100000])))
    target = np.array([float(rng.uniform(0.3, 0.9) * length), float(rng.uniform(0.8 * floor.max(), -5.))]) if rng.random() < 0.5 else None
    angles = np.sort(rng.uniform(-1.3, 1.3, 48))
    angles[0] = 0.0                                                                      # a horizontal launch: slope exactly 0
    return env, source, target, angles, kw


@pytest.mark.parametrize('seed', range(12))
def test_random_environment_bit_exact(rt, orc, seed):
    rng = np.random.default_rng(1000 + seed)
    env, source, target, angles, kw = _random_case(rt, rng)
    oenv = common.oracle_env_from_product(orc, env)
    init = common.host_init(float(env.penv.calc_c(0.)), angles)
    fan = rt.RayFan(env, source, angles, target=target, init=init, **kw)
    fan.validate()            # (a fan whose rebound log overflowed is cast again with an exact one: the oracle gets the same room)
    p = orc.make_params(target=target, rebound_cap=max(64, int(fan.rebound_cap)), **kw)
    rays = [orc.cast_ray(oenv, source, a, p, init=(init[0][i], init[1][i])) for i, a in enumerate(angles)]
    assert np.array_equal(fan.host('status'), [r.status for r in rays])
    assert np.array_equal(fan.host('n_pts'), [r.steps for r in rays])
    assert np.array_equal(fan.host('stop_code'), [r.stop_code for r in rays])
    assert np.array_equal(fan.host('n_rebounds'), [r.n_rebounds for r in rays])
    assert np.array_equal(fan.host('T_end'), [r.T[-1] for r in rays])
    assert np.array_equal(fan.host('margin'), [r.margin for r in rays])
    if target is not None:
        np.testing.assert_array_equal(fan.host('dist'), [r.dist_to_target for r in rays])
    for i, r in enumerate(rays):
        assert np.array_equal(rt.Ray2D._from_fan(fan, i).XZ, r.XZ), f'seed {seed} ray {i}'
        nl = min(len(r.reb_step), fan.rebound_cap)
        assert np.array_equal(fan.host('reb_step')[:nl, i], r.reb_step[:nl])
        assert np.array_equal(fan.host('reb_angle')[:nl, i], r.reb_angle[:nl])
    stops = sorted(set(fan.host('stop_code').tolist()))
    print(f'seed {seed}: stop codes {stops}, {int(fan.host("n_rebounds").sum())} rebounds, {fan.ray_steps()} ray-steps, '
          f'{int((fan.host("status") != 0).sum())} ray errors, kw {kw}')


@pytest.mark.parametrize('seed', range(9))      # (seed 8: the clipped last point of a ray jumps eight rows inside its column)
def test_random_environment_heat_maps_agree(rt, seed):
    """The three ways to a heat map -- stored trajectories + gain + binning kernels, one pass over stored trajectories,
    fused cast+gain+binning -- on random environments, cell sizes and frequencies: same map to 1e-12."""
    rng = np.random.default_rng(2000 + seed)
    env, source, target, angles, kw = _random_case(rt, rng)
    kw['n_steps_max'] = min(kw['n_steps_max'], 3000)
    res = [int(v) for v in rng.choice([[50, 25], [20, 10], [100, 50], [10, 5]])]
    freq = float(rng.choice([1., 200., 3000.]))
    sims = {}
    for name, store in (('stored', True), ('fused', False)):
        sims[name] = rt.Simulation2D(env, source)
        sims[name].cast(angles, store=store, **kw)
    H = {'stored': sims['stored'].heatmap_full(resolution=res, freq=freq, stream=False)[0].cpu().numpy(),
         'stream': sims['stored'].heatmap_full(resolution=res, freq=freq, stream=True)[0].cpu().numpy(),
         'fused': sims['fused'].heatmap_full(resolution=res, freq=freq)[0].cpu().numpy()}
    ref = H['stored']
    for name in ('stream', 'fused'):
        assert H[name].shape == ref.shape
        assert np.array_equal(np.isfinite(H[name]), np.isfinite(ref)) and np.array_equal(H[name] == 0, ref == 0), name
        m = np.isfinite(ref) & (ref != 0)
        assert (np.abs(H[name] - ref)[m] / np.abs(ref[m])).max(initial=0) < 1e-12, name
    print(f'seed {seed}: res {res}, freq {freq}, map {ref.shape}, {int((ref != 0).sum())} cells hit')


def test_a_ray_the_masks_let_through_behind_a_pending_cast(rt, monkeypatch):
    """Seed 8 again (the clipped last point of a ray jumps eight rows inside its column, which the slope bound of the masks
    does not see), through the form of the one-pass map that is queued behind a cast whose summary has not been read: the
    rays are sorted out on the device (rt2d_classify_rays), the kernel counts the ray it could not bin, and the map is
    done again the exact way once the summary is read."""
    from ray_tracer_amd import rays as rays_mod
    rng = np.random.default_rng(2000 + 8)
    env, source, target, angles, kw = _random_case(rt, rng)
    kw['n_steps_max'] = min(kw['n_steps_max'], 3000)
    res = [int(v) for v in rng.choice([[50, 25], [20, 10], [100, 50], [10, 5]])]
    freq = float(rng.choice([1., 200., 3000.]))
    ref = rt.Simulation2D(env, source)
    ref.cast(angles, **kw)
    Hs = ref.heatmap_full(resolution=res, freq=freq, stream=False)[0].cpu().numpy()
    monkeypatch.setattr(rays_mod, 'STREAM_MIN_RAYS', 1)
    sim = rt.Simulation2D(env, source)
    fan = sim.cast(angles, **kw)
    assert fan._pending
    H = sim.heatmap_full(resolution=res, freq=freq, small=False)[0].cpu().numpy()
    assert int(fan.last_fused_violations.item()) > 0 and fan.__dict__.get('_exact_binning') is True
    assert H.shape == Hs.shape and np.array_equal(H == 0, Hs == 0)
    m = Hs != 0
    assert (np.abs(H - Hs)[m] / np.abs(Hs[m])).max() < 1e-12
