"""GPU: angles that are not a monotonic sweep.  Simulation2D.cast sorts them on the device before casting (a wave of
unrelated rays takes the general step almost every time), the pack keeps the caller's order (ray.py:451-465: rays in the
order given, the first ray of every angle), the heat map is the same map."""

import numpy as np
import pytest

import common

pytestmark = pytest.mark.gpu


def test_shuffled_angles_are_cast_in_order_and_packed_as_given(rt):
    g = common.load('c2_fan999')
    env = common.product_env(rt, g)
    kw = common.prop_kwargs(g)
    angles = np.asarray(g['fan_angles'], dtype=np.float64)
    perm = np.random.default_rng(7).permutation(len(angles))
    shuffled = angles[perm]

    plain = rt.Simulation2D(env, g['fan_source'])
    fan_p = plain.cast(angles, **kw)
    mixed = rt.Simulation2D(env, g['fan_source'])
    fan_m = mixed.cast(shuffled, **kw)
    assert fan_p.order is None and fan_m.order is not None
    a_fan = fan_m.angles_dev.cpu().numpy()
    assert (np.diff(a_fan) > 0).all()                                     # cast as a sorted sweep
    assert np.array_equal(shuffled[fan_m.order.cpu().numpy()], a_fan)     # fan ray k came from position order[k]

    # heat map first (the one-fan path takes the permuted fan as it is) ...
    Hp = plain.heatmap(resolution=[50, 25], cutoff=.5)
    Hm = mixed.heatmap(resolution=[50, 25], cutoff=.5)
    assert Hp.shape == Hm.shape and np.nanmax(np.abs(Hp - Hm)) < 1e-12
    # ... then the pack: the caller's order, the same rays bit for bit
    pack_p, pack_m = plain.raypacks[plain.pack_default], mixed.raypacks[mixed.pack_default]
    assert pack_m.n_rays == pack_p.n_rays
    ok = fan_p.host('status') == 0
    want = [a for a in shuffled.tolist() if ok[np.nonzero(angles == a)[0][0]]]
    got = list(pack_m.rays.keys()) if hasattr(pack_m.rays, 'keys') else [r.angle for r in pack_m.rays]
    assert got == want
    by_angle = {float(a): i for i, a in enumerate(angles.tolist())}
    rays_m = pack_m.rays
    for a in want[::97]:
        ray_m = rays_m[a] if hasattr(rays_m, 'keys') else [r for r in rays_m if r.angle == a][0]
        ray_p = rt.Ray2D._from_fan(fan_p, by_angle[a])
        assert np.array_equal(ray_m.XZ, ray_p.XZ)
    assert np.array_equal(pack_m.range_min_plot, pack_p.range_min_plot) and np.array_equal(pack_m.range_max_plot, pack_p.range_max_plot)


def test_sort_can_be_switched_off(rt):
    g = common.load('c2_fan999')
    env = common.product_env(rt, g)
    kw = common.prop_kwargs(g)
    shuffled = np.random.default_rng(3).permutation(np.asarray(g['fan_angles'], dtype=np.float64))[:200]
    a = rt.Simulation2D(env, g['fan_source']); fa = a.cast(shuffled, sort=False, **kw)
    b = rt.Simulation2D(env, g['fan_source']); fb = b.cast(shuffled, **kw)
    assert fa.order is None and np.array_equal(fa.angles_dev.cpu().numpy(), shuffled)
    Ha, Hb = a.heatmap(resolution=[50, 25]), b.heatmap(resolution=[50, 25])
    assert np.nanmax(np.abs(Ha - Hb)) < 1e-12
    # the same rays either way: point counts by angle
    na = dict(zip(shuffled.tolist(), fa.host('n_pts').tolist()))
    nb = dict(zip(fb.angles_dev.cpu().numpy().tolist(), fb.host('n_pts').tolist()))
    assert na == nb
