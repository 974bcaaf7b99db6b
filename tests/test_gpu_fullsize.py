"""BASELINE's full size (1M rays, 20 rebounds) through size-independent properties: the oracle cannot follow there, so
(1) rays are independent: the C2 999-ray fan embedded in the big fan comes out bit for bit as on its own (and that
    small fan is pinned to the oracle and the reference by the other tests);
(2) the heat map is additive over rays: map(all) = map(even rays) + map(odd rays), to summation order -- a cell near the
    source takes a deposit of every one of the 2^20 rays, so the bound is n_rays * 2^-53 relative (a lost or doubled
    deposit is ~1 in absolute terms, the bound ~1e-4);
(3) the stored pipeline (padded single pass -> gain -> tile + general binning) and the fused kernel agree."""

import numpy as np
import pytest

import common

pytestmark = pytest.mark.gpu


def _bathymetry_env(rt):
    """The environment of bench.py's large_fan_bathymetry block: Map.from_asc of the synthetic grid, a 63 km cut as a floor
    table of 751 samples, source at -50 m (north_star's "1M-ray heat map over a bathymetry cut")."""
    import os
    bmap = rt.Map.from_asc(os.path.join(common.GOLD, 'synth_bathy.asc'))
    b0, b1 = (3.05, 41.7), (3.9, 41.75)
    length = float(bmap.cut_length(b0, b1))
    env = rt.Environment2D(bmap.floor_table(b0, b1, npoints=751), (np.array([0., length]), np.array([0., 0.])),
                           np.array([0., -2501.]), np.array([length, 1.]))
    return env, np.array([0., -50.])


@pytest.mark.parametrize('where_env', ['c2', 'bathymetry'])
def test_million_ray_fan_properties(rt, where_env):
    import torch
    g = common.load('c2_fan999' if where_env == 'c2' else 'c3_bathy999')
    kw = common.prop_kwargs(g)
    if where_env == 'c2':
        env = common.product_env(rt, g)
    else:
        env, src = _bathymetry_env(rt)
        g = dict(g); g['fan_source'] = src
    n = 1 << 20
    small = np.asarray(g['fan_angles'], dtype=np.float64)
    angles = -(np.arange(n) + 1) * (0.999 / n)
    where = np.linspace(0, n - 1, len(small)).astype(np.int64)            # the small fan's rays spread over the tiles
    angles[where] = small
    fan = rt.RayFan(env, g['fan_source'], angles, **kw)
    assert fan.passes == 'padded'
    ref = rt.RayFan(env, g['fan_source'], small, **kw)
    idx = torch.as_tensor(where, device=fan.device)
    for name in ('n_pts', 'stop_code', 'n_rebounds', 'status'):
        assert torch.equal(getattr(fan, name)[idx], getattr(ref, name)), name
    assert torch.equal(fan.range[:, idx], ref.range)
    for j in (0, 17, 500, 998):
        a, b = rt.Ray2D._from_fan(fan, int(where[j])), rt.Ray2D._from_fan(ref, j)
        assert np.array_equal(a.XZ, b.XZ)
    ok = fan.status == 0
    steps = int((fan.n_pts[ok] - 1).sum().item())
    assert steps > 8e8
    print(f'{where_env}: {steps} ray-steps, {int((~ok).sum().item())} rays with an error status')

    # one shape for all maps: the whole fan's plot range
    sim = rt.Simulation2D(env, g['fan_source'])
    sim.raypacks['main'].add_fan(fan)
    pack = sim.raypacks['main']
    res = np.array([50, 25])
    shape = np.ceil(pack.size_plot / res).astype(int) + 1
    def bin_rays(mask):
        H = torch.zeros(int(shape[0]) * int(shape[1]), dtype=torch.float64, device=fan.device)
        fan.heatmap_accumulate(H, res, shape, freq=1, ray_mask=mask)
        return H
    ok8 = ok.to(torch.uint8)
    even = ok8.clone(); even[1::2] = 0
    odd = ok8.clone(); odd[0::2] = 0
    H_all, H_even, H_odd = bin_rays(ok8), bin_rays(even), bin_rays(odd)
    assert float(H_all.max().item()) > 1e3
    tol = n * 2.0 ** -53 * float(H_all.max().item())        # 1.2e-10 relative: summation order only; one deposit is ~1
    assert tol < 1e-3
    assert float((H_all - (H_even + H_odd)).abs().max().item()) <= tol
    assert torch.equal(H_all == 0, (H_even + H_odd) == 0)

    # gain + binning in one pass over the stored fan (rt2d_gain_heatmap_rest: the lean kernel, its general block, the side rays)
    H_s = torch.zeros_like(H_all)
    fan.heatmap_accumulate(H_s, res, shape, freq=1000., ray_mask=ok8, stream=True)
    H_3 = torch.zeros_like(H_all)
    fan.heatmap_accumulate(H_3, res, shape, freq=1000., ray_mask=ok8, stream=False)
    assert int(fan.last_fused_violations.item()) == 0
    fin = torch.isfinite(H_3)
    assert torch.equal(torch.isfinite(H_s), fin) and torch.equal(H_s == 0, H_3 == 0)
    assert float((H_s - H_3)[fin].abs().max().item()) <= n * 2.0 ** -53 * float(H_3[fin].max().item())

    # fused kernel (nothing stored) on the same rays
    lean = rt.RayFan(env, g['fan_source'], angles, store=False, **kw)
    assert torch.equal(lean.n_pts, fan.n_pts) and torch.equal(lean.stop_code, fan.stop_code)
    H_f = torch.zeros_like(H_all)
    lean.heatmap_accumulate(H_f, res, shape, freq=1)
    assert int(lean.last_fused_violations.item()) == 0
    assert float((H_f - H_all).abs().max().item()) <= tol


def test_mid_size_fan_tile_binner_equals_general_binner(rt):
    """20 000 rays: few tiles, so the tile binner shares each tile's rows among several waves and hands the rays marked
    "known not to qualify" (ray_mask bit 4) to the general kernel unchecked; the general kernel alone must give the same
    map, with and without the caller's hints."""
    g = common.load('c2_fan999')
    env = common.product_env(rt, g)
    n = 20000
    angles = -(np.arange(n) + 1) * (0.999 / n)
    sim = rt.Simulation2D(env, g['fan_source'])
    fan = sim.cast(angles, **common.prop_kwargs(g))
    for freq in (1., 12000.):
        Hg, _ = sim.heatmap_full(resolution=[50, 25], freq=freq, fast=False)
        for hint in (True, False):
            Ht, _ = sim.heatmap_full(resolution=[50, 25], freq=freq, fast=True, hint=hint)
            assert 0 < fan.last_slow_rays < n // 5
            Ht_, Hg_ = Ht.cpu().numpy(), Hg.cpu().numpy()
            assert np.array_equal(np.isfinite(Ht_), np.isfinite(Hg_)) and np.array_equal(Ht_ == 0, Hg_ == 0)
            fin = np.isfinite(Hg_)
            assert np.abs(Ht_[fin] - Hg_[fin]).max() <= 1e-12 * np.abs(Hg_[fin]).max()
