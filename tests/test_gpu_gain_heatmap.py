"""GPU parity of the gain kernel (rt2d_gain), the post-pass (rt2d_post), rt2d_dist and the heat-map kernels."""

import ctypes as C

import numpy as np
import pytest

import common
from common import SCALAR_RTOL, KNIFE_EDGE_M

pytestmark = pytest.mark.gpu

GAIN_RTOL = 1e-9      # cumulative gain [dB] relative tolerance vs the reference
HEAT_RTOL = 1e-9      # heat-map cell tolerance (relative to the cell value)


def _cast(rt, g, angles=None, **over):
    env = common.product_env(rt, g)
    kw = common.prop_kwargs(g)
    kw.update(over)
    fan = rt.RayFan(env, g['fan_source'], g['fan_angles'] if angles is None else angles, target=common.target_of(g), **kw)
    return fan, env


def _oracle_rays_from_fan(orc, rt, fan):
    return common.oracle_rays_from_fan(orc, rt, fan)


def _oracle_post(orc, oenv, r, target=None):
    return orc.post_ray(oenv, r, target)


def test_post_arrays_bit_exact(rt, orc):
    """dL, L, C, dT, T of rt2d_post equal the oracle's bit for bit on the same points (ray.py:271-275)."""
    g = common.load('c3_bathy999')
    fan, env = _cast(rt, g)
    oenv = common.oracle_env_from_product(orc, env)
    rays = _oracle_rays_from_fan(orc, rt, fan)
    for i in range(0, fan.n, 7):
        _oracle_post(orc, oenv, rays[i])
        ray = rt.Ray2D._from_fan(fan, i)
        assert np.array_equal(ray.dL, rays[i].dL_full[1:])
        assert np.array_equal(ray.L, rays[i].L)
        assert np.array_equal(ray.C, rays[i].C)
        assert np.array_equal(ray.dT, rays[i].dT_full[1:])
        assert np.array_equal(ray.T, rays[i].T)
        assert ray.T[-1] == fan.host('T_end')[i] and ray.L[-1] == fan.host('L_end')[i]


def test_dist_kernel(rt, orc):
    """rt2d_dist on stored trajectories == the in-kernel distance == the oracle's sorted interpolation."""
    g = common.load('c1_flat250')
    fan, env = _cast(rt, g)
    oenv = common.oracle_env_from_product(orc, env)
    d2 = fan.dist_to(g['fan_target']).cpu().numpy()
    np.testing.assert_array_equal(d2, fan.host('dist'))
    rays = _oracle_rays_from_fan(orc, rt, fan)
    for i in range(0, fan.n, 5):
        d = _oracle_post(orc, oenv, rays[i], target=g['fan_target'])
        assert (np.isnan(d) and np.isnan(d2[i])) or d == d2[i]
    # a target left of every point and one exactly at the source abscissa
    assert np.all(np.isnan(fan.dist_to([-5., -100.]).cpu().numpy()))
    d0 = fan.dist_to([0., -100.]).cpu().numpy()
    for i in range(0, fan.n, 25):
        assert d0[i] == _oracle_post(orc, oenv, rays[i], target=np.array([0., -100.]))


@pytest.mark.parametrize('name', ['c4_multifreq', 'c2_fan999', 't2_single', 't3_default10'])
def test_gain_vs_reference(rt, name):
    """populate_NEW against the reference's G (end values and sampled points)."""
    g = common.load(name)
    fan, env = _cast(rt, g)
    freqs = [float(f) for f in g['fan_freqs']]
    fan.populate(*freqs)
    ok = (g['error'] == 0) & (fan.host('n_pts') == g['steps'])
    worst = 0.0
    for k, f in enumerate(freqs):
        ge = fan.G_end[f].cpu().numpy()
        ref = g['G_end'][k]
        fin = ok & np.isfinite(ref)
        assert np.array_equal(np.isneginf(ge[ok]), np.isneginf(ref[ok]))
        e = np.abs(ge[fin] - ref[fin]) / np.maximum(1e-3, np.abs(ref[fin]))
        worst = max(worst, e.max(initial=0))
        for i in np.nonzero(fin)[0][::17]:
            gi = fan.ray_slice(fan.G[f], int(i)).cpu().numpy()[g['sample_idx'][i]]
            ri = g['sample_G'][k, i]
            m = np.isfinite(ri)
            assert np.all(np.abs(gi[m] - ri[m]) <= GAIN_RTOL * np.maximum(1e-3, np.abs(ri[m])) * 100)
    print(f'{name}: worst relative gain error at ray end {worst:.2e} over {len(freqs)} frequencies')
    assert worst < 1e-7
    # per-ray view
    ray = rt.Ray2D._from_fan(fan, 0)
    ray.populate_NEW(*freqs[:2])
    assert ray.freqs == sorted(freqs[:2]) and ray.G[freqs[0]].shape[0] == ray.steps


def test_gain_vs_oracle_same_points(rt, orc):
    """Gain kernel in isolation: the oracle integrates over the GPU's own points and rebound log."""
    g = common.load('c2_fan999')
    fan, env = _cast(rt, g, angles=g['fan_angles'][::9])
    oenv = common.oracle_env_from_product(orc, env)
    freqs = [1, 250., 1000., 4000., 10000., 12345.6, 20000., 3., 7., 11.]      # 10 -> two frequency chunks
    fan.populate(*freqs)
    rays = _oracle_rays_from_fan(orc, rt, fan)
    worst = 0.0
    for i, r in enumerate(rays):
        if r.status != 0:
            continue
        _oracle_post(orc, oenv, r)
        for f in freqs:
            G = orc.gain(oenv, r, f)
            Gd = fan.ray_slice(fan.G[f], i).cpu().numpy()
            assert np.array_equal(np.isfinite(G), np.isfinite(Gd))
            m = np.isfinite(G)
            worst = max(worst, (np.abs(G[m] - Gd[m]) / np.maximum(1e-6, np.abs(G[m]))).max(initial=0))
            assert Gd[-1] == fan.G_end[f][i].item() or (np.isneginf(Gd[-1]) and np.isneginf(fan.G_end[f][i].item()))
    print(f'gain kernel vs oracle on identical points: worst relative error {worst:.2e}')
    assert worst < 1e-11


@pytest.mark.parametrize('fast', [True, 'nohint', False])
@pytest.mark.parametrize('name', ['c1_flat250', 'c2_fan999', 'c3_bathy999', 't3_default10'])
def test_heatmap_kernel_vs_oracle_same_points(rt, orc, name, fast):
    """Binning in isolation: oracle heat map over the GPU's own points and gains -> every cell equal to 1e-12.
    fast=True: coalesced tile kernel + general kernel for the queued rays; fast=False: general kernel only."""
    g = common.load(name)
    fan, env = _cast(rt, g)
    oenv = common.oracle_env_from_product(orc, env)
    sim = rt.Simulation2D(env, g['fan_source'])
    sim.raypacks['main'].add_fan(fan)
    hint = fast is True          # 'nohint': tile kernel with its own monotone / in-map check of every ray
    fast = bool(fast)
    H, dropped = sim.heatmap_full(resolution=[50, 25], fast=fast, hint=hint)
    print(f'{name}: fast={fast} hint={hint}: {fan.last_slow_rays} of {fan.n} rays went through the general kernel')
    H = H.cpu().numpy()
    rays = [r for r in _oracle_rays_from_fan(orc, rt, fan) if r.status == 0]
    for r in rays:
        _oracle_post(orc, oenv, r)
    Href, plot_ref, drop_ref = orc.heatmap(oenv, rays, res=(50, 25), cutoff=.02)
    assert H.shape == Href.shape
    assert int(dropped.item()) == drop_ref
    assert np.array_equal(H == 0, Href == 0), f'{int(((H == 0) != (Href == 0)).sum())} cells differ in occupancy'
    rel = np.abs(H - Href) / np.maximum(1e-300, np.abs(Href))
    assert rel[Href != 0].max() < 1e-12
    for cutoff in (.02, .1, .5):
        out = sim.heatmap(resolution=[50, 25], cutoff=cutoff, reduction_power=.5, fast=fast, hint=hint)
        _, plot_ref, _ = orc.heatmap(oenv, rays, res=(50, 25), cutoff=cutoff)
        assert out.shape == plot_ref.shape
        assert np.abs(out - plot_ref).max() < 1e-12


@pytest.mark.parametrize('name', ['c1_flat250', 'c2_fan999', 'c3_bathy999', 't3_default10'])
def test_heatmap_vs_reference(rt, name):
    """Whole pipeline against the reference's heat map; bin-edge disagreements are counted and bounded."""
    g = common.load(name)
    fan, env = _cast(rt, g)
    sim = rt.Simulation2D(env, g['fan_source'])
    sim.raypacks['main'].add_fan(fan)
    pack = sim.raypacks['main']
    assert pack.n_rays == int((g['error'] == 0).sum())
    assert np.allclose(pack.range_min_plot, g['range_min_plot'], rtol=1e-9) and np.allclose(pack.range_max_plot, g['range_max_plot'], rtol=1e-9)
    H, dropped = sim.heatmap_full(resolution=[int(v) for v in g['heatmap_res']])
    H = H.cpu().numpy()
    ref = g['heatmap_full']
    assert H.shape == ref.shape and int(dropped.item()) == 0
    knife = fan.host('margin') <= KNIFE_EDGE_M
    differ = np.abs(H - ref) > HEAT_RTOL * np.maximum(1.0, np.abs(ref))
    n_bad = int(differ.sum())
    # Cells that may legitimately differ: (a) cells next to a point that sits on a cell edge to within 1e-6 m
    # (dz_max equals the cell height, so steep rays land on multiples of 25 m and the last bits decide the bin),
    # (b) any cell visited by a knife-edge ray (its later points may be shifted by one step).
    res = [int(v) for v in g['heatmap_res']]
    X, Z, valid = fan.padded_host()
    suspect = np.zeros(H.shape, dtype=bool)
    fx, fz = X / res[0], Z / res[1]
    edge = valid & ((np.abs(fx - np.round(fx)) < 1e-6 / res[0]) | (np.abs(fz - np.round(fz)) < 1e-6 / res[1]))
    edge |= valid & knife[:, None]
    ix = np.floor_divide(np.trunc(X[edge]).astype(np.int64), res[0])
    iz = np.floor_divide(np.trunc(Z[edge]).astype(np.int64), -res[1])
    for dx in (-1, 0, 1):
        for dz in (-1, 0, 1):
            a, b = ix + dx, iz + dz
            m = (a >= 0) & (a < H.shape[0]) & (b >= 0) & (b < H.shape[1])
            suspect[a[m], b[m]] = True
    print(f'{name}: heat map {H.shape}: {n_bad} of {H.size} cells differ from the reference; '
          f'{int((differ & ~suspect).sum())} of them are NOT explained by a cell-edge / knife-edge point '
          f'(edge points: {int(edge.sum())}, knife-edge rays: {int(knife.sum())})')
    assert int((differ & ~suspect).sum()) == 0
    assert n_bad <= 0.02 * H.size
    # the mass of the map moves by at most ~one deposit per edge point (a flip can merge two cells of one ray)
    assert abs(H.sum() - ref.sum()) <= 1e-9 * ref.sum() + 2.0 * int(edge.sum())
    # the presented maps at the three saturation levels of the sample images (simulation.py:133-138): the normalising
    # maximum agrees, and so does every cell outside the ones counted above
    assert abs(H.max() - ref.max()) <= 1e-9 * ref.max()
    for cutoff in (.02, .1, .5):
        out = sim.heatmap(resolution=res, cutoff=cutoff, reduction_power=.5)
        want = g[f'heatmap_plot_cut{cutoff}']
        assert out.shape == want.shape
        assert np.abs(out - want)[~differ.T].max() < 1e-9


def test_heatmap_backward_and_out_of_map_rays(rt, orc):
    """Rays that turn back (x decreases) or leave the map cannot use the column-monotone fast kernel: they are
    queued for the general kernel, and the result still equals the oracle's in every cell."""
    xs = np.linspace(0., 20000., 401)
    zs = np.where(xs < 9000., -900., -900. + (xs - 9000.) * 0.9)            # a steep wall turns rays back
    zs = np.minimum(zs, -20.)
    rmin, rmax = np.array([0., -1001.]), np.array([20000., 1.])
    env = rt.Environment2D((xs, zs), (np.array([0., 20000.]), np.array([0., 0.])), rmin, rmax)
    oenv = common.oracle_env_from_product(orc, env)
    angles = np.linspace(-0.6, 0.2, 130)
    fan = rt.RayFan(env, np.array([0., -40.]), angles, dx_max=50., dz_max=25., n_rebounds_max=12)
    X, Z, valid = fan.padded_host()
    went_back = (np.diff(np.where(valid, X, np.inf), axis=1) < 0).any(axis=1)
    assert went_back.sum() >= 5, 'fixture must contain rays that move backwards'
    sim = rt.Simulation2D(env, np.array([0., -40.]))
    sim.raypacks['main'].add_fan(fan)
    rays = [r for r in common.oracle_rays_from_fan(orc, rt, fan) if r.status == 0]
    for r in rays:
        orc.post_ray(oenv, r)
    for res in ([50, 25], [50, 7]):          # nz = 41 (one mask word) and nz = 145 (four mask words)
        Href, _, drop_ref = orc.heatmap(oenv, rays, res=tuple(res))
        for fast in (True, False):
            H, dropped = sim.heatmap_full(resolution=res, fast=fast)
            H = H.cpu().numpy()
            assert H.shape == Href.shape and int(dropped.item()) == drop_ref
            assert np.array_equal(H == 0, Href == 0)
            assert (np.abs(H - Href) / np.maximum(1e-300, np.abs(Href)))[Href != 0].max() < 1e-12
        assert fan.last_slow_rays == fan.n            # fast=False last
    H, _ = sim.heatmap_full(resolution=[50, 25], fast=True)
    assert 0 < fan.last_slow_rays < fan.n
    # a deliberately small map: points outside are dropped (the reference would raise IndexError) and counted
    Hs = np.zeros(100 * 41)
    import torch
    Hd = torch.zeros(100 * 41, dtype=torch.float64, device=fan.device)
    dr = torch.zeros(1, dtype=torch.int64, device=fan.device)
    fan.heatmap_accumulate(Hd, [50, 25], [100, 41], dropped=dr)
    L = orc.lib()
    res_c, shape_c = np.array([50, 25], dtype=np.int32), np.array([100, 41], dtype=np.int32)
    nd = 0
    for r in common.oracle_rays_from_fan(orc, rt, fan):
        orc.post_ray(oenv, r)
        G = orc.gain(oenv, r, 1)
        Xc, Zc = np.ascontiguousarray(r.XZ[:, 0]), np.ascontiguousarray(r.XZ[:, 1])
        nd += L.orc_heatmap_add(orc._dp(Xc), orc._dp(Zc), orc._dp(G), r.steps, orc._ip(res_c), orc._ip(shape_c), orc._dp(Hs))
    assert int(dr.item()) == nd and nd > 0
    assert np.abs(Hd.cpu().numpy() - Hs).max() <= 1e-12 * Hs.max()


@pytest.mark.parametrize('n_freqs', [3, 16, 4, 2, 1])
def test_gain_two_phase_equals_one_pass(rt, monkeypatch, n_freqs):
    """Small fans take the two-phase gain (per-point dG, then running sums), large ones the one-pass kernel: same
    expressions in the same order, so the arrays must be identical bit for bit.  3 frequencies: a group of four that is not
    full; 16 / 4 / 2 / 1: the instances of the one-pass kernel whose groups are all full (stores without a condition, the
    wait for the next point counted by the compiler)."""
    g = common.load('c2_fan999')
    freqs = [1., 1000., 20000.] if n_freqs == 3 else [float(f) for f in np.geomspace(1., 20000., n_freqs)]
    fan, env = _cast(rt, g)
    fan.populate(*freqs)
    monkeypatch.setenv('RT2D_GAIN_TWO_PHASE_MAX_RAYS', '0')
    fan1, _ = _cast(rt, g)
    fan1.populate(*freqs)
    # ... and the one-pass kernel with the rebound losses evaluated inside its loop (no workspace) instead of by their
    # own kernel
    from ray_tracer_amd import rays
    monkeypatch.setattr(rays, 'LOSS_WS_BYTES', 0)
    fan2, _ = _cast(rt, g)
    fan2.populate(*freqs)
    for other in (fan1, fan2):
        for f in freqs:
            assert np.array_equal(fan.G_end[f].cpu().numpy(), other.G_end[f].cpu().numpy(), equal_nan=True)
            for i in range(0, fan.n, 37):
                a, b = fan.ray_slice(fan.G[f], i).cpu().numpy(), other.ray_slice(other.G[f], i).cpu().numpy()
                assert np.array_equal(a, b, equal_nan=True)


def test_heatmaps_for_an_array_of_frequencies(rt, orc):
    """Config 4: one map per frequency from one batched gain launch; each equals the oracle's map over the GPU's own
    points at that frequency (so the frequency-dependent rebound losses are the oracle's)."""
    g = common.load('c4_multifreq')
    env = common.product_env(rt, g)
    oenv = common.oracle_env_from_product(orc, env)
    sim = rt.Simulation2D(env, g['fan_source'])
    fan = sim.cast(g['fan_angles'], **common.prop_kwargs(g))
    freqs = [float(f) for f in g['fan_freqs']]                 # all 64, binned in one pass over the trajectories
    maps = sim.heatmaps(freqs, resolution=[50, 25], cutoff=.1)
    assert set(fan.G) >= set(freqs) and len(maps) == 64
    rays = [r for r in _oracle_rays_from_fan(orc, rt, fan) if r.status == 0]
    for r in rays:
        _oracle_post(orc, oenv, r)
    one_by_one = sim.heatmaps(freqs[::9], resolution=[50, 25], cutoff=.1, batched=False)
    for f, m in one_by_one.items():
        fin = np.isfinite(m)
        assert np.array_equal(np.isfinite(maps[f]), fin) and np.abs(maps[f] - m)[fin].max(initial=0) < 1e-12
    freqs = [freqs[k] for k in (0, 5, 21, 63)]
    for f in freqs:
        _, plot_ref, _ = orc.heatmap(oenv, rays, res=(50, 25), cutoff=.1, freq=f)
        assert maps[f].shape == plot_ref.shape
        fin = np.isfinite(plot_ref)
        assert np.array_equal(np.isfinite(maps[f]), fin)
        assert np.abs(maps[f] - plot_ref)[fin].max() < 1e-11
    assert np.nanmax(np.abs(maps[freqs[0]] - maps[freqs[1]])) > 1e-3           # the maps do depend on the frequency


@pytest.mark.gpu
def test_heatmaps_with_the_gain_of_the_next_frequencies_beside_the_binning(rt, monkeypatch):
    """Many new frequencies on a big fan: the gain of the next group runs on a second stream beside the binning of this
    one (RayFan.heatmap_accumulate_multi).  Same maps as with everything on one stream, to the order of the atomics."""
    from ray_tracer_amd import rays as rays_mod
    g = common.load('c2_fan999')
    env = common.product_env(rt, g)
    n, F = 16384, 12
    angles = -(np.arange(n) + 1) * (0.999 / n)
    freqs = [float(f) for f in np.geomspace(20., 15000., F)]
    maps = {}
    for group in (4, 0):
        monkeypatch.setattr(rays_mod, 'MULTI_GROUP', group)
        monkeypatch.setattr(rays_mod, 'MULTI_PIPELINE_MIN', 1)
        sim = rt.Simulation2D(env, g['fan_source'])
        fan = sim.cast(angles, **common.prop_kwargs(g))
        maps[group] = sim.heatmaps(freqs, resolution=[50, 25], cutoff=.1)
        assert set(fan.G) >= set(freqs)
    for f in freqs:
        fin = np.isfinite(maps[0][f])
        assert np.array_equal(np.isfinite(maps[4][f]), fin)
        assert np.abs(maps[4][f] - maps[0][f])[fin].max(initial=0) < 1e-12
    assert np.nanmax(np.abs(maps[4][freqs[0]] - maps[4][freqs[-1]])) > 1e-3


@pytest.mark.gpu
def test_extreme_gains_take_the_pow_path_in_both_kernels(rt):
    """Gains beyond +-3000 dB (a rebound coefficient that underflowed to 0 gives -inf) leave the 2^x shortcut for the
    reference's two pows (ray.py:317, simulation.py:129): (10^(G/10))^-0.1 is inf / 0 there.  The tile kernel keeps
    that path out of its loop, the general kernel inline: both must agree with NumPy's own arithmetic."""
    import torch
    g = common.load('c2_fan999')
    fan, env = _cast(rt, g)
    fan.populate(1.)
    G = fan.G[1.]
    rng = np.random.default_rng(5)
    pick = torch.from_numpy(rng.random(G.numel()) < 0.02).to(G.device)
    vals = torch.from_numpy(rng.choice([-np.inf, -5000., -3000., 3000., 4000., -3200.5], G.numel())).to(G.device)
    fan.G[1.] = torch.where(pick, vals, G)
    sim = rt.Simulation2D(env, g['fan_source'])
    sim.raypacks['main'].add_fan(fan)
    Hf, _ = sim.heatmap_full(resolution=[50, 25], fast=True, hint=True)
    Hg, _ = sim.heatmap_full(resolution=[50, 25], fast=False)
    Hf, Hg = Hf.cpu().numpy(), Hg.cpu().numpy()
    assert np.isinf(Hg).any() and (Hg[np.isfinite(Hg)] > 0).any()
    assert np.array_equal(np.isinf(Hf), np.isinf(Hg))
    fin = np.isfinite(Hg)
    assert np.abs(Hf[fin] - Hg[fin]).max() <= 1e-12 * np.abs(Hg[fin]).max()
    # one ray against NumPy: last write wins per cell, value by the two pows
    i = int(np.nonzero(fan.host('status') == 0)[0][3])
    ray = rt.Ray2D._from_fan(fan, i)
    Gi = fan.G[1.][fan.tile_off[i // 64] + i % 64::64][:ray.steps].cpu().numpy()
    mask = torch.zeros(fan.n, dtype=torch.uint8, device=G.device); mask[i] = 1
    shape = Hg.shape
    H1 = torch.zeros(shape[0] * shape[1], dtype=torch.float64, device=G.device)
    fan.heatmap_accumulate(H1, [50, 25], shape, freq=1., ray_mask=mask, fast=True, hint=True)
    want = np.zeros(shape)
    with np.errstate(all='ignore'):
        v = np.power(np.power(10., Gi / 10), -0.1)
    ix, iz = ray.XZ[:, 0].astype(int) // 50, ray.XZ[:, 1].astype(int) // -25
    want[ix, iz] = v
    got = H1.cpu().numpy().reshape(shape)
    assert np.array_equal(np.isinf(got), np.isinf(want))
    f = np.isfinite(want)
    assert np.abs(got[f] - want[f]).max() <= 1e-12 * max(1.0, np.abs(want[f]).max())


def test_finalize_of_a_negative_map(rt):
    """rt2d_heatmap_finalize normalises by np.max of the map whatever its sign (rt2d_cast_heatmap can accumulate with
    sign = -1): a map of negative cells against the NumPy expression of simulation.py:133-136, and a NaN cell -> NaN."""
    import torch
    rng = np.random.default_rng(3)
    nx, nz = 37, 11
    for H in (-(rng.random((nx, nz)) * 5 + 1), rng.random((nx, nz)) * 3, rng.normal(size=(nx, nz)) + 0.1):
        Hd = torch.from_numpy(np.ascontiguousarray(H)).cuda()
        out = torch.empty((nz, nx), dtype=torch.float64, device='cuda')
        scratch = torch.empty(2, dtype=torch.float64, device='cuda')
        rt._lib.check(rt._lib.lib().rt2d_heatmap_finalize(Hd.data_ptr(), (C.c_int32 * 2)(nx, nz), .5, 1.0, out.data_ptr(),
                                                          scratch.data_ptr(), torch.cuda.current_stream().cuda_stream))
        norm = H / np.max(H)
        norm[norm > .5] = .5
        assert np.abs(out.cpu().numpy() - norm.T).max() < 1e-15
    Hn = torch.ones((nx, nz), dtype=torch.float64, device='cuda')
    Hn[3, 4] = float('nan')
    out = torch.empty((nz, nx), dtype=torch.float64, device='cuda')
    scratch = torch.empty(2, dtype=torch.float64, device='cuda')
    rt._lib.check(rt._lib.lib().rt2d_heatmap_finalize(Hn.data_ptr(), (C.c_int32 * 2)(nx, nz), .5, .5, out.data_ptr(),
                                                      scratch.data_ptr(), torch.cuda.current_stream().cuda_stream))
    assert bool(torch.isnan(out).all().item())            # np.max of a map with a NaN is NaN, and so is every quotient


def test_populate_after_keep_arrays_false(rt):
    """A frequency populated with keep_arrays=False (end gains only) gets its array when it is asked for later."""
    g = common.load('c1_flat250')
    fan, env = _cast(rt, g, angles=g['fan_angles'][:70])
    fan.populate(1000., keep_arrays=False)
    assert 1000. in fan.G_end and 1000. not in fan.G
    import torch
    H = torch.zeros(1100 * 45, dtype=torch.float64, device='cuda')
    fan.heatmap_accumulate(H, (50, 25), (1100, 45), freq=1000.)
    assert 1000. in fan.G and float(H.sum().item()) > 0


def test_multi_frequency_binning_on_a_large_fan(rt):
    """20 000 rays (tile kernel for the forward rays, general kernel for the turned-back ones), 6 frequencies -- not a
    multiple of the 4 maps a pass bins -- in one pass against one heat map per frequency: every cell to 1e-12."""
    g = common.load('c2_fan999')
    env = common.product_env(rt, g)
    sim = rt.Simulation2D(env, g['fan_source'])
    fan = sim.cast(-(np.arange(20000) + 1) * (0.999 / 20000), **common.prop_kwargs(g))
    freqs = [1., 40., 300., 1000., 5000., 20000.]
    both = sim.heatmaps(freqs, resolution=[50, 25], cutoff=.5)
    assert fan.last_slow_rays > 0
    for f in freqs:
        one = sim.heatmap(freq=f, resolution=[50, 25], cutoff=.5)
        assert one.shape == both[f].shape
        fin = np.isfinite(one)
        assert np.array_equal(np.isfinite(both[f]), fin) and not np.isnan(both[f][fin]).any()
        assert np.abs(both[f] - one)[fin].max() < 1e-12


@pytest.mark.parametrize('name,freq', [('c1_flat250', 1), ('c2_fan999', 1), ('c3_bathy999', 1), ('c2_fan999', 4000.), ('t2_single', 1000.)])
def test_small_fan_heatmap_in_one_launch(rt, name, freq, monkeypatch):
    """rt2d_small_fan_heatmap (gain + binning of a ray by one workgroup, then max + normalisation: what Simulation2D.heatmap
    runs for a small fan) against rt2d_gain + rt2d_heatmap_accumulate + rt2d_heatmap_finalize of the same stored fan, which
    the tests above pin to the oracle and the reference: every cell to 1e-12, same dropped-point count, and no gain
    array is made.  Through the C ABI directly and through the product API.  t2_single: a ray of 100 001 points does not
    fit a workgroup -- the kernel says so in the word behind the fan's summary and the product bins the fan the other way."""
    import torch
    from ray_tracer_amd import rays
    g = common.load(name)
    fan, env = _cast(rt, g)                        # (fresh single-pass cast: its summary has not been read yet)
    shape = (1610, 210) if name == 't2_single' else (1600, 80)
    cells = shape[0] * shape[1]
    Ha = torch.zeros(cells, dtype=torch.float64, device='cuda'); Hb = torch.zeros_like(Ha)
    da = torch.zeros(1, dtype=torch.int64, device='cuda'); db = torch.zeros_like(da)
    assert fan.small_tail_room(freq, policy=False) > 0
    out = torch.empty((shape[1], shape[0]), dtype=torch.float64, device='cuda')
    scratch = torch.empty(2, dtype=torch.float64, device='cuda')
    fan.heatmap_small(Ha, (50, 25), shape, freq=freq, dropped=da, out=out, scratch=scratch, cutoff=.1, reduction_power=.5)
    assert freq not in fan.G
    assert fan.validate() is False
    monkeypatch.setattr(rays, 'HEATMAP_FAST_MIN_RAYS', 6000)      # (the test session sets the tile kernel's threshold to 0)
    if name == 't2_single':
        assert fan._onepass_overflow and float(Ha.sum()) == 0.0
        assert fan.small_tail_room(freq, policy=False) == 0        # ... and once the summary is known the fan no longer qualifies
    else:
        assert not fan._onepass_overflow
        mask = (fan.status == 0).to(torch.uint8)
        fan.heatmap_accumulate(Hb, (50, 25), shape, freq=freq, ray_mask=mask, dropped=db, stream=False)
        Ha_, Hb_ = Ha.cpu().numpy(), Hb.cpu().numpy()
        assert int(da.item()) == int(db.item())
        assert np.array_equal(Ha_ == 0, Hb_ == 0) and np.array_equal(np.isinf(Ha_), np.isinf(Hb_)) and not np.isnan(Ha_).any()
        both = np.isfinite(Hb_) & (Hb_ != 0)
        assert (np.abs(Ha_ - Hb_)[both] / np.abs(Hb_[both])).max() < 1e-12
        ref = torch.empty_like(out)
        rt._lib.check(rt._lib.lib().rt2d_heatmap_finalize(Hb.data_ptr(), (C.c_int32 * 2)(*shape), .1, .5, ref.data_ptr(), scratch.data_ptr(),
                                                          torch.cuda.current_stream().cuda_stream))
        o, q = out.cpu().numpy(), ref.cpu().numpy()
        fin = np.isfinite(q)
        assert np.array_equal(np.isfinite(o), fin) and np.abs(o[fin] - q[fin]).max() < 1e-12
    # the product API takes this path for a freshly cast small fan (below the tile kernel's threshold), and the general
    # (three-launch) path on request -- or by itself when a ray turns out too long
    sim = rt.Simulation2D(env, g['fan_source'])
    sim.cast(g['fan_angles'], **common.prop_kwargs(g))
    a = sim.heatmap(resolution=[50, 25], cutoff=.1, freq=freq)
    sim2 = rt.Simulation2D(env, g['fan_source'])
    sim2.cast(g['fan_angles'], **common.prop_kwargs(g))
    b = sim2.heatmap(resolution=[50, 25], cutoff=.1, freq=freq, small=False)
    fin = np.isfinite(b)
    assert a.shape == b.shape and np.array_equal(np.isfinite(a), fin) and np.abs(a[fin] - b[fin]).max() < 1e-12
    # argument checks of the entry point: a room beyond 2048 points needs the overflow word
    with pytest.raises(rt._lib.Rt2dError, match='2048'):
        rt._lib.check(rt._lib.lib().rt2d_small_fan_heatmap(
            env.device_handle(0), fan.X.data_ptr(), fan.Z.data_ptr(), fan.tile_off.data_ptr(), fan.n_pts.data_ptr(), C.byref(fan._out),
            fan.rebound_cap, fan.n, 5000, 1.0, None, (C.c_int32 * 2)(50, 25), (C.c_int32 * 2)(*shape), Ha.data_ptr(), None, .1, .5, None, None,
            None, None))


def test_gain_of_rays_with_thousands_of_rebounds(rt, orc):
    """No rebound limit in 10 m of water: steep rays rebound every few steps and log over a thousand rebounds, more than the
    LDS of the scan kernels holds.  The small-fan gain then adds the losses into the per-point dG (rt2d_reb_scatter_kernel)
    and scans without a log; against the oracle on the GPU's own points and log, and against the ray-sequential kernel."""
    xs = np.linspace(0., 6000., 61)
    env = rt.Environment2D((xs, np.full_like(xs, -10.)), (np.array([0., 6000.]), np.array([0., 0.])),
                           np.array([0., -31.]), np.array([6000., 1.]))
    oenv = common.oracle_env_from_product(orc, env)
    angles = np.linspace(-1.3, -0.9, 40)
    kw = dict(dx_max=10., dz_max=20., n_rebounds_max=-1, n_steps_max=8000)
    fan = rt.RayFan(env, np.array([0., -5.]), angles, **kw)
    fan.validate()
    assert int(fan.n_rebounds.max().item()) > 1400 and fan.rebound_cap > 1400
    freqs = [1., 2500.]
    fan.populate(*freqs)
    rays = common.oracle_rays_from_fan(orc, rt, fan)
    worst = 0.0
    for i in range(0, fan.n, 3):
        r = rays[i]
        if r.status != 0:
            continue
        orc.post_ray(oenv, r)
        for f in freqs:
            G = orc.gain(oenv, r, f)
            Gd = fan.ray_slice(fan.G[f], i).cpu().numpy()
            assert np.array_equal(np.isfinite(G), np.isfinite(Gd))
            m = np.isfinite(G)
            worst = max(worst, (np.abs(G[m] - Gd[m]) / np.maximum(1e-6, np.abs(G[m]))).max(initial=0))
            ge = fan.G_end[f][i].item()
            assert Gd[-1] == ge or (np.isneginf(Gd[-1]) and np.isneginf(ge))
    print(f'gain with the losses scattered into dG vs oracle: worst relative error {worst:.2e}')
    assert worst < 1e-10


def test_gains_queued_behind_a_cast_that_is_done_again(rt):
    """populate() may be queued behind a single-pass cast whose summary has not been read.  When that cast turns out to need
    doing again (here: rebound logs beyond the provisional 64 entries) the gains are computed again on the final cast:
    fan.G[f] is still there after validate() and equals the gains of a fan that was validated first."""
    xs = np.linspace(0., 6000., 61)
    env = rt.Environment2D((xs, np.full_like(xs, -10.)), (np.array([0., 6000.]), np.array([0., 0.])),
                           np.array([0., -31.]), np.array([6000., 1.]))
    angles = np.linspace(-1.3, -0.9, 40)
    kw = dict(dx_max=10., dz_max=20., n_rebounds_max=-1, n_steps_max=2000)
    first = rt.RayFan(env, np.array([0., -5.]), angles, **kw)
    first.validate()
    first.populate(1., 2500.)
    late = rt.RayFan(env, np.array([0., -5.]), angles, **kw)
    assert late._pending
    late.populate(1., 2500.)
    assert late.validate()                                   # cast again: the log overflowed
    assert late.rebound_cap == first.rebound_cap > 64
    for f in (1., 2500.):
        assert np.array_equal(late.G_end[f].cpu().numpy(), first.G_end[f].cpu().numpy(), equal_nan=True)
        for i in range(0, late.n, 7):
            assert np.array_equal(late.ray_slice(late.G[f], i).cpu().numpy(), first.ray_slice(first.G[f], i).cpu().numpy(), equal_nan=True)


def test_hint_bits_and_the_map_minor_scratch(rt):
    """Two helpers of the binning entry points through the C ABI.  rt2d_heatmap_hints (the hint bits of the ray mask in one
    launch) against the same conditions written with tensor operations; rt2d_heatmap_accumulate_multi with and without its
    map-minor scratch H_minor (one atomic request per run for four maps, added into the maps at the end): same maps to
    1e-12, for a number of frequencies that is not a multiple of four, on a fan that uses both binning kernels."""
    import ctypes as C
    import torch
    from ray_tracer_amd import rays
    g = common.load('c2_fan999')
    env = common.product_env(rt, g)
    fan = rt.RayFan(env, g['fan_source'], -(np.arange(9000) + 1) * (0.999 / 9000), **common.prop_kwargs(g))
    fan.validate()
    L = rt._lib.lib()
    res, shape = [50, 25], [1500, 61]
    res_c, shape_c = (C.c_int32 * 2)(*res), (C.c_int32 * 2)(*shape)
    stream = torch.cuda.current_stream().cuda_stream
    keep = (torch.arange(fan.n, device='cuda') % 7 != 3).to(torch.uint8)
    for mask in (None, keep):
        out = torch.empty(fan.n, dtype=torch.uint8, device='cuda')
        rt._lib.check(L.rt2d_heatmap_hints(fan.monotone.data_ptr(), fan.range.data_ptr(), rays._ptr(mask), fan.n, res_c, shape_c,
                                           out.data_ptr(), stream))
        rng = fan.range
        inside = ((rng[0] > -1) & (rng[3] < 1)
                  & (torch.div(torch.trunc(rng[2]), res[0], rounding_mode='floor') < shape[0])
                  & (torch.div(-torch.trunc(rng[1]), res[1], rounding_mode='floor') < shape[1]))
        base = torch.ones(fan.n, dtype=torch.uint8, device='cuda') if mask is None else (mask != 0).to(torch.uint8)
        good = base * (inside & (fan.monotone != 0)).to(torch.uint8)
        want = base + 2 * good + 4 * (base - good)
        assert torch.equal(out, want)
        assert int((out == 3).sum().item()) > 0 and int((out == 5).sum().item()) > 0
    with pytest.raises(rt._lib.Rt2dError):
        rt._lib.check(L.rt2d_heatmap_hints(None, fan.range.data_ptr(), None, fan.n, res_c, shape_c, out.data_ptr(), stream))

    freqs = [1., 300., 1000., 5000., 20000.]
    fan.populate(*freqs)
    cells = shape[0] * shape[1]
    ok = (fan.status == 0).to(torch.uint8)
    hints = fan.heatmap_mask(res, shape, ok)
    for nf in (5, 4):                                        # (groups of eight maps in the scratch; groups of four)
        ptrs = torch.tensor([fan.G[f].data_ptr() for f in freqs[:nf]], dtype=torch.int64, device='cuda')
        maps = []
        for use_minor in (False, True):
            H = torch.zeros((nf, cells), dtype=torch.float64, device='cuda')
            work = torch.empty(fan.n + 2, dtype=torch.int32, device='cuda')
            minor = torch.full(((nf + 3) // 4 * 4 * cells,), float('nan'), dtype=torch.float64, device='cuda') if use_minor else None   # (contents irrelevant on entry)
            rt._lib.check(L.rt2d_heatmap_accumulate_multi(fan.X.data_ptr(), fan.Z.data_ptr(), ptrs.data_ptr(), nf,
                                                          fan.tile_off.data_ptr(), fan.n_pts.data_ptr(), fan.n, hints.data_ptr(),
                                                          res_c, shape_c, H.data_ptr(), None, work.data_ptr(), rays._ptr(minor), stream))
            assert int(work[0].item()) > 0                   # (rays went to the general kernel too)
            maps.append(H.cpu().numpy())
        a, b = maps
        fin = np.isfinite(a)
        assert np.array_equal(np.isfinite(b), fin) and np.array_equal(a == 0, b == 0)
        m = fin & (a != 0)
        assert (np.abs(a[m] - b[m]) / np.abs(a[m])).max() < 1e-12
