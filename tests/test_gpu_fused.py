"""GPU: the fused cast -> gain -> heat-map kernel (rt2d_cast_heatmap) against the stored pipeline
(rt2d_cast_write -> rt2d_gain -> rt2d_heatmap_accumulate), which the other GPU tests pin to the oracle and to the
reference.  Same rays, same map: every cell to 1e-12 relative (the f64 atomics arrive in a different order)."""

import numpy as np
import pytest

import common

pytestmark = pytest.mark.gpu


def _maps(rt, g, angles=None, freq=1, res=(50, 25), env=None, **over):
    env = env or common.product_env(rt, g)
    kw = common.prop_kwargs(g)
    kw.update(over)
    angles = g['fan_angles'] if angles is None else angles
    stored = rt.Simulation2D(env, g['fan_source'])
    stored.cast(angles, **kw)
    fused = rt.Simulation2D(env, g['fan_source'])
    fan = fused.cast(angles, store=False, **kw)
    Hs, _ = stored.heatmap_full(resolution=list(res), freq=freq)
    Hf, _ = fused.heatmap_full(resolution=list(res), freq=freq)
    return Hs.cpu().numpy(), Hf.cpu().numpy(), fan, stored, fused


@pytest.mark.parametrize('name', ['c1_flat250', 'c2_fan999', 'c3_bathy999'])
def test_fused_map_equals_stored_map(rt, name):
    g = common.load(name)
    Hs, Hf, fan, stored, fused = _maps(rt, g)
    assert Hs.shape == Hf.shape
    assert int(fan.last_fused_violations.item()) == 0
    print(f'{name}: {fan.n - fan.last_slow_rays} of {fan.n} rays binned by the fused kernel, {fan.last_slow_rays} by the general path')
    assert fan.last_slow_rays < fan.n // 2
    rel = np.abs(Hf - Hs) / np.maximum(1e-300, np.abs(Hs))
    assert np.array_equal(Hf == 0, Hs == 0)
    assert rel[Hs != 0].max() < 1e-12
    for cutoff in (.02, .5):
        a = stored.heatmap(resolution=[50, 25], cutoff=cutoff)
        b = fused.heatmap(resolution=[50, 25], cutoff=cutoff)
        assert np.abs(a - b).max() < 1e-12


@pytest.mark.parametrize('freq,res', [(1000., (50, 25)), (1, (20, 5)), (5000., (100, 10)), (1, (7, 3))])
def test_fused_other_frequencies_and_cells(rt, freq, res):
    """Finer cells put several rows and several points into one column (the slot path); a high frequency makes the
    rebound losses matter."""
    g = common.load('c2_fan999')
    Hs, Hf, fan, _, _ = _maps(rt, g, angles=g['fan_angles'][::3], freq=freq, res=res)
    assert int(fan.last_fused_violations.item()) == 0
    print(f'freq {freq} res {res}: {fan.last_slow_rays} of {fan.n} rays through the general path')
    assert np.array_equal(Hf == 0, Hs == 0)
    both = np.isfinite(Hs) & (Hs != 0)
    # a surface rebound at high frequency gives G = -inf, i.e. a deposited value of +inf: the same cells must hold it in
    # both maps, and no cell may turn into NaN (inf - inf when a ray comes back to a row it left inside one column)
    assert np.array_equal(np.isinf(Hf), np.isinf(Hs))
    assert not np.isnan(Hf).any() and not np.isnan(Hs).any()
    assert (np.abs(Hf - Hs)[both] / np.abs(Hs[both])).max() < 1e-12


def test_fused_large_fan(rt):
    """65 536 rays: the launch shape of real use (full workgroups, many waves per CU)."""
    g = common.load('c2_fan999')
    angles = -(np.arange(1 << 16) + 1) * (0.999 / (1 << 16))
    Hs, Hf, fan, _, _ = _maps(rt, g, angles=angles)
    assert int(fan.last_fused_violations.item()) == 0
    assert np.array_equal(Hf == 0, Hs == 0)
    assert (np.abs(Hf - Hs)[Hs != 0] / Hs[Hs != 0]).max() < 1e-12


def test_fused_kernel_flags_a_ray_it_cannot_bin(rt):
    """Called without the mask, the kernel must flag (not silently mis-bin) rays that move backwards."""
    import ctypes as C
    import torch
    from ray_tracer_amd import _lib
    g = common.load('c2_fan999')
    env = common.product_env(rt, g)
    fan = rt.RayFan(env, g['fan_source'], g['fan_angles'], store=False, **common.prop_kwargs(g))
    n_back = int(((fan.monotone == 0) & (fan.status == 0)).sum().item())
    assert n_back > 0
    shape = (1600, 70)
    H = torch.zeros(shape[0] * shape[1], dtype=torch.float64, device='cuda')
    irr = torch.zeros(fan.n, dtype=torch.uint8, device='cuda')
    viol = torch.zeros(1, dtype=torch.int64, device='cuda')
    src = (C.c_double * 2)(*[float(v) for v in g['fan_source']])
    p = fan._params(0)
    _lib.check(_lib.lib().rt2d_cast_heatmap(env.device_handle(0), fan.angles_dev.data_ptr(), fan.n, src, C.byref(p), None, None,
                                           None, 1.0, (C.c_int32 * 2)(50, 25), (C.c_int32 * 2)(*shape), 1.0, H.data_ptr(), None,
                                           irr.data_ptr(), viol.data_ptr(), None, 0, torch.cuda.current_stream().cuda_stream))
    assert int(viol.item()) == int(irr.sum().item()) >= n_back
    assert bool((irr[(fan.monotone == 0) & (fan.status == 0)] == 1).all().item())


def test_a_ray_that_slips_through_the_mask_makes_the_map_start_over(rt, monkeypatch):
    """The masks are meant to keep every ray the one-pass binner cannot take out of it.  Should one slip through -- here the
    mask is made to vouch for rays that turn back -- the kernels count it (last_fused_violations) and the map would be
    incomplete: heatmap() must notice, mark the fan and bin it again the exact way.  Both forms: the fused kernel of a
    store=False fan and the one pass over stored trajectories."""
    import torch
    from ray_tracer_amd import rays as rays_mod
    g = common.load('c2_fan999')
    env = common.product_env(rt, g)
    kw = common.prop_kwargs(g)
    ref_sim = rt.Simulation2D(env, g['fan_source'])
    ref_sim.cast(g['fan_angles'], **kw)
    Hs = ref_sim.heatmap_full(resolution=[50, 25], stream=False)[0].cpu().numpy()

    def vouch_for_all(self, res, shape, ray_mask=None):
        ok = self.status == 0
        if ray_mask is not None:
            ok = ok & (ray_mask != 0)
        return ok.to(torch.uint8)
    monkeypatch.setattr(rays_mod.RayFan, 'fused_mask', vouch_for_all)
    for store, extra in ((False, {}), (True, {'stream': True})):
        sim = rt.Simulation2D(env, g['fan_source'])
        fan = sim.cast(g['fan_angles'], store=store, **kw)
        assert int(((fan.monotone == 0) & (fan.status == 0)).sum().item()) > 0
        H = sim.heatmap_full(resolution=[50, 25], **extra)[0].cpu().numpy()
        assert fan.__dict__.get('_exact_binning') is True
        if not store:                                     # (the second pass sends only the flagged rays the stored way)
            n_flagged = int(fan._irregular.sum().item())
            assert 0 < n_flagged == fan.last_slow_rays < fan.n // 2
        assert np.array_equal(H == 0, Hs == 0)
        assert (np.abs(H - Hs)[Hs != 0] / Hs[Hs != 0]).max() < 1e-12


def test_fused_at_default_resolution(rt):
    """dx_max = 10, dz_max = 0.1 (preferences.py:6-7): up to 100 001 points per ray, the case the fused kernel exists for
    (a stored 1 M-ray fan would need 1.6 TB).  64 rays on the c2 environment, fused against stored."""
    g = common.load('c2_fan999')
    angles = np.linspace(-1.2, -0.05, 64)
    Hs, Hf, fan, _, _ = _maps(rt, g, angles=angles, res=(50, 25), dx_max=10., dz_max=.1, n_rebounds_max=-1)
    assert int(fan.n_pts.max().item()) > 50000
    assert int(fan.last_fused_violations.item()) == 0
    print(f'default resolution: {fan.ray_steps()} ray-steps, {fan.last_slow_rays} of {fan.n} rays through the stored path')
    assert np.array_equal(Hf == 0, Hs == 0)
    assert (np.abs(Hf - Hs)[Hs != 0] / Hs[Hs != 0]).max() < 1e-12


def test_fused_falls_back_when_the_tables_do_not_fit(rt):
    """The tests_2.ipynb environment (5 km deep, 4001 floor samples) has 200 KB of tables: the fused kernel declines and
    heatmap() on a store=False fan bins every ray through the stored path instead -- same map."""
    g = common.load('t2_single')
    angles = np.linspace(-1.0, -0.1, 50)
    Hs, Hf, fan, _, _ = _maps(rt, g, angles=angles, dx_max=50., dz_max=25., n_rebounds_max=6)
    assert 'LDS' in fan.fused_unavailable and fan.last_slow_rays == int((fan.status == 0).sum().item())
    assert np.array_equal(Hf == 0, Hs == 0)
    assert (np.abs(Hf - Hs)[Hs != 0] / Hs[Hs != 0]).max() < 1e-12
