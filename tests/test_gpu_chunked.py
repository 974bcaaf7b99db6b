"""GPU: heat maps of fans that keep no trajectories, chunk by chunk (RayFan.heatmap_accumulate_chunked: cast a chunk with
trajectories -> bin it through the stored pipeline -> drop it, chunks alternating between two streams), and the lazy
store=False fan that gets its per-ray results from those casts.  Against the stored pipeline of the same rays, which
the other GPU tests pin to the oracle and to the reference: every cell to 1e-12 relative (the f64 atomics arrive in a
different order), per-ray results bit for bit."""

import numpy as np
import pytest

import common

pytestmark = pytest.mark.gpu

SCALARS = ('n_pts', 'stop_code', 'n_rebounds', 'status', 'monotone', 'slope_max', 'range')


@pytest.fixture
def small_chunks(monkeypatch):
    from ray_tracer_amd import rays
    monkeypatch.setattr(rays, 'CHUNK_RAYS', 300)
    monkeypatch.setattr(rays, 'CHUNKED_MIN_RAYS', 500)
    return rays


def _stored(rt, g, env, freq=1, **kw):
    sim = rt.Simulation2D(env, g['fan_source'])
    fan = sim.cast(g['fan_angles'], **kw)
    H, _ = sim.heatmap_full(resolution=[50, 25], freq=freq)
    return sim, fan, H.cpu().numpy()


def _same_map(Hc, Hs):
    assert Hc.shape == Hs.shape
    assert np.array_equal(Hc == 0, Hs == 0)
    assert np.array_equal(np.isinf(Hc), np.isinf(Hs)) and not np.isnan(Hc).any()
    both = np.isfinite(Hs) & (Hs != 0)
    assert (np.abs(Hc - Hs)[both] / np.abs(Hs[both])).max() < 1e-12


@pytest.mark.parametrize('name,freq', [('c2_fan999', 1), ('c3_bathy999', 1), ('c2_fan999', 1000.)])
def test_lazy_fan_is_cast_by_its_heat_map(rt, small_chunks, name, freq):
    """999 rays in four chunks on two streams: same map as the stored fan, the same per-ray results, and no counting
    pass (the fan says how it was cast)."""
    g = common.load(name)
    env = common.product_env(rt, g)
    kw = common.prop_kwargs(g)
    sim_s, fan_s, Hs = _stored(rt, g, env, freq=freq, **kw)
    sim = rt.Simulation2D(env, g['fan_source'])
    fan = sim.cast(g['fan_angles'], store=False, **kw)
    assert fan.__dict__['_uncast']                      # nothing has been cast yet
    Hc, _ = sim.heatmap_full(resolution=[50, 25], freq=freq)
    assert fan.passes == 'chunked' and not fan.__dict__['_uncast']
    _same_map(Hc.cpu().numpy(), Hs)
    for f in SCALARS:
        assert np.array_equal(getattr(fan, f).cpu().numpy(), getattr(fan_s, f).cpu().numpy()), f
    assert fan.last_slow_rays == fan_s.last_slow_rays
    # the normalised map through the one-fan path, and the pack behind it
    a = sim_s.heatmap(resolution=[50, 25], cutoff=.5, freq=freq)
    b = sim.heatmap(resolution=[50, 25], cutoff=.5, freq=freq)     # a second map of the same fan: chunked again, counted sizes
    assert np.nanmax(np.abs(a - b)) < 1e-12
    assert sim.raypacks[sim.pack_default].n_rays == sim_s.raypacks[sim_s.pack_default].n_rays


def test_lazy_fan_is_cast_by_the_first_result_asked_for(rt, small_chunks):
    g = common.load('c2_fan999')
    env = common.product_env(rt, g)
    kw = common.prop_kwargs(g)
    eager = rt.RayFan(env, g['fan_source'], g['fan_angles'], store=False, **kw)
    lazy = rt.RayFan(env, g['fan_source'], g['fan_angles'], store=False, lazy=True, **kw)
    assert lazy.__dict__['_uncast'] and 'n_pts' not in lazy.__dict__
    assert np.array_equal(lazy.n_pts.cpu().numpy(), eager.n_pts.cpu().numpy())          # this casts it (counting pass)
    assert lazy.passes == 'count' and not lazy.__dict__['_uncast']
    for f in SCALARS:
        assert np.array_equal(getattr(lazy, f).cpu().numpy(), getattr(eager, f).cpu().numpy()), f
    # ... and its heat map then runs chunk by chunk with the known point counts
    import torch
    shape = (1600, 70)
    Hc = torch.zeros(shape[0] * shape[1], dtype=torch.float64, device='cuda')
    lazy.heatmap_accumulate(Hc, (50, 25), shape)
    stored = rt.RayFan(env, g['fan_source'], g['fan_angles'], **kw)
    Hs = torch.zeros_like(Hc)
    stored.heatmap_accumulate(Hs, (50, 25), shape, ray_mask=(stored.status == 0).to(torch.uint8))
    _same_map(Hc.cpu().numpy(), Hs.cpu().numpy())


def test_chunk_that_outgrows_its_provisional_layout(rt, small_chunks, monkeypatch):
    """Room for 100 points per ray where rays have over a thousand: the chunked map notices in its one read, the fan is
    cast the ordinary way and the map is built again from counted sizes."""
    monkeypatch.setattr(small_chunks, 'SPECULATIVE_BYTES', 0)
    monkeypatch.setattr(small_chunks, 'SPECULATIVE_ROOM', 100)
    g = common.load('c2_fan999')
    env = common.product_env(rt, g)
    kw = common.prop_kwargs(g)
    sim = rt.Simulation2D(env, g['fan_source'])
    fan = sim.cast(g['fan_angles'], store=False, **kw)
    Hc, _ = sim.heatmap_full(resolution=[50, 25])
    assert fan.passes == 'count' and int(fan.n_pts.max().item()) > 100
    monkeypatch.undo()
    _, _, Hs = _stored(rt, g, env, **kw)
    _same_map(Hc.cpu().numpy(), Hs)


def test_counted_chunks_size_the_rebound_log_by_the_known_counts(rt, small_chunks):
    """a store=False fan that was counted before casts each chunk straight into its exact layout (subset() ->
    _need_store()).  With n_rebounds_max = -1 the rebound log of that pass used to be the default 64 entries; rays with
    more rebounds then carried the log-overflow status and silently fell out of the map.  Shallow water (100 m), steep
    rays, no rebound limit: well over 64 rebounds per ray; same map as the stored fan, no ray lost."""
    import torch
    xs = np.linspace(0., 20000., 201)
    env = rt.Environment2D((xs, np.full_like(xs, -100.)), (np.array([0., 20000.]), np.array([0., 0.])),
                           np.array([0., -101.]), np.array([20000., 1.]))
    source = np.array([0., -50.])
    angles = np.linspace(-1.2, -0.4, 600)
    kw = dict(dx_max=50., dz_max=25., n_rebounds_max=-1, n_steps_max=4000)
    counted = rt.RayFan(env, source, angles, store=False, **kw)
    assert int(counted.n_rebounds.max().item()) > 64
    shape = (402, 6)
    Hc = torch.zeros(shape[0] * shape[1], dtype=torch.float64, device='cuda')
    dropped_c = torch.zeros(1, dtype=torch.int64, device='cuda')
    assert counted.heatmap_accumulate_chunked(Hc, (50, 25), shape, dropped=dropped_c)
    stored = rt.RayFan(env, source, angles, **kw)
    assert np.array_equal(stored.host('n_rebounds'), counted.host('n_rebounds'))
    assert int((stored.host('status') != 0).sum()) == 0
    Hs = torch.zeros_like(Hc)
    stored.heatmap_accumulate(Hs, (50, 25), shape, ray_mask=(stored.status == 0).to(torch.uint8))
    assert float(Hs.sum()) > 0
    _same_map(Hc.cpu().numpy(), Hs.cpu().numpy())
