"""GPU: gain + binning in one pass over stored trajectories (rt2d_gain_heatmap) against rt2d_gain followed by
rt2d_heatmap_accumulate, which the other GPU tests pin to the oracle and to the reference."""

import numpy as np
import pytest

import common

pytestmark = pytest.mark.gpu


def _both(rt, g, angles=None, freq=1, res=(50, 25)):
    env = common.product_env(rt, g)
    sim = rt.Simulation2D(env, g['fan_source'])
    fan = sim.cast(g['fan_angles'] if angles is None else angles, **common.prop_kwargs(g))
    Hs, _ = sim.heatmap_full(resolution=list(res), freq=freq, stream=True)
    Hs = Hs.cpu().numpy()
    slow, viol = fan.last_slow_rays, int(fan.last_fused_violations.item())
    assert freq not in fan.G                       # the one-pass form never materialises the gain array
    Hr, _ = sim.heatmap_full(resolution=list(res), freq=freq, stream=False)
    return Hs, Hr.cpu().numpy(), fan, slow, viol


@pytest.mark.parametrize('name', ['c1_flat250', 'c2_fan999', 'c3_bathy999'])
def test_stream_map_equals_two_pass_map(rt, name):
    Hs, Hr, fan, slow, viol = _both(rt, common.load(name))
    assert viol == 0 and slow < fan.n // 2
    assert np.array_equal(Hs == 0, Hr == 0)
    assert (np.abs(Hs - Hr)[Hr != 0] / Hr[Hr != 0]).max() < 1e-12


@pytest.mark.parametrize('freq,res', [(1000., (50, 25)), (1, (20, 5)), (5000., (100, 10)), (1, (7, 3))])
def test_stream_other_frequencies_and_cells(rt, freq, res):
    g = common.load('c2_fan999')
    Hs, Hr, fan, slow, viol = _both(rt, g, angles=g['fan_angles'][::3], freq=freq, res=res)
    assert viol == 0
    assert np.array_equal(Hs == 0, Hr == 0) and np.array_equal(np.isinf(Hs), np.isinf(Hr))
    assert not np.isnan(Hs).any() and not np.isnan(Hr).any()
    both = np.isfinite(Hr) & (Hr != 0)
    assert (np.abs(Hs - Hr)[both] / np.abs(Hr[both])).max() < 1e-12


def test_stream_large_fan_padded_layout(rt):
    g = common.load('c2_fan999')
    angles = -(np.arange(1 << 16) + 1) * (0.999 / (1 << 16))
    Hs, Hr, fan, slow, viol = _both(rt, g, angles=angles)
    assert fan.passes == 'padded' and viol == 0
    assert np.array_equal(Hs == 0, Hr == 0)
    assert (np.abs(Hs - Hr)[Hr != 0] / Hr[Hr != 0]).max() < 1e-12


def test_one_pass_map_queued_without_waiting_for_the_cast(rt, monkeypatch):
    """Simulation2D.heatmap on a big stored fan queues the one-pass map behind the cast without reading anything back: the
    rays are sorted out on the device (rt2d_classify_rays against RayFan.fused_mask), the side rays get rows of a compact
    gain array sized by a bound; with too few rows the flag behind the fan's summary sends the map through the waiting
    path.  Same map three ways."""
    import torch
    from ray_tracer_amd import rays
    g = common.load('c2_fan999')
    env = common.product_env(rt, g)
    kw = common.prop_kwargs(g)
    monkeypatch.setattr(rays, 'STREAM_MIN_RAYS', 100)

    def one(**hk):
        sim = rt.Simulation2D(env, g['fan_source'])
        fan = sim.cast(g['fan_angles'], **kw)
        pending = fan._pending
        H, _ = sim.heatmap_full(resolution=[50, 25], **hk)
        return fan, pending, H.cpu().numpy()

    fan_a, _, Ha = one(stream=False)                      # three kernels
    fan_b, pending, Hb = one()                            # one pass, queued without waiting
    assert pending and not fan_b._onepass_overflow and fan_b.G == {}
    n_side = fan_b.last_slow_rays
    assert 0 < n_side < fan_b.n // 4
    # the device classification is fused_mask
    m_host = fan_b.fused_mask([50, 25], Hb.shape, (fan_b.status == 0).to(torch.uint8))
    assert n_side == int(((fan_b.status == 0) & (m_host == 0)).sum().item())
    monkeypatch.setattr(rays, 'ONEPASS_MIN_ROWS', 1)
    monkeypatch.setattr(rays, 'ONEPASS_ROWS_DIV', 10 ** 9)
    fan_c, _, Hc = one()                                  # one row for all side rays: overflow -> the waiting path
    assert fan_c.last_slow_rays == n_side
    for Hx in (Hb, Hc):
        assert Hx.shape == Ha.shape and np.array_equal(Hx == 0, Ha == 0)
        both = np.isfinite(Ha) & (Ha != 0)
        assert (np.abs(Hx - Ha)[both] / np.abs(Ha[both])).max() < 1e-12
