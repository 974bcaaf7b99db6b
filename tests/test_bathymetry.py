"""Bathymetry map -> cut -> floor table (reference: topography/bathymetry.py:13-108; SURVEY 8 f-1, BASELINE config 3).

Golden: tests/golden/c3_cut.npz holds the cut the reference's own Map made of tests/golden/synth_bathy.asc (the real
.asc is absent from the reference, SURVEY trap 3; the grid is synthetic, written by oracle/gen_golden.py)."""

import os

import numpy as np
import pytest

import common

ASC = os.path.join(common.GOLD, 'synth_bathy.asc')


@pytest.fixture(scope='module')
def cut():
    return np.load(os.path.join(common.GOLD, 'c3_cut.npz'))


@pytest.mark.parametrize('backend', ['scipy', 'own'])
def test_cut_matches_the_reference(rt, cut, backend):
    m = rt.Map.from_asc(ASC, backend=backend)
    x, z = m.cut(tuple(cut['start']), tuple(cut['stop']), npoints=int(cut['npoints']))
    assert np.array_equal(x, cut['x'])                      # the reference's axis x = res * t, bit for bit
    err = np.abs(z - cut['z']).max()
    print(f'{backend}: max |z - reference| = {err:.3e} m')
    # scipy backend = the reference's own FITPACK spline -> identical; own spline: same interpolant, different solver
    assert err == 0.0 if backend == 'scipy' else err < 1e-8


def test_map_attributes_follow_the_reference(rt):
    m = rt.Map.from_asc(ASC)
    assert (m.shape_x, m.shape_y) == (96, 64) and m.shape == (96, 64)
    assert m.cellsize == 0.01
    assert np.allclose(m.llcorner_deg, [3.0, 41.4]) and np.allclose(m.urcorner_deg, [3.96, 42.04])
    assert m.res == 6378000 * np.deg2rad(0.01)
    # NODATA cells (the land patch) are NaN in map_array and filled with 0 for the interpolant
    assert np.isnan(m.map_array).sum() == 48
    # the interpolant goes through the samples: row 0 of the file is the NORTH edge (map.T[:, ::-1])
    grid = np.loadtxt(ASC, skiprows=6)
    lon = np.linspace(m.llcorner_rad[0], m.urcorner_rad[0], 96)
    lat = np.linspace(m.llcorner_rad[1], m.urcorner_rad[1], 64)
    got = m.z_from_rad(lon[40], lat[10])
    assert abs(float(np.ravel(got)[0]) - grid[63 - 10, 40]) < 1e-8


def test_own_spline_equals_scipy_on_a_grid(rt):
    a, b = rt.Map.from_asc(ASC, backend='scipy'), rt.Map.from_asc(ASC, backend='own')
    rng = np.random.default_rng(1)
    lon = rng.uniform(3.0, 3.96, 500)
    lat = rng.uniform(41.4, 42.04, 500)
    za = a._z_interp(np.deg2rad(lon), np.deg2rad(lat), grid=False)
    zb = b._z_interp(np.deg2rad(lon), np.deg2rad(lat), grid=False)
    assert np.abs(za - zb).max() < 1e-8


def test_distance_axis_and_floor_table(rt, cut):
    m = rt.Map.from_asc(ASC)
    start, stop, n = tuple(cut['start']), tuple(cut['stop']), int(cut['npoints'])
    xr, zr = m.cut(start, stop, npoints=n)
    xd, zd = m.cut(start, stop, npoints=n, axis='distance')
    assert np.array_equal(zr, zd)
    d = m.cut_length(start, stop)
    assert xd[0] == 0 and abs(xd[-1] - d) < 1e-9 and abs(xr[-1] * (n - 1) - d) < 1e-6    # x = res*t spans d/(n-1)
    tab = m.floor_table(start, stop, npoints=n)
    assert tab.order == 2 and tab.n_pieces == n - 1
    assert np.array_equal(tab(xd), zd)
    # the table drops into Environment2D unchanged (config 3: a fan over the cut)
    env = rt.Environment2D(tab, (np.array([0., d]), np.array([0., 0.])), np.array([0., np.floor(zd.min()) - 1.]), np.array([d, 1.]))
    assert env.floor is tab and np.array_equal(env.floor(xd), zd)


def test_from_asc_rejects_a_truncated_header(rt, tmp_path):
    p = tmp_path / 'bad.asc'
    p.write_text('NCOLS 4\nNROWS 4\n1 2 3 4\n')
    with pytest.raises(Exception):
        rt.Map.from_asc(str(p))


@pytest.mark.gpu
def test_fan_over_a_map_cut_on_the_gpu(rt, cut):
    """Config 3 end to end: .asc -> Map.cut -> floor table -> fan.  The golden fan c3_bathy999 was cast by the reference
    over this same cut stretched to 60 km; the environment built straight from the Map must reproduce the fan cast on
    the fixture's stored floor bit for bit, and hence the reference's terminations and rebound counts."""
    import torch
    g = common.load('c3_bathy999')
    m = rt.Map.from_asc(ASC)
    _, z = m.cut(tuple(cut['start']), tuple(cut['stop']), npoints=int(cut['npoints']))
    floor = rt.linear_interp_table(np.linspace(0., 60000., len(z)), z, oob=1)
    env = rt.Environment2D(floor, (g['env_ceil_x'], g['env_ceil_z']), g['env_range_min'], g['env_range_max'])
    kw = common.prop_kwargs(g)
    fan = rt.RayFan(env, g['fan_source'], g['fan_angles'], **kw)
    ref = rt.RayFan(common.product_env(rt, g), g['fan_source'], g['fan_angles'], **kw)
    assert torch.equal(fan.n_pts, ref.n_pts) and torch.equal(fan.X, ref.X) and torch.equal(fan.Z, ref.Z)
    ok = g['error'] == 0
    assert np.array_equal(fan.host('stop_code')[ok], g['stop_code'][ok])
    assert np.array_equal(fan.host('n_rebounds')[ok], g['n_rebounds'][ok])
