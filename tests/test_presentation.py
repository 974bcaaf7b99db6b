"""Heat-map presentation: ticks, floor mask, three-channel image.  The reference keeps this as a
commented notebook cell (deprecated.ipynb cell 4), so there is nothing to import: the expected values below are that
cell's NumPy expressions written out on a small synthetic map."""

import numpy as np

import common


def _env(rt):
    g = common.load('c2_fan999')
    return common.product_env(rt, g)


def test_ticks(rt):
    res = np.array([50, 25])
    xt, xl, yt, yl = rt.heatmap_ticks((51, 1471), res)
    assert np.array_equal(xt, np.linspace(0, 1471, 10)) and np.array_equal(yt, np.linspace(0, 51, 6))
    assert xl[0] == 0 and xl[-1] == 1471 * 50 and yl[-1] == -51 * 25
    assert xl.dtype.kind == 'i' and yl.dtype.kind == 'i'


def test_floor_mask_is_the_notebook_expression(rt):
    env = _env(rt)
    res = np.array([50, 25])
    xdim, ydim = 61, 1400                                   # rows (depth), columns (distance): inside the floor table
    m = rt.floor_mask(env, (xdim, ydim), res)
    z_floor = env.floor(np.arange(0, ydim, 1) * res[0] * 1.0)
    z_width = np.arange(0, xdim, 1) * -1 * res[1]
    want = (np.tile(z_width, (ydim, 1)).T < z_floor).T
    assert m.shape == (ydim, xdim) and np.array_equal(m, want)
    assert m.any() and not m.all()
    # beyond the table the columns stay unmasked instead of raising
    m2 = rt.floor_mask(env, (xdim, 1600), res)
    assert not m2[1501:].any() and np.array_equal(m2[:ydim], want)


def test_image(rt):
    rng = np.random.default_rng(0)
    hp = rng.random((7, 11)) + 0.25
    mask = rng.random((11, 7)) < 0.3
    img = rt.heatmap_image(hp, mask)
    norm = (hp - np.min(hp)) / np.max(hp)
    ch = np.transpose(np.tile(norm * 255, (3, 1, 1)), (1, 2, 0))
    overlay = np.transpose(np.tile(mask * 255, (3, 1, 1)), (2, 1, 0)).astype(float)
    overlay[:, :, 1:] = 0
    assert np.array_equal(img, ch + overlay)
    assert np.array_equal(rt.heatmap_image(hp), ch)
