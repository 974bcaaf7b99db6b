"""Heat-map presentation: tick mapping, floor-mask overlay and the three-channel image of the reference's notebook
(deprecated.ipynb cell 4).  Host-side NumPy on the finished (nz, nx) map -- cosmetic, O(map), nothing here is on the
device path.  ``heatmap_plot`` below is what ``Simulation2D.heatmap`` returns: rows are depth cells, columns are
distance cells."""

import numpy as np


def heatmap_ticks(shape, resolution, xticks_npoints=10, yticks_npoints=6):
    """Tick positions [cells] and labels [m] of a (nz, nx) map (deprecated.ipynb cell 4, "add tickmarks" block).
    Returns (xticks_heatmap, xticks_true, yticks_heatmap, yticks_true)."""
    xdim, ydim = int(shape[0]), int(shape[1])                     # the notebook's names: xdim = rows (depth)
    res = np.asarray(resolution)
    xticks_heatmap = np.linspace(0, ydim, xticks_npoints)
    yticks_heatmap = np.linspace(0, xdim, yticks_npoints)
    xticks_true = np.round(xticks_heatmap * res[0], 0).astype(int)
    yticks_true = np.round(yticks_heatmap * -1 * res[1], 0).astype(int)
    return xticks_heatmap, xticks_true, yticks_heatmap, yticks_true


def floor_mask(env, shape, resolution):
    """(nx, nz) boolean mask of the cells below the sea floor ("Plot ground" block of the same cell): cell (i, j) is
    masked when its depth ``-j * res_z`` lies under ``env.floor(i * res_x)``.  Columns beyond the floor table's range
    are left unmasked (the reference's interpolant would raise there)."""
    xdim, ydim = int(shape[0]), int(shape[1])
    res = np.asarray(resolution)
    x = np.arange(0, ydim, 1) * res[0]
    inside = (x >= env.range_min[0]) & (x <= env.range_max[0])
    z_floor = np.full(ydim, -np.inf)
    z_floor[inside] = np.asarray(env.floor(x[inside].astype(np.float64)), dtype=np.float64)
    z_width = np.arange(0, xdim, 1) * -1 * res[1]
    return z_width[None, :] < z_floor[:, None]


def heatmap_image(heatmap_plot, mask=None):
    """(nz, nx, 3) float image in 0..255 as the notebook builds it for cv2.imwrite: ``(map - min) / max`` (sic: the
    reference divides by the maximum, not by the span) on all three channels, and +255 on channel 0 where ``mask``
    (from floor_mask, (nx, nz)) is set."""
    hp = np.asarray(heatmap_plot, dtype=np.float64)
    norm = (hp - np.min(hp)) / np.max(hp)
    img = np.repeat((norm * 255)[:, :, None], 3, axis=2)
    if mask is not None:
        m = np.asarray(mask)
        if m.shape != hp.shape[::-1]:
            raise ValueError(f'floor mask {m.shape} does not match the transposed map {hp.shape[::-1]}')
        img[:, :, 0] += m.T * 255.0
    return img


def plot_heatmap(heatmap_plot, resolution, env=None, ax=None, **imshow_kwargs):
    """imshow of the map with metre ticks; with ``env`` the sea floor is painted over it.  Returns the axes."""
    import matplotlib.pyplot as plt
    ax = plt.gca() if ax is None else ax
    hp = np.asarray(heatmap_plot)
    if env is not None:
        hp = np.where(floor_mask(env, hp.shape, resolution).T, np.nan, hp)
    xt, xl, yt, yl = heatmap_ticks(hp.shape, resolution)
    ax.imshow(hp, **imshow_kwargs)
    ax.set_xticks(xt, xl)
    ax.set_yticks(yt, yl)
    ax.set_xlabel('Distance [m]')
    ax.set_ylabel('Altitude [m]')
    return ax
