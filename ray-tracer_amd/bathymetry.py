"""Bathymetry maps: ESRI ASCII grid -> great-circle cut -> sea-floor table for the stepping kernel.

Host-side mirror of the reference's ``topography/bathymetry.py`` ``Map`` (bathymetry.py:13-121): same
constructor, ``from_asc``, ``cut(start, stop, npoints=100)``, ``z_from_rad`` / ``z_from_deg``, ``display`` and
attributes.  This is one-off O(npoints) work per environment and stays on the host; what reaches the GPU is
the piecewise-linear floor table built from the cut (``Map.floor_table`` -> ``Environment2D(floor=table, ...)``),
staged in LDS by the stepping kernel like any other floor.

Quirks of the reference that are reproduced, not fixed:
  * the grid is stored transposed with the rows flipped (``map.T[:, ::-1]``, bathymetry.py:24) and NODATA cells are
    filled with 0 before interpolation (bathymetry.py:41-42);
  * the longitude / latitude axes run from the lower-left corner to ``ll + shape * cellsize`` in ``shape`` samples
    (bathymetry.py:35-40), so the sample spacing is cellsize * n / (n - 1), not cellsize;
  * the Cartesian conversion treats the latitude as a colatitude (``sin(lat) cos(lon), sin(lat) sin(lon),
    cos(lat)``, bathymetry.py:84-89), which changes the angle between the two end points;
  * ``EARTH_RADIUS_AVG = 6378000`` (bathymetry.py:10);
  * the SLERP weights are applied to the (lon, lat) pairs, not to unit vectors (bathymetry.py:99-102);
  * the returned abscissa is ``x = res * t`` with ``res = d / (npoints - 1)`` (bathymetry.py:96,105): it spans
    0 .. d / (npoints - 1), not 0 .. d.  ``cut(..., axis='distance')`` is the build's corrected axis ``x = d * t``;
    the default ``axis='reference'`` is the reference's.

The bicubic interpolant is scipy's ``RectBivariateSpline`` when scipy is importable (the reference's own object,
so the cut is the reference's to the last bit) and otherwise the same spline built here: tensor-product cubic
B-spline, not-a-knot knots (FITPACK's choice for an interpolating spline, s = 0), collocation solved by banded
elimination.  ``Map(..., backend='own')`` forces the latter (tests compare both).
"""

import numpy as np

try:                                        # host-side table construction only; never on the compute path
    from scipy.interpolate import RectBivariateSpline as _RectBivariateSpline
except ImportError:                         # pragma: no cover
    _RectBivariateSpline = None

from .pptable import linear_interp_table, OOB_CLAMP

EARTH_RADIUS_AVG = 6378000  # in meters (bathymetry.py:10)


# -- interpolating cubic B-spline (not-a-knot), numpy only ------------------------------------------------

def _knots_not_a_knot(x, k=3):
    """FITPACK's knots for an interpolating spline of degree k through all of x (s = 0)."""
    n = x.shape[0]
    if n < k + 1:
        raise ValueError(f'bicubic interpolation needs at least {k + 1} samples per axis')
    interior = x[(k + 1) // 2:n - (k + 1) // 2]
    return np.concatenate((np.full(k + 1, x[0]), interior, np.full(k + 1, x[-1])))


def _find_span(t, k, n, x):
    """Knot span m with t[m] <= x < t[m+1], clipped to [k, n-1] (n = number of coefficients)."""
    m = np.searchsorted(t, x, side='right') - 1
    return np.clip(m, k, n - 1)


def _basis(t, k, m, x):
    """The k+1 B-splines of degree k alive on span m at x (Cox-de Boor).  x, m: arrays (p,).  Returns (p, k+1)."""
    x = np.asarray(x, dtype=np.float64)
    p = x.shape[0]
    N = np.zeros((p, k + 1))
    N[:, 0] = 1.0
    left = np.zeros((p, k + 1))
    right = np.zeros((p, k + 1))
    for j in range(1, k + 1):
        left[:, j] = x - t[m + 1 - j]
        right[:, j] = t[m + j] - x
        saved = np.zeros(p)
        for r in range(j):
            den = right[:, r + 1] + left[:, j - r]
            temp = np.where(den != 0.0, N[:, r] / np.where(den != 0.0, den, 1.0), 0.0)
            N[:, r] = saved + right[:, r + 1] * temp
            saved = left[:, j - r] * temp
        N[:, j] = saved
    return N


def _solve_banded(A_rows, cols0, rhs, bw):
    """Gaussian elimination without pivoting of a banded system given row-wise: row i holds A_rows[i, :] at columns
    cols0[i] .. cols0[i] + width - 1.  The collocation matrix of an interpolating B-spline is totally positive, so no
    pivoting is needed.  rhs: (n, m).  Works on a dense band of half-width bw."""
    n, width = A_rows.shape
    W = 2 * bw + 1
    band = np.zeros((n, W))                     # band[i, bw + (j - i)] = A[i, j]
    for i in range(n):
        for q in range(width):
            j = cols0[i] + q
            v = A_rows[i, q]
            if v != 0.0:
                off = j - i
                if abs(off) > bw:
                    raise AssertionError('collocation matrix wider than its band')
                band[i, bw + off] = v
    b = np.array(rhs, dtype=np.float64, copy=True)
    for i in range(n):
        piv = band[i, bw]
        for r in range(i + 1, min(n, i + bw + 1)):
            f = band[r, bw + (i - r)]
            if f == 0.0:
                continue
            w = f / piv
            lo = i - r                          # offsets of row r that overlap row i's [0, bw] offsets
            for off in range(0, bw + 1):
                j = i + off
                if j >= n:
                    break
                band[r, bw + (j - r)] -= w * band[i, bw + off]
            b[r] -= w * b[i]
    out = np.empty_like(b)
    for i in range(n - 1, -1, -1):
        acc = b[i].copy()
        for off in range(1, bw + 1):
            j = i + off
            if j >= n:
                break
            acc -= band[i, bw + off] * out[j]
        out[i] = acc / band[i, bw]
    return out


def _interp_coefs(x, Y, k=3):
    """B-spline coefficients (n, m) of the degree-k not-a-knot spline through (x, Y[:, j]) for every column j."""
    x = np.asarray(x, dtype=np.float64)
    n = x.shape[0]
    t = _knots_not_a_knot(x, k)
    m = _find_span(t, k, n, x)
    N = _basis(t, k, m, x)                      # row i: weights of coefficients m[i]-k .. m[i]
    return t, _solve_banded(N, m - k, np.asarray(Y, dtype=np.float64).reshape(n, -1), bw=k)


class _BicubicOwn:
    """Tensor-product interpolating cubic spline z(x, y) on a rectangular grid (what RectBivariateSpline(x, y, z)
    with its defaults kx = ky = 3, s = 0 represents)."""

    def __init__(self, x, y, z):
        self.k = 3
        z = np.asarray(z, dtype=np.float64)
        self.tx, cx = _interp_coefs(x, z, self.k)                 # along x for every y column: (nx, ny)
        self.ty, cy = _interp_coefs(y, cx.T, self.k)              # then along y: (ny, nx)
        self.c = np.ascontiguousarray(cy.T)                       # (nx, ny)
        self.nx, self.ny = self.c.shape
        self.x0, self.x1, self.y0, self.y1 = float(x[0]), float(x[-1]), float(y[0]), float(y[-1])

    def __call__(self, xq, yq, grid=True):
        xq = np.atleast_1d(np.asarray(xq, dtype=np.float64))
        yq = np.atleast_1d(np.asarray(yq, dtype=np.float64))
        if grid:
            X, Y = np.meshgrid(xq, yq, indexing='ij')
            return self(X.ravel(), Y.ravel(), grid=False).reshape(xq.shape[0], yq.shape[0])
        # FITPACK clamps the arguments into the grid rectangle before evaluating (fpbisp)
        xq = np.clip(xq, self.x0, self.x1)
        yq = np.clip(yq, self.y0, self.y1)
        k = self.k
        mx = _find_span(self.tx, k, self.nx, xq)
        my = _find_span(self.ty, k, self.ny, yq)
        Bx = _basis(self.tx, k, mx, xq)                           # (p, 4)
        By = _basis(self.ty, k, my, yq)
        out = np.zeros(xq.shape[0])
        for a in range(k + 1):
            row = np.zeros(xq.shape[0])
            for b in range(k + 1):
                row += By[:, b] * self.c[mx - k + a, my - k + b]
            out += Bx[:, a] * row
        return out


class Map:

    def __init__(self, map, xllcorner, yllcorner, cellsize, backend=None):
        """
        :param map: depth grid as read from the file, rows from north to south (NaN = no data)
        :param xllcorner: lower-left corner longitude (degrees)
        :param yllcorner: lower-left corner latitude (degrees)
        :param cellsize: cell size (degrees)
        :param backend: 'scipy' | 'own' | None (scipy when importable) -- build-specific
        """
        map = np.asarray(map, dtype=np.float64)
        self.map_array = map.T[:, ::-1]
        self._map_nan = np.isnan(self.map_array)
        self.cellsize = cellsize
        self.res = EARTH_RADIUS_AVG * np.deg2rad(self.cellsize)          # map resolution in metres
        self.shape_y, self.shape_x = map.shape
        self.shape = self.map_array.shape

        self.llcorner_deg = np.array([xllcorner, yllcorner])
        self.llcorner_rad = np.deg2rad(self.llcorner_deg)
        self.urcorner_deg = self.llcorner_deg + np.array([self.shape_x, self.shape_y], dtype=float) * cellsize
        self.urcorner_rad = np.deg2rad(self.urcorner_deg)

        self._x = np.linspace(self.llcorner_rad[0], self.urcorner_rad[0], self.shape_x)   # longitudes (rad)
        self._y = np.linspace(self.llcorner_rad[1], self.urcorner_rad[1], self.shape_y)   # latitudes (rad)
        self._z = self.map_array.copy()
        self._z[self._map_nan] = 0.
        if backend is None:
            backend = 'scipy' if _RectBivariateSpline is not None else 'own'
        if backend == 'scipy':
            if _RectBivariateSpline is None:
                raise ImportError('backend="scipy" needs scipy')
            self._z_interp = _RectBivariateSpline(self._x, self._y, self._z)
        elif backend == 'own':
            self._z_interp = _BicubicOwn(self._x, self._y, self._z)
        else:
            raise ValueError(backend)
        self.backend = backend
        self.z_from_rad = lambda phi, theta: self._z_interp(phi, theta)
        self.z_from_deg = lambda long, lat: self._z_interp(np.deg2rad(long), np.deg2rad(lat))

    def __repr__(self):
        return f'map with llcorner: {self.llcorner_deg} and urcorner: {self.urcorner_deg}.'

    @classmethod
    def from_asc(cls, path, **kwargs):
        """ESRI ASCII grid: six header lines (NCOLS, NROWS, XLLCORNER, YLLCORNER, CELLSIZE, NODATA_VALUE), then the
        rows (bathymetry.py:51-67)."""
        header_len = 6
        header = dict()
        with open(path, 'r') as f:
            for _ in range(header_len):
                key, val = f.readline().strip().split()
                header[key] = int(val) if key in ('NCOLS', 'NROWS') else float(val)
        grid = np.loadtxt(path, skiprows=header_len, ndmin=2)
        grid[grid == header['NODATA_VALUE']] = np.nan
        return cls(grid, header['XLLCORNER'], header['YLLCORNER'], header['CELLSIZE'], **kwargs)

    def cut(self, start, stop, npoints=100, axis='reference'):
        """Depth profile along the path from ``start`` to ``stop`` ((long, lat) in degrees) (bathymetry.py:69-108).

        :param axis: 'reference' -> x = res * t as the reference returns it (spans 0 .. d / (npoints - 1));
                     'distance'  -> x = d * t, metres along the path (build-specific)
        Returns (x, z).
        """
        ends = np.deg2rad(np.asarray([start, stop], dtype=np.float64))          # rows: (long, lat) in radians

        def on_sphere(q):
            # the reference's (non-standard) use of the latitude as polar angle, kept for its numbers
            return np.array([np.sin(q[1]) * np.cos(q[0]), np.sin(q[1]) * np.sin(q[0]), np.cos(q[1])])

        theta = np.arccos(np.dot(on_sphere(ends[0]), on_sphere(ends[1])))        # angle between the two end points
        d = theta * EARTH_RADIUS_AVG                                             # path length in metres
        res = d / (npoints - 1)

        # spherical linear interpolation of the (long, lat) pairs, weights sin(theta (1 - t)) and sin(theta t)
        t = np.linspace(0, 1, npoints)
        w_start = np.sin(theta * (1 - t))[:, None]
        w_stop = np.sin(theta * t)[:, None]
        p = (ends[0] * w_start + ends[1] * w_stop) / np.sin(theta)

        if axis == 'reference':
            x = res * t
        elif axis == 'distance':
            x = d * t
        else:
            raise ValueError(axis)
        z = self._z_interp(p[:, 0], p[:, 1], grid=False)
        return x, z

    def cut_length(self, start, stop):
        """Length [m] of the path ``cut`` follows (the reference's d = theta * EARTH_RADIUS_AVG, bathymetry.py:95)."""
        x, _ = self.cut(start, stop, npoints=2, axis='distance')
        return float(x[-1])

    def floor_table(self, start, stop, npoints=100, axis='distance'):
        """Piecewise-linear sea-floor table z(x) along a cut, ready for ``Environment2D(floor=...)``: the device sees
        exactly these ``npoints`` samples (no Python callable in between)."""
        x, z = self.cut(start, stop, npoints=npoints, axis=axis)
        return linear_interp_table(x, z, oob=OOB_CLAMP)

    def display(self, ax=None, cmap='gray_r', nodata_color=(0.6, 0.3, 0.)):
        """Depth picture of the grid, north up: shallow is light, deep is dark, cells without data in
        ``nodata_color`` (what the reference's display shows; drawn here as one masked image with a colour map)."""
        import matplotlib.pyplot as plt
        from matplotlib import colormaps
        depth = np.ma.masked_invalid(np.abs(self.map_array.T[::-1]))            # back to file order: rows north to south
        shades = colormaps[cmap].copy()
        shades.set_bad(nodata_color)
        ax = plt.gca() if ax is None else ax
        return ax.imshow(depth, cmap=shades, vmin=0., vmax=float(depth.max()))
