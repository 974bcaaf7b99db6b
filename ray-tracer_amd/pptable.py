"""Piecewise-polynomial tables: the one data format every device kernel reads.

The reference evaluates its fields through ``scipy.interpolate.interp1d`` objects
(quadratic: physics_environment.py:71-86, environment.py:50-51; linear: profile_ph.py:27-33)
and through user callables (floor / ceiling, ray.py:172-173).  None of those can run on a
GPU, so every field is converted ONCE, on the host, into a :class:`PPTable`:

    piece j covers  breaks[j] <= x < breaks[j+1]
    value(x) = (c0*d + c1)*d + c2     with d = x - breaks[j]      (order 3, quadratic)
    value(x) =  c0*d + c1             with d = x - breaks[j]      (order 2, linear)

evaluated without FMA contraction.  The HIP kernels (csrc/rt2d_device.h) evaluate exactly this
expression, so host ``PPTable.__call__`` and device lookups agree bit for bit.

``quadratic_interp_table`` builds the same spline as ``interp1d(kind='quadratic')``
(= ``make_interp_spline(k=2)``: knots at the mid-points of the data sites, not-a-knot ends)
.  The collocation solve is delegated to scipy when it is importable (the reference itself needs
scipy), so the B-spline coefficients are the reference's to the last bit -- this matters because the ray
dynamics amplify table differences by up to 1e6 ("Sensitivity") -- and falls back to the
tridiagonal solve written here otherwise.  Only the power-basis conversion and the evaluation are ours.
"""

import numpy as np

try:                                        # host-side table construction only; never on the compute path
    from scipy.interpolate import make_interp_spline as _make_interp_spline
except ImportError:                         # pragma: no cover
    _make_interp_spline = None

OOB_EXTRAPOLATE = 0   # continue the end pieces (interp1d fill_value='extrapolate')
OOB_CLAMP = 1         # clamp x into [breaks[0], breaks[-1]] (np.interp-style end values)
OOB_ERROR = 2         # out of range is an error (interp1d bounds_error=True)


class PPTable:
    """Callable piecewise polynomial of order 2 (linear) or 3 (quadratic)."""

    def __init__(self, breaks, coef, oob=OOB_EXTRAPOLATE):
        self.breaks = np.ascontiguousarray(breaks, dtype=np.float64)
        self.coef = np.ascontiguousarray(coef, dtype=np.float64)
        if self.coef.ndim != 2 or self.coef.shape[1] not in (2, 3):
            raise ValueError('coef must be (n_pieces, 2|3)')
        if self.breaks.ndim != 1 or self.breaks.shape[0] != self.coef.shape[0] + 1:
            raise ValueError('breaks must hold n_pieces + 1 values')
        if self.coef.shape[0] < 1:
            raise ValueError('a table needs at least one piece')
        if not np.all(np.diff(self.breaks) > 0):
            raise ValueError('breaks must be strictly increasing')
        self.order = int(self.coef.shape[1])
        self.n_pieces = int(self.coef.shape[0])
        self.oob = int(oob)
        # uniform-grid hint used by the device for an O(1) piece guess (always verified against breaks)
        self.x0 = float(self.breaks[0])
        self.inv_h = float(self.n_pieces / (self.breaks[-1] - self.breaks[0]))

    # -- evaluation -----------------------------------------------------------------------
    def piece_index(self, x):
        x = np.asarray(x, dtype=np.float64)
        j = np.searchsorted(self.breaks, x, side='right') - 1
        return np.clip(j, 0, self.n_pieces - 1)

    def in_range(self, x):
        x = np.asarray(x, dtype=np.float64)
        return (x >= self.breaks[0]) & (x <= self.breaks[-1])

    def __call__(self, x):
        x_in = x
        x = np.asarray(x, dtype=np.float64)
        if self.oob == OOB_ERROR and not np.all(self.in_range(x)):
            raise ValueError('A value in x_new is outside the interpolation range.')
        if self.oob == OOB_CLAMP:
            x = np.clip(x, self.breaks[0], self.breaks[-1])
        j = self.piece_index(x)
        d = x - self.breaks[j]
        c = self.coef[j]
        if self.order == 3:
            y = (c[..., 0] * d + c[..., 1]) * d + c[..., 2]
        else:
            y = c[..., 0] * d + c[..., 1]
        if np.ndim(x_in) == 0:
            return np.float64(y)
        return y

    def __repr__(self):
        return f'PPTable(order={self.order}, n_pieces={self.n_pieces}, range=[{self.breaks[0]}, {self.breaks[-1]}], oob={self.oob})'


# -- builders -----------------------------------------------------------------------------

def linear_interp_table(x, y, oob=OOB_CLAMP):
    """Piecewise-linear interpolant through (x, y): value = slope_j * (x - x_j) + y_j."""
    x = np.asarray(x, dtype=np.float64)
    y = np.asarray(y, dtype=np.float64)
    if x.shape[0] < 2:
        raise ValueError('need at least two samples')
    slope = (y[1:] - y[:-1]) / (x[1:] - x[:-1])
    coef = np.stack((slope, y[:-1]), axis=1)
    return PPTable(x, coef, oob=oob)


def _solve_tridiagonal(lower, diag, upper, rhs):
    """Thomas algorithm; rhs may be (n,) or (n, m).  The collocation matrix of an interpolating
    quadratic spline is totally positive, so elimination without pivoting is stable."""
    n = diag.shape[0]
    d = diag.astype(np.float64).copy()
    b = np.array(rhs, dtype=np.float64, copy=True)
    for i in range(1, n):
        w = lower[i] / d[i - 1]
        d[i] -= w * upper[i - 1]
        b[i] -= w * b[i - 1]
    out = np.empty_like(b)
    out[n - 1] = b[n - 1] / d[n - 1]
    for i in range(n - 2, -1, -1):
        out[i] = (b[i] - upper[i] * out[i + 1]) / d[i]
    return out


def quadratic_spline_pieces(x, y):
    """Breaks and power-basis coefficients of the C1 quadratic spline interpolating (x, y).

    Same spline as scipy's interp1d(kind='quadratic') (reference: physics_environment.py:71-86):
    B-spline knots  t = [x0]*3 + [(x_i + x_{i+1})/2 for i = 1..n-3] + [x_{n-1}]*3,  n coefficients
    fixed by collocation at the n data sites.  ``y`` may be (n,) or (n, m) (m independent curves).
    Returns breaks (n-1,), coef (n-2, 3) or (n-2, 3, m).
    """
    x = np.asarray(x, dtype=np.float64)
    y = np.asarray(y, dtype=np.float64)
    n = x.shape[0]
    if n < 3:
        raise ValueError('quadratic interpolation needs at least 3 samples')
    if not np.all(np.diff(x) > 0):
        raise ValueError('x must be strictly increasing')
    mid = 0.5 * (x[1:] + x[:-1])
    t = np.concatenate((np.full(3, x[0]), mid[1:-1], np.full(3, x[-1])))   # n + 3 knots

    # Collocation: data site i lies in knot span m = clip(i + 1, 2, n - 1); the three quadratic
    # B-splines alive there are B_{m-2}, B_{m-1}, B_m (Cox-de Boor, written out for degree 2).
    lower = np.zeros(n)
    diag = np.zeros(n)
    upper = np.zeros(n)
    for i in range(n):
        m = min(max(i + 1, 2), n - 1)
        xi = x[i]
        # degree-1 pair on span m
        l1 = xi - t[m]
        r1 = t[m + 1] - xi
        b0 = r1 / (r1 + l1)
        b1 = l1 / (r1 + l1)
        # degree-2 triple
        l2 = xi - t[m - 1]
        r2 = t[m + 2] - xi
        n0 = b0 * r1 / (r1 + l2)
        n1 = b0 * l2 / (r1 + l2) + b1 * r2 / (r2 + l1)
        n2 = b1 * l1 / (r2 + l1)
        cols = (m - 2, m - 1, m)
        vals = (n0, n1, n2)
        for cidx, v in zip(cols, vals):
            off = cidx - i
            if off == -1:
                lower[i] = v
            elif off == 0:
                diag[i] = v
            elif off == 1:
                upper[i] = v
            elif v != 0.0:   # only at the two end rows, where the out-of-band weight is exactly 0
                raise AssertionError('collocation matrix is not tridiagonal')
    c = None
    if _make_interp_spline is not None:
        spl = _make_interp_spline(x, y, k=2, check_finite=False)
        if spl.t.shape == t.shape and np.array_equal(spl.t, t):
            c = np.asarray(spl.c, dtype=np.float64)
    if c is None:
        c = _solve_tridiagonal(lower, diag, upper, y)

    # Power basis about the left break of each span m = 2 .. n-1 (piece j = m - 2).
    m = np.arange(2, n)
    tm1, tm, tp1, tp2 = t[m - 1], t[m], t[m + 1], t[m + 2]
    shape = (-1,) + (1,) * (y.ndim - 1)
    w_l = ((tp1 - tm) / (tp1 - tm1)).reshape(shape)
    w_r = ((tm - tm1) / (tp1 - tm1)).reshape(shape)
    val = c[m - 2] * w_l + c[m - 1] * w_r
    d_lo = 2.0 * (c[m - 1] - c[m - 2]) / (tp1 - tm1).reshape(shape)
    d_hi = 2.0 * (c[m] - c[m - 1]) / (tp2 - tm).reshape(shape)
    quad = 0.5 * (d_hi - d_lo) / (tp1 - tm).reshape(shape)
    breaks = t[2:n + 1]
    coef = np.stack((quad, d_lo, val), axis=1)
    return breaks, coef


def quadratic_interp_table(x, y, oob=OOB_EXTRAPOLATE):
    breaks, coef = quadratic_spline_pieces(x, y)
    return PPTable(breaks, coef, oob=oob)


def tabulate_callable(fn, x_min, x_max, n_samples, oob=OOB_CLAMP):
    """Sample a user callable on a uniform grid and return its piecewise-linear table.

    This is the tabulation contract for floor / ceiling callables (reference: ray.py:172-173 calls
    them per step in Python): the device sees the table, and the table object itself is a callable
    that can be handed to any host code (plots, the reference implementation) unchanged.
    """
    xs = np.linspace(x_min, x_max, int(n_samples))
    zs = np.asarray(fn(xs), dtype=np.float64)
    if zs.ndim == 0:
        zs = np.full_like(xs, float(zs))
    if zs.shape != xs.shape:
        zs = np.array([float(fn(v)) for v in xs], dtype=np.float64)
    return linear_interp_table(xs, zs, oob=oob)
