"""CPU oracle for the 2D ray path -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the
product package (ray-tracer_amd/) never does.

What it is: a restatement of the reference's algorithm (ahernot/ray-tracer) for this path:
  * environment tables are built the way the reference builds them -- scipy ``interp1d`` objects over
    NumPy physics (physics_environment.py:61-86, environment.py:43-53, physics/*.py) -- and then
    converted to power-basis pieces with ``PPoly.from_spline`` so that C can evaluate them;
  * the per-ray loop, post-pass, gain and heat-map are the scalar C of ``rt2d_oracle.c``.
It is pinned against the reference itself by the fixtures under tests/golden/ (made by
oracle/gen_golden.py, which imports /root/reference in the build container).
"""

import ctypes as C
import os
import subprocess

import numpy as np
from scipy import interpolate

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, 'librt2d_oracle.so')
_SRC = os.path.join(_HERE, 'rt2d_oracle.c')

STOP_NAMES = [None, 'exit-xmin', 'exit-xmax', 'exit-zmin', 'exit-zmax', 'max-rebounds', 'backprop', 'max-iter',
              'power-cutoff']


def compile(force=False):
    """gcc -O2 -ffp-contract=off; idempotent."""
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(_SRC):
        subprocess.check_call(['gcc', '-O2', '-ffp-contract=off', '-fno-fast-math', '-fPIC', '-shared',
                               '-o', _SO, _SRC, '-lm'])
    return _SO


# ------------------------------------------------------------------------------------------------
# ctypes mirror of the C structs
class OrcTable(C.Structure):
    _fields_ = [('order', C.c_int32), ('n_pieces', C.c_int32), ('oob', C.c_int32), ('reserved', C.c_int32),
                ('breaks', C.POINTER(C.c_double)), ('coef', C.POINTER(C.c_double))]


class OrcEnv(C.Structure):
    _fields_ = [('range_min', C.c_double * 2), ('range_max', C.c_double * 2),
                ('c', OrcTable), ('dz_c', OrcTable), ('floor', OrcTable), ('ceil', OrcTable),
                ('dx_floor', OrcTable), ('dx_ceil', OrcTable), ('impedance', OrcTable),
                ('absorb', OrcTable * 6)]


class OrcParams(C.Structure):
    _fields_ = [('dx_max', C.c_double), ('dz_max', C.c_double),
                ('n_steps_max', C.c_int32), ('n_rebounds_max', C.c_int32),
                ('backprop', C.c_int32), ('has_target', C.c_int32),
                ('target', C.c_double * 2),
                ('rebound_cap', C.c_int32), ('reserved', C.c_int32),
                ('power_cutoff', C.c_double), ('cutoff_freq', C.c_double)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        compile()
        _lib = C.CDLL(_SO)
        dp = C.POINTER(C.c_double)
        ip = C.POINTER(C.c_int32)
        _lib.orc_propagate.restype = C.c_int32
        _lib.orc_propagate.argtypes = [C.POINTER(OrcEnv), dp, C.c_double, C.POINTER(OrcParams),
                                       C.c_int32, C.c_double, C.c_double, dp, dp, C.c_int64,
                                       ip, ip, ip, ip, C.POINTER(C.c_uint8), dp, dp, dp, dp, dp]
        _lib.orc_post.restype = None
        _lib.orc_post.argtypes = [C.POINTER(OrcEnv), dp, dp, C.c_int32, dp, dp, dp, dp, dp, dp, dp, ip, dp, dp]
        _lib.orc_gain.restype = None
        _lib.orc_gain.argtypes = [C.POINTER(OrcEnv), dp, dp, C.c_int32, C.c_double, C.c_int32, ip,
                                  C.POINTER(C.c_uint8), dp, dp, dp, C.c_double, C.c_double, dp]
        _lib.orc_heatmap_add.restype = C.c_int64
        _lib.orc_heatmap_add.argtypes = [dp, dp, dp, C.c_int32, ip, ip, dp]
        _lib.orc_heatmap_finalize.restype = None
        _lib.orc_heatmap_finalize.argtypes = [dp, ip, C.c_double, C.c_double, dp]
        _lib.orc_table_eval.restype = C.c_double
        _lib.orc_table_eval.argtypes = [C.POINTER(OrcTable), C.c_double, ip]
        _lib.orc_run_fan.restype = C.c_int64
        _lib.orc_run_fan.argtypes = [C.POINTER(OrcEnv), dp, dp, C.c_int64, C.POINTER(OrcParams), C.c_double,
                                     C.c_double, C.c_double, ip, ip, dp, dp, C.POINTER(C.c_int64)]
        _lib.orc_check_div_by_const.restype = C.c_int64
        _lib.orc_check_div_by_const.argtypes = [C.c_double, C.c_int64, C.c_uint64]
        _lib.orc_absorption.restype = C.c_double
        _lib.orc_absorption.argtypes = [C.POINTER(OrcEnv), C.c_double, C.c_double]
    return _lib


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32))


# ------------------------------------------------------------------------------------------------
# Floor / ceiling as a piecewise-linear table that is ALSO a Python callable.  The same object is
# handed to the reference when golden vectors are generated, so both sides see one floor.
class LinearFloor:
    def __init__(self, xs, zs):
        self.xs = np.ascontiguousarray(xs, dtype=np.float64)
        self.zs = np.ascontiguousarray(zs, dtype=np.float64)
        self.slope = (self.zs[1:] - self.zs[:-1]) / (self.xs[1:] - self.xs[:-1])

    def __call__(self, x):
        xa = np.clip(np.asarray(x, dtype=np.float64), self.xs[0], self.xs[-1])
        j = np.clip(np.searchsorted(self.xs, xa, side='right') - 1, 0, self.xs.shape[0] - 2)
        y = self.slope[j] * (xa - self.xs[j]) + self.zs[j]
        return np.float64(y) if np.ndim(x) == 0 else y


# ------------------------------------------------------------------------------------------------
# Physics restated in the reference's own evaluation order (host NumPy + scipy, as the reference).
_T_PROFILE = np.array([[0, 22], [-100, 21], [-200, 20], [-300, 16], [-400, 10], [-500, 4], [-600, 3.5], [-700, 3],
                       [-800, 3], [-900, 3], [-1000, 3], [-1100, 3], [-1200, 3], [-1300, 3], [-1400, 3],
                       [-1500, 3], [-10000, 3]], dtype=float)                       # profile_temperature.py:10-28
_PH_PROFILE = np.array([[0, 7.98], [-100, 7.88], [-200, 7.84], [-250, 7.87], [-300, 7.85], [-400, 7.79],
                        [-450, 7.7], [-500, 7.7], [-600, 7.71], [-800, 7.73], [-1000, 7.72], [-2000, 7.73],
                        [-3000, 7.74], [-10000, 7.74]], dtype=float)                # profile_ph.py:10-25
_calc_T = interpolate.interp1d(_T_PROFILE.T[0][::-1], _T_PROFILE.T[1][::-1], kind='quadratic',
                               bounds_error=False, fill_value='extrapolate')        # profile_temperature.py:30-36
_calc_pH = interpolate.interp1d(_PH_PROFILE.T[0][::-1], _PH_PROFILE.T[1][::-1], kind='linear',
                                bounds_error=False, fill_value='extrapolate')       # profile_ph.py:27-33
ZP1 = float(np.mean([2390000, 2550000, 2660000, 2680000, 3160000]))                # profile_impedance.py:17-18
ZP2 = 17000000.0                                                                   # profile_impedance.py:19


def _medwin(z, S, T):                                                              # model_velocity.py:53-56
    return 1449.2 + 4.6 * T - 0.055 * np.power(T, 2) + 0.00029 * np.power(T, 3) + (1.34 - 0.01 * T) * (S - 35) - 0.016 * z


def _rho(z, S, T, P):                                                              # model_rho.py:19-47
    rho_W = 999.842594 + 6.793952e-2 * T - 9.09529e-3 * np.power(T, 2) + 1.001685e-4 * np.power(T, 3) - 1.120083e-6 * np.power(T, 4) + 6.536336e-9 * np.power(T, 5)
    A = 8.24493e-1 - 4.0899e-3 * T + 7.6438e-5 * np.power(T, 2) - 8.2467e-7 * np.power(T, 3) + 5.3875e-9 * np.power(T, 4)
    B = -5.72466e-3 + 1.0227e-4 * T - 1.6546e-6 * np.power(T, 2)
    Cc = 4.8314e-4
    rho_1atm = rho_W + A * S + B * np.power(S, 1.5) + Cc * np.power(S, 2)
    K_W = 19652.21 + 148.4206 * T - 2.327105 * np.power(T, 2) + 1.360477e-2 * np.power(T, 3) - 5.155288e-5 * np.power(T, 4)
    A_W = 3.239908 + 1.43713e-3 * T + 1.16092e-4 * np.power(T, 2) - 5.79905e-7 * np.power(T, 3)
    B_W = 8.50935e-5 - 6.12293e-6 * T + 5.2787e-8 * np.power(T, 2)
    K_1atm = K_W + (54.6746 - 0.603459 * T + 1.09987e-2 * np.power(T, 2) - 6.1670e-5 * np.power(T, 3)) * S + (7.944e-2 + 1.6483e-2 * T - 5.3009e-4 * np.power(T, 2)) * np.power(S, 1.5)
    A = A_W + (2.2838e-3 - 1.0981e-5 * T - 1.6078e-6 * np.power(T, 2)) * S + 1.91075e-3 * np.power(S, 1.5)
    B = B_W + (-9.9348e-7 + 2.0816e-8 * T + 9.1697e-10 * np.power(T, 2)) * S
    K = K_1atm + A * P + B * np.power(P, 2)
    return rho_1atm / (1 - P / K)


def _absorb_coefs(z, T, S, pH):                                                    # model_absorption.py:8-16
    return np.stack((0.78 * np.sqrt(S / 35) * np.exp(T / 26), 42 * np.exp(T / 17), 0.52 * (1 + T / 43) * (S / 35),
                     np.exp((pH - 8) / 0.56), np.exp(z / 6000), np.exp(z / 17000 - T / 27)))


def _pp_from_interp1d(f):
    """Power-basis pieces of a scipy quadratic interp1d (its ._spline is a k=2 BSpline)."""
    spl = f._spline
    assert spl.k == 2
    breaks = np.unique(spl.t)
    left, mid = breaks[:-1], 0.5 * (breaks[:-1] + breaks[1:])
    c2 = spl(left)                                   # value at the left break (right-continuous)
    c1 = spl.derivative(1)(left)                     # slope there (the spline is C1)
    c0 = spl.derivative(2)(mid) / 2                  # piecewise-constant curvature
    return breaks, np.stack((c0, c1, c2), axis=1)    # (n_pieces, 3[, m])


class OracleEnv:
    """Tables + ctypes struct for one environment.

    floor / ceil: LinearFloor objects (the callables the reference would be given).
    """

    def __init__(self, floor, ceil, range_min, range_max, res_x=100, res_z=10):
        self.floor, self.ceil = floor, ceil
        self.range_min = np.asarray(range_min, dtype=float)
        self.range_max = np.asarray(range_max, dtype=float)
        self.res_x, self.res_z = res_x, res_z
        size = self.range_max - self.range_min
        n_x = int(size[0] // res_x) + 1                                             # physics_environment.py:53-56
        n_z = int(size[1] // res_z) + 1
        self.x = np.linspace(self.range_min[0], self.range_max[0], n_x)
        self.z = z = np.linspace(self.range_min[1], self.range_max[1], n_z)
        S = np.ones_like(z) * 35                                                    # profile_salinity.py:9-12
        T = _calc_T(z)
        P = np.abs(z) * 0.1                                                         # profile_pressure.py:13
        pH = _calc_pH(z)
        kw = dict(kind='quadratic', bounds_error=False, fill_value='extrapolate')
        self.c_samples = _medwin(z, S, T)                                           # physics_environment.py:70-73
        self.calc_c = interpolate.interp1d(z, self.c_samples, **kw)
        self.calc_dz_c = interpolate.interp1d(z, np.gradient(self.c_samples) / res_z, **kw)
        rho = _rho(z, S, T, P)                                                      # physics_environment.py:76-86
        self.calc_absorb = interpolate.interp1d(z, _absorb_coefs(z, T, S, pH), axis=1, **kw)
        self.calc_Z = interpolate.interp1d(z, rho * self.c_samples, **kw)
        floor_s, ceil_s = floor(self.x), ceil(self.x)                               # environment.py:46-53
        self.dx_floor = interpolate.interp1d(self.x, np.gradient(floor_s) / res_x, kind='quadratic', bounds_error=True)
        self.dx_ceil = interpolate.interp1d(self.x, np.gradient(ceil_s) / res_x, kind='quadratic', bounds_error=True)
        self.floor_avg, self.ceil_avg = np.mean(floor_s), np.mean(ceil_s)

        self._keep = []
        self.cenv = OrcEnv()
        self.cenv.range_min[:] = list(self.range_min)
        self.cenv.range_max[:] = list(self.range_max)
        self.cenv.c = self._quad(self.calc_c, 0)
        self.cenv.dz_c = self._quad(self.calc_dz_c, 0)
        self.cenv.dx_floor = self._quad(self.dx_floor, 2)
        self.cenv.dx_ceil = self._quad(self.dx_ceil, 2)
        self.cenv.impedance = self._quad(self.calc_Z, 0)
        b, co = _pp_from_interp1d(self.calc_absorb)                                 # (n, 3, 6)
        for k in range(6):
            self.cenv.absorb[k] = self._table(b, co[:, :, k], 0)
        self.cenv.floor = self._lin(floor)
        self.cenv.ceil = self._lin(ceil)

    @classmethod
    def from_tables(cls, tables, range_min, range_max):
        """Oracle environment over GIVEN tables ({name: object with .breaks, .coef, .oob}; names as in
        Rt2dEnvDesc, absorb0..absorb5).  Used to compare the HIP kernels with the C restatement bit for bit:
        both then read the very same numbers."""
        self = cls.__new__(cls)
        self.range_min = np.asarray(range_min, dtype=float)
        self.range_max = np.asarray(range_max, dtype=float)
        self._keep = []
        self.cenv = OrcEnv()
        self.cenv.range_min[:] = list(self.range_min)
        self.cenv.range_max[:] = list(self.range_max)
        for name in ('c', 'dz_c', 'floor', 'ceil', 'dx_floor', 'dx_ceil', 'impedance'):
            t = tables[name]
            setattr(self.cenv, name, self._table(t.breaks, t.coef, t.oob))
        for k in range(6):
            t = tables[f'absorb{k}']
            self.cenv.absorb[k] = self._table(t.breaks, t.coef, t.oob)
        return self

    def _table(self, breaks, coef, oob):
        breaks = np.ascontiguousarray(breaks, dtype=np.float64)
        coef = np.ascontiguousarray(coef, dtype=np.float64)
        self._keep += [breaks, coef]
        t = OrcTable()
        t.order, t.n_pieces, t.oob = coef.shape[1], coef.shape[0], oob
        t.breaks, t.coef = _dp(breaks), _dp(coef)
        return t

    def _quad(self, f, oob):
        b, co = _pp_from_interp1d(f)
        return self._table(b, co, oob)

    def _lin(self, lf):
        return self._table(lf.xs, np.stack((lf.slope, lf.zs[:-1]), axis=1), 1)


def make_params(dx_max=10., dz_max=0.1, n_steps_max=100000, n_rebounds_max=-1, backprop=True, target=None,
                rebound_cap=None):
    p = OrcParams()
    p.dx_max, p.dz_max = dx_max, dz_max
    p.n_steps_max, p.n_rebounds_max = n_steps_max, n_rebounds_max
    p.backprop = 1 if backprop else 0
    p.has_target = 0 if target is None else 1
    if target is not None:
        p.target[:] = [float(target[0]), float(target[1])]
    p.rebound_cap = rebound_cap if rebound_cap is not None else (n_rebounds_max + 1 if n_rebounds_max > -1 else 64)
    return p


class OracleRay:
    """Result of one oracle ray; attribute names follow Ray2D (ray.py)."""
    pass


def cast_ray(env, source, angle, params, init=None, cap=None, post=True):
    """Propagate one ray.  init=(slope, mult) injects host-computed start values."""
    L = lib()
    src = np.asarray(source, dtype=np.float64)
    cap = int(cap if cap is not None else params.n_steps_max + 2)
    X = np.empty(cap)
    Z = np.empty(cap)
    rc = params.rebound_cap
    n_pts, stop, n_reb = C.c_int32(), C.c_int32(), C.c_int32()
    margin = C.c_double()
    rs = np.zeros(rc, dtype=np.int32)
    rsurf = np.zeros(rc, dtype=np.uint8)
    rv, ra, rpx, rpz = np.zeros(rc), np.zeros(rc), np.zeros(rc), np.zeros(rc)
    use_init = 0 if init is None else 1
    s0, m0 = (0.0, 0.0) if init is None else (float(init[0]), float(init[1]))
    status = L.orc_propagate(C.byref(env.cenv), _dp(src), float(angle), C.byref(params), use_init, s0, m0,
                             _dp(X), _dp(Z), cap, C.byref(n_pts), C.byref(stop), C.byref(n_reb),
                             _ip(rs), rsurf.ctypes.data_as(C.POINTER(C.c_uint8)), _dp(rv), _dp(ra), _dp(rpx), _dp(rpz),
                             C.byref(margin))
    r = OracleRay()
    r.margin = margin.value
    r.angle, r.status = angle, status
    r.steps = n_pts.value
    r.stop_code = stop.value
    r.stop_reason = STOP_NAMES[stop.value]
    r.n_rebounds = n_reb.value
    nl = min(r.n_rebounds, rc)
    r.reb_step, r.reb_surface = rs[:nl].copy(), rsurf[:nl].copy()
    r.reb_velocity, r.reb_angle, r.reb_px, r.reb_pz = rv[:nl].copy(), ra[:nl].copy(), rpx[:nl].copy(), rpz[:nl].copy()
    n = r.steps
    r.XZ = np.stack((X[:n], Z[:n]), axis=1)
    if post:
        r.dL_full, r.L, r.C, r.dT_full, r.T = np.empty(n), np.empty(n), np.empty(n), np.empty(n), np.empty(n)
        rng = np.empty(4)
        st, tt, dd = C.c_int32(-1), C.c_double(np.nan), C.c_double(np.nan)
        tgt = None
        if params.has_target:
            tgt = np.array([params.target[0], params.target[1]])
        Xc, Zc = np.ascontiguousarray(r.XZ[:, 0]), np.ascontiguousarray(r.XZ[:, 1])
        L.orc_post(C.byref(env.cenv), _dp(Xc), _dp(Zc), n, _dp(tgt) if tgt is not None else None,
                   _dp(r.dL_full), _dp(r.L), _dp(r.C), _dp(r.dT_full), _dp(r.T), _dp(rng),
                   C.byref(st), C.byref(tt), C.byref(dd))
        r.dL, r.dT = r.dL_full[1:], r.dT_full[1:]
        r.range_min, r.range_max = rng[:2].copy(), rng[2:].copy()
        if tgt is not None:
            r.step_target, r.T_target, r.dist_to_target = st.value, tt.value, dd.value
        else:
            r.step_target = r.T_target = r.dist_to_target = None
    return r


def ray_from_points(XZ, reb_step=(), reb_surface=(), reb_velocity=(), reb_angle=(), reb_pz=(), status=0):
    """Oracle ray over GIVEN points and rebound log (e.g. the GPU's own), to check later stages in isolation."""
    r = OracleRay()
    r.XZ = np.ascontiguousarray(XZ, dtype=np.float64)
    r.steps = int(r.XZ.shape[0])
    r.reb_step = np.ascontiguousarray(reb_step, dtype=np.int32)
    r.reb_surface = np.ascontiguousarray(reb_surface, dtype=np.uint8)
    r.reb_velocity = np.ascontiguousarray(reb_velocity, dtype=np.float64)
    r.reb_angle = np.ascontiguousarray(reb_angle, dtype=np.float64)
    r.reb_pz = np.ascontiguousarray(reb_pz, dtype=np.float64)
    r.n_rebounds = len(r.reb_step)
    r.status = status
    return r


def post_ray(env, r, target=None):
    """Post-pass of one oracle ray in place (ray.py:263-287); returns dist_to_target (NaN without target)."""
    L = lib()
    n = r.steps
    r.dL_full, r.L, r.C, r.dT_full, r.T = (np.empty(n) for _ in range(5))
    rng = np.empty(4)
    st, tt, dd = C.c_int32(-1), C.c_double(np.nan), C.c_double(np.nan)
    X, Z = np.ascontiguousarray(r.XZ[:, 0]), np.ascontiguousarray(r.XZ[:, 1])
    tgt = None if target is None else np.ascontiguousarray(target, dtype=np.float64)
    L.orc_post(C.byref(env.cenv), _dp(X), _dp(Z), n, _dp(tgt) if tgt is not None else None,
               _dp(r.dL_full), _dp(r.L), _dp(r.C), _dp(r.dT_full), _dp(r.T), _dp(rng),
               C.byref(st), C.byref(tt), C.byref(dd))
    r.dL, r.dT = r.dL_full[1:], r.dT_full[1:]
    r.range_min, r.range_max = rng[:2].copy(), rng[2:].copy()
    r.step_target, r.T_target, r.dist_to_target = st.value, tt.value, dd.value
    return dd.value


def gain(env, ray, freq):
    """populate_NEW for one frequency -> G (steps,)."""
    L = lib()
    n = ray.steps
    G = np.empty(n)
    Zc = np.ascontiguousarray(ray.XZ[:, 1])
    L.orc_gain(C.byref(env.cenv), _dp(Zc), _dp(ray.dL_full), n, float(freq), len(ray.reb_step), _ip(ray.reb_step),
               ray.reb_surface.ctypes.data_as(C.POINTER(C.c_uint8)), _dp(ray.reb_velocity), _dp(ray.reb_angle),
               _dp(ray.reb_pz), ZP1, ZP2, _dp(G))
    return G


def apply_power_cutoff(env, ray, power_cutoff, freq, target=None):
    """Opt-in power termination (SURVEY trap 2; not in the reference).  The rule is stated on the UN-truncated ray:
    keep points 0..j where j is the first index with 10**(G[freq][j]/10) < power_cutoff (G from populate_NEW,
    ray.py:339-369); stop reason 'power-cutoff' -- except when j is the final point of a ray that ended on
    max-rebounds, which stays as it is.  Rebounds after point j are dropped, the post-pass is redone, a ray that
    failed (status -1) AFTER point j becomes a good truncated ray.  Returns (ray, j or None, dB distance of the
    decision from the threshold: min |G - G_cut| over the examined points)."""
    if not hasattr(ray, 'dL_full'):
        post_ray(env, ray, target)
    G = gain(env, ray, freq)
    with np.errstate(over='ignore'):
        below = np.nonzero(np.power(10., G / 10) < power_cutoff)[0]
    g_cut = 10 * np.log10(power_cutoff)
    if below.size == 0:
        return ray, None, float(np.min(np.abs(G - g_cut)))
    j = int(below[0])
    dist = float(np.min(np.abs(G[:j + 1] - g_cut)))
    if ray.stop_code == 5 and ray.status == 0 and j == ray.steps - 1:
        return ray, j, dist
    keep = ray.reb_step <= j
    out = ray_from_points(ray.XZ[:j + 1], ray.reb_step[keep], ray.reb_surface[keep], ray.reb_velocity[keep],
                          ray.reb_angle[keep], ray.reb_pz[keep])
    out.reb_px = ray.reb_px[keep] if hasattr(ray, 'reb_px') else None
    out.angle = getattr(ray, 'angle', None)
    out.margin = getattr(ray, 'margin', np.inf)
    out.stop_code, out.stop_reason = 8, 'power-cutoff'
    post_ray(env, out, target)
    return out, j, dist


def plot_range(rays):
    """RayPack2D.add range fold (ray.py:441-472): starts from zeros, so the origin is always included."""
    mn, mx = np.zeros(2), np.zeros(2)
    for r in rays:
        mn = np.minimum(mn, r.range_min)
        mx = np.maximum(mx, r.range_max)
    return mn, mx


def heatmap(env, rays, res=(50, 25), reduction_power=.5, cutoff=.02, freq=1):
    """Restated Simulation2D.heatmap (simulation.py:108-138). Returns (heatmap_full, heatmap_plot_T, dropped)."""
    L = lib()
    mn, mx = plot_range(rays)
    res = np.asarray(res, dtype=np.int32)
    shape = (np.ceil((mx - mn) / res).astype(int) + 1).astype(np.int32)           # simulation.py:117
    H = np.zeros(int(shape[0]) * int(shape[1]))
    dropped = 0
    for r in rays:
        G = gain(env, r, freq)
        Xc, Zc = np.ascontiguousarray(r.XZ[:, 0]), np.ascontiguousarray(r.XZ[:, 1])
        dropped += L.orc_heatmap_add(_dp(Xc), _dp(Zc), _dp(G), r.steps, _ip(res), _ip(shape), _dp(H))
    out = np.empty(int(shape[0]) * int(shape[1]))
    L.orc_heatmap_finalize(_dp(H), _ip(shape), float(cutoff), float(reduction_power), _dp(out))
    return H.reshape(shape[0], shape[1]), out.reshape(shape[1], shape[0]), dropped


def run_fan(env, source, angles, params, freq=1, res=(50, 25)):
    """Whole hot path for a fan in C (cast + post + gain + heat map), as timed by bench.py's cpu_baseline.
    Returns (ray_steps, heatmap_full (nx, nz), n_errors)."""
    L = lib()
    src = np.asarray(source, dtype=np.float64)
    ang = np.ascontiguousarray(angles, dtype=np.float64)
    res = np.asarray(res, dtype=np.int32)
    rng = np.zeros(4)
    nerr = C.c_int64(0)
    L.orc_run_fan(C.byref(env.cenv), _dp(src), _dp(ang), len(ang), C.byref(params), float(freq), ZP1, ZP2,
                  _ip(res), None, None, _dp(rng), C.byref(nerr))
    shape = (np.ceil((rng[2:] - rng[:2]) / res).astype(int) + 1).astype(np.int32)
    H = np.zeros(int(shape[0]) * int(shape[1]))
    steps = L.orc_run_fan(C.byref(env.cenv), _dp(src), _dp(ang), len(ang), C.byref(params), float(freq), ZP1, ZP2,
                          _ip(res), _ip(shape), _dp(H), _dp(rng), C.byref(nerr))
    return int(steps), H.reshape(int(shape[0]), int(shape[1])), int(nerr.value)
