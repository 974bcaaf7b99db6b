"""CPU tests of the host side: tables, environment mirror, C-ABI library surface, loud failure without a GPU."""

import os
import re
import subprocess
import sys

import numpy as np
import pytest
from scipy.interpolate import interp1d

import common

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_quadratic_table_matches_interp1d(rt):
    rng = np.random.default_rng(0)
    for n in (3, 4, 5, 17, 151, 501):
        x = np.sort(rng.uniform(-5000, 1, n)) if n != 501 else np.linspace(-5001, 1, 501)
        y = 1500 + 30 * np.sin(x / 300)
        f = interp1d(x, y, kind='quadratic', bounds_error=False, fill_value='extrapolate')
        t = rt.quadratic_interp_table(x, y)
        xs = rng.uniform(x[0] - 50, x[-1] + 50, 5000)
        assert np.max(np.abs(f(xs) - t(xs)) / np.abs(f(xs))) < 5e-15
        assert np.max(np.abs(t(x) - y)) < 1e-11


def test_own_tridiagonal_solver_agrees_with_scipy(rt):
    from ray_tracer_amd import pptable
    x = np.linspace(-1501, 1, 151)
    y = 1500 + 30 * np.sin(x / 300)
    b1, c1 = pptable.quadratic_spline_pieces(x, y)
    saved = pptable._make_interp_spline
    pptable._make_interp_spline = None
    try:
        b2, c2 = pptable.quadratic_spline_pieces(x, y)
    finally:
        pptable._make_interp_spline = saved
    assert np.array_equal(b1, b2)
    assert np.max(np.abs(c1[:, 2] - c2[:, 2])) < 1e-10 and np.max(np.abs(c1[:, 1] - c2[:, 1])) < 1e-11


def test_linear_table_semantics(rt):
    t = rt.linear_interp_table([0., 10., 30.], [-5., -15., -15.])
    assert t(0.) == -5. and t(10.) == -15. and t(5.) == -10.
    assert t(-3.) == -5. and t(99.) == -15.        # clamped ends
    assert np.array_equal(t(np.array([0., 5., 20.])), [-5., -10., -15.])


def test_environment_mirror_matches_oracle_tables(rt, orc):
    """Product tables (own power-basis conversion) against the oracle's (scipy BSpline derivatives)."""
    g = common.load('c2_fan999')
    env, oenv = common.product_env(rt, g), common.oracle_env(orc, g)
    assert env.penv.z.shape[0] == 151 and env.penv.x.shape[0] == 751      # SURVEY a-6 / a-7
    assert abs(float(env.penv.calc_c(0.)) - 1526.8679) < 1e-3             # SURVEY 8(a) default profile [probe]
    zs = np.random.default_rng(1).uniform(-1600, 20, 2000)
    assert np.max(np.abs(env.penv.calc_c(zs) - oenv.calc_c(zs)) / oenv.calc_c(zs)) < 1e-12
    assert np.max(np.abs(env.penv.calc_dz_c(zs) - oenv.calc_dz_c(zs))) < 1e-13
    assert np.max(np.abs(env.penv.calc_Z(zs) - oenv.calc_Z(zs)) / oenv.calc_Z(zs)) < 1e-12
    assert np.max(np.abs(env.penv.calc_dl_dG_coefs_interp(zs) - oenv.calc_absorb(zs))) < 1e-12
    xs = np.random.default_rng(2).uniform(0, 75000, 2000)
    assert np.max(np.abs(env.dx_floor(xs) - oenv.dx_floor(xs))) < 1e-15
    assert np.array_equal(env.floor(xs), oenv.floor(xs))
    assert env.floor_avg == oenv.floor_avg and env.ceil_avg == oenv.ceil_avg
    with pytest.raises(ValueError):
        env.dx_floor(75000.1)
    # absorption formula
    f = np.array([1., 1000., 20000.])
    a = env.penv.calc_dl_dG(f, -300.)
    b = np.array([orc.lib().orc_absorption(oenv.cenv, float(v), -300.) for v in f])
    assert np.max(np.abs(a - b) / b) < 1e-13


def test_callable_floor_is_tabulated(rt):
    env = rt.Environment2D(lambda x: -1000 + 0.001 * x, lambda x: 0 * x, np.array([0., -1000.]), np.array([100000., 0.]))
    assert isinstance(env.floor, rt.PPTable) and env.floor.order == 2
    assert env.floor.n_pieces == 4 * 1000
    assert abs(env.floor(50000.) + 950.) < 1e-9
    assert env.env_desc()[0].floor.n_pieces == 4000


def test_library_exports_every_declared_symbol(rt):
    so = rt._lib.SO_PATH
    assert os.path.exists(so), 'librt2d_hip.so missing: run __graft_entry__.build()'
    header = open(os.path.join(ROOT, 'include', 'rt2d.h')).read()
    declared = set(re.findall(r'\b(rt2d_[a-z_0-9]+)\s*\(', header))
    declared.discard('rt2d_env_s')
    syms = subprocess.check_output(['nm', '-D', '--defined-only', so], text=True)
    exported = set(re.findall(r' T (rt2d_[a-z_0-9]+)', syms))
    assert declared <= exported, declared - exported
    assert declared == set(rt._lib.EXPORTS)


def test_library_loads_without_gpu(rt):
    L = rt._lib.lib()
    assert L.rt2d_version() >= 100
    assert isinstance(L.rt2d_last_error(), bytes)


def test_no_cpu_fallback(rt):
    import torch
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    g = common.load('c1_flat250')
    env = common.product_env(rt, g)
    with pytest.raises(rt._lib.Rt2dError):
        rt.RayFan(env, g['fan_source'], g['fan_angles'][:4])
    with pytest.raises(rt._lib.Rt2dError):
        rt.Ray2D(env, g['fan_source'], -0.1).propagate()


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, 'ray-tracer_amd')
    for name in os.listdir(pkg):
        if name.endswith('.py'):
            src = open(os.path.join(pkg, name)).read()
            assert not re.search(r'^\s*(from|import)\s+oracle', src, flags=re.M), name


def test_raypack_bookkeeping_without_gpu(rt):
    """RayPack2D.add with hand-made rays: duplicate angles skipped, plot range folds from zero (ray.py:451-475)."""
    g = common.load('c1_flat250')
    env = common.product_env(rt, g)
    pack = rt.RayPack2D()
    rays = []
    for a, rmin, rmax, d in ((-0.1, [0., -500.], [900., -50.], 3.0), (-0.2, [0., -800.], [400., -50.], 1.0),
                             (-0.1, [0., -9999.], [99999., 0.], 0.5)):
        r = rt.Ray2D(env, g['fan_source'], a, target=np.array([100., -100.]))
        r.range_min, r.range_max, r.dist_to_target, r.stop_reason = np.array(rmin), np.array(rmax), d, 'exit-xmax'
        rays.append(r)
    pack.add(*rays)
    assert pack.n_rays == 2 and pack.n_angles == 2
    assert np.array_equal(pack.range_min_plot, [0., -800.]) and np.array_equal(pack.range_max_plot, [900., 0.])
    assert np.array_equal(pack.size_plot, [900., 800.])
    assert list(pack.dist_sorted) == [1.0, 3.0]
    assert pack.dist[1.0][0] is rays[1] and pack.angles[-0.1] is rays[0]
    assert len(pack.stop_reasons['exit-xmax']) == 2


def test_argument_checks_answer_before_any_gpu_work(rt):
    """Every entry point rejects null / negative arguments with a negative status and a message (include/rt2d.h:
    "every call returns int32 status, message via rt2d_last_error"); these checks run before the first HIP call, so
    they hold on a machine without a GPU."""
    import ctypes as C
    L = rt._lib.lib()
    calls = {
        'rt2d_env_create': (None, None),
        'rt2d_env_surface_speed': (None, None),
        'rt2d_traj_layout': (None, 4, None, None, None),
        'rt2d_dist': (None, None, None, None, 4, None, None, None),
        'rt2d_heatmap_finalize': (None, None, 0.02, 0.5, None, None, None),
    }
    for name, args in calls.items():
        rc = getattr(L, name)(*args)
        assert rc < 0, name
        msg = L.rt2d_last_error().decode()
        assert name.replace('rt2d_', '') in msg or 'null' in msg, (name, msg)
    # an environment whose tables are malformed never reaches the device either
    g = common.load('c1_flat250')
    env = common.product_env(rt, g)
    desc, keep = env.env_desc()
    desc.floor.n_pieces = 0
    handle = C.c_void_p()
    assert L.rt2d_env_create(C.byref(desc), C.byref(handle)) < 0
    assert b'floor' in L.rt2d_last_error() and not handle.value


def test_half_offset_floor_division():
    """The identity the heat-map kernels use for int(x) // res (csrc/rt2d_kernels.hip, rt2d_fdiv_pos): for integers
    |t| < 2^31 and 1 <= b < 2^20, floor((t + 0.5) * RN(1/b)) == t // b, in double arithmetic as the device does it."""
    rng = np.random.default_rng(0)
    for b in [1, 2, 3, 5, 7, 10, 25, 49, 50, 51, 100, 999, 1000, 4096, 65535, (1 << 20) - 1]:
        inv = np.float64(1.0) / np.float64(b)
        ks = rng.integers(-(2 ** 31) // b, (2 ** 31) // b, 50000)
        t = np.concatenate([ks * b, ks * b + 1, ks * b - 1, ks * b + b // 2, rng.integers(-2 ** 31 + 1, 2 ** 31 - 1, 100000),
                            np.array([0, 1, -1, 2 ** 31 - 1, -2 ** 31 + 1])])
        t = t[(t > -2 ** 31) & (t < 2 ** 31)]
        q = np.floor((t.astype(np.float64) + 0.5) * inv).astype(np.int64)
        assert np.array_equal(q, t // b), b


def test_ctypes_signatures_follow_the_header(rt):
    """Every prototype of include/rt2d.h has a ctypes signature of the same arity in ray-tracer_amd/_lib.py (the
    definitions themselves are checked against the header by the compiler: csrc includes it inside extern "C")."""
    header = open(os.path.join(ROOT, 'include', 'rt2d.h')).read()
    header = re.sub(r'/\*.*?\*/', '', header, flags=re.S)
    L = rt._lib.lib()
    seen = 0
    for m in re.finditer(r'\b(?:int32_t|const char\*)\s+(rt2d_\w+)\s*\(([^;]*?)\)\s*;', header, flags=re.S):
        name, args = m.group(1), m.group(2).strip()
        n = 0 if args in ('void', '') else len(args.split(','))
        at = getattr(L, name).argtypes
        assert at is not None and len(at) == n, (name, n, at)
        seen += 1
    assert seen == len(rt._lib.EXPORTS)


def test_bench_launches_one_process_per_gpu():
    """`python bench.py --gpus N` (no launcher around it) starts N ranks itself, with the torch.distributed environment
    set, before anything touches a GPU; a failing rank makes the whole run fail.  --dry-run: the ranks only print."""
    import json
    import subprocess
    bench = os.path.join(ROOT, 'bench.py')
    out = subprocess.run([sys.executable, bench, '--gpus', '4', '--dry-run'], capture_output=True, text=True, timeout=60)
    assert out.returncode == 0, out.stderr[-2000:]
    recs = sorted((json.loads(l) for l in out.stdout.splitlines() if l.startswith('{')), key=lambda r: r['rank'])
    assert [r['rank'] for r in recs] == [0, 1, 2, 3] and all(r['world'] == 4 and r['local_rank'] == r['rank'] for r in recs)
    assert len({r['port'] for r in recs}) == 1
    env = dict(os.environ, RT2D_BENCH_DRY_FAIL_RANK='2')
    bad = subprocess.run([sys.executable, bench, '--gpus', '4', '--dry-run'], capture_output=True, text=True, timeout=60, env=env)
    assert bad.returncode != 0
