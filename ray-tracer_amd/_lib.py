"""ctypes binding of librt2d_hip.so (C ABI: include/rt2d.h).

The library is the ONLY compute path of this package: if it is missing, or there is no GPU, the
calls raise -- there is no CPU fallback (the CPU restatement lives in oracle/ and is test-only).
"""

import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.environ.get('RT2D_SO') or os.path.join(_HERE, 'librt2d_hip.so')      # RT2D_SO: an experimental build (development)
SOURCES = [os.path.join(_HERE, 'csrc', 'rt2d_kernels.hip')]
HEADERS = [os.path.join(_HERE, 'csrc', 'rt2d_device.h'), os.path.join(os.path.dirname(_HERE), 'include', 'rt2d.h')]

STOP_REASONS = [None, 'exit-xmin', 'exit-xmax', 'exit-zmin', 'exit-zmax', 'max-rebounds', 'backprop', 'max-iter',
                'power-cutoff']
RAY_OK, RAY_ERR_INTERP, RAY_ERR_REBOUND_LOG = 0, -1, -2

EXPORTS = ['rt2d_last_error', 'rt2d_version', 'rt2d_env_create', 'rt2d_env_destroy', 'rt2d_env_surface_speed',
           'rt2d_cast_count', 'rt2d_traj_layout', 'rt2d_fan_summary', 'rt2d_cast_write', 'rt2d_traj_compact', 'rt2d_post', 'rt2d_gain', 'rt2d_dist', 'rt2d_target_products',
           'rt2d_heatmap_accumulate', 'rt2d_heatmap_accumulate_noalloc', 'rt2d_heatmap_accumulate_multi', 'rt2d_heatmap_finalize_multi', 'rt2d_heatmap_finalize', 'rt2d_cast_heatmap', 'rt2d_gain_heatmap', 'rt2d_gain_heatmap_rest', 'rt2d_classify_rays', 'rt2d_layout_summary', 'rt2d_rebound_loss', 'rt2d_alpha_table_error', 'rt2d_small_fan_heatmap', 'rt2d_heatmap_hints', 'rt2d_heatmap_accumulate_hinted']


# (-falign-loops=64: the stepping loops of lone waves are sensitive to where their heads fall in the instruction cache --
# 999-ray step 0.689 / 0.711 / 0.685 / 0.692 ms with none / 32 / 64 / 128)
BUILD_FLAGS = ['--offload-arch=gfx950', '-O3', '-ffp-contract=off', '-falign-loops=64', '-fPIC', '-shared', '-std=c++17']


def source_hash():
    """sha256 (16 hex digits) over the kernel sources, the C header and the compiler flags: names the build a profile
    measured."""
    import glob
    import hashlib
    h = hashlib.sha256()
    h.update(' '.join(BUILD_FLAGS).encode())
    files = sorted(glob.glob(os.path.join(_HERE, 'csrc', '*'))) + [os.path.join(os.path.dirname(_HERE), 'include', 'rt2d.h')]
    for path in files:
        if os.path.isfile(path) and not path.endswith(('.o', '.so')):
            h.update(os.path.basename(path).encode())
            with open(path, 'rb') as f:
                h.update(f.read())
    return h.hexdigest()[:16]


def build_info():
    """{source_sha256, git_head, git_dirty}: the hash is computed from the tree; the git fields come from the file written
    next to the library when it was built here (there is no .git on the GPU box)."""
    import json
    info = {'source_sha256': source_hash(), 'git_head': None, 'git_dirty': None}
    path = os.path.join(_HERE, 'BUILD_INFO.json')
    if os.path.exists(path):
        try:
            with open(path) as f:
                old = json.load(f)
            if old.get('source_sha256') == info['source_sha256']:
                info.update(git_head=old.get('git_head'), git_dirty=old.get('git_dirty'))
        except (OSError, ValueError):
            pass
    return info


def _write_build_info():
    import json
    import subprocess
    info = {'source_sha256': source_hash(), 'git_head': None, 'git_dirty': None}
    root = os.path.dirname(_HERE)
    try:
        info['git_head'] = subprocess.check_output(['git', '-C', root, 'rev-parse', 'HEAD'], text=True, stderr=subprocess.DEVNULL).strip()
        info['git_dirty'] = bool(subprocess.check_output(['git', '-C', root, 'status', '--porcelain', '--', 'ray-tracer_amd/csrc', 'include'],
                                                         text=True, stderr=subprocess.DEVNULL).strip())
    except (OSError, subprocess.CalledProcessError):
        pass
    try:                                                    # (atomic: N ranks may import at once)
        path = os.path.join(_HERE, 'BUILD_INFO.json')
        tmp = f'{path}.{os.getpid()}.tmp'
        with open(tmp, 'w') as f:
            json.dump(info, f)
        os.replace(tmp, path)
    except OSError:
        pass


def _build_info_is_current():
    import json
    try:
        with open(os.path.join(_HERE, 'BUILD_INFO.json')) as f:
            return json.load(f).get('source_sha256') == source_hash()
    except (OSError, ValueError):
        return False


def build(force=False, verbose=False):
    """Compile the HIP kernels for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    deps = SOURCES + HEADERS
    if not force and os.path.exists(SO_PATH) and all(os.path.getmtime(SO_PATH) >= os.path.getmtime(d) for d in deps):
        if not _build_info_is_current():                   # (only when the stamp names other sources: no git calls, no
            _write_build_info()                            # rewrite on every import)
        return SO_PATH
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    cmd = [hipcc] + BUILD_FLAGS + ['-o', SO_PATH] + SOURCES
    if verbose:
        print(' '.join(cmd))
    subprocess.check_call(cmd)
    _write_build_info()
    return SO_PATH


class Rt2dTable(C.Structure):
    _fields_ = [('order', C.c_int32), ('n_pieces', C.c_int32), ('oob', C.c_int32), ('reserved', C.c_int32),
                ('breaks', C.c_void_p), ('coef', C.c_void_p)]


class Rt2dEnvDesc(C.Structure):
    _fields_ = [('range_min', C.c_double * 2), ('range_max', C.c_double * 2),
                ('c', Rt2dTable), ('dz_c', Rt2dTable), ('floor', Rt2dTable), ('ceil', Rt2dTable),
                ('dx_floor', Rt2dTable), ('dx_ceil', Rt2dTable), ('impedance', Rt2dTable),
                ('absorb', Rt2dTable * 6)]


class Rt2dParams(C.Structure):
    _fields_ = [('dx_max', C.c_double), ('dz_max', C.c_double),
                ('n_steps_max', C.c_int32), ('n_rebounds_max', C.c_int32),
                ('backprop', C.c_int32), ('has_target', C.c_int32),
                ('target', C.c_double * 2),
                ('rebound_cap', C.c_int32), ('launch_hint', C.c_int32),
                ('power_cutoff', C.c_double), ('cutoff_freq', C.c_double)]


RAY_OUT_FIELDS = ['n_pts', 'stop_code', 'n_rebounds', 'status', 'range', 'L_end', 'T_end', 'dist_to_target',
                  'step_target', 'T_target', 'reb_step', 'reb_surface', 'reb_velocity', 'reb_angle', 'reb_pos_x',
                  'reb_pos_z', 'margin', 'monotone', 'slope_max']


class Rt2dRayOut(C.Structure):
    _fields_ = [(name, C.c_void_p) for name in RAY_OUT_FIELDS]


class Rt2dError(RuntimeError):
    pass


_lib = None


def lib():
    """Load the shared library, failing loudly when it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(SO_PATH):
        raise Rt2dError(f'{SO_PATH} is missing: build it with `python -c "import __graft_entry__ as g; g.build()"` '
                        '(hipcc --offload-arch=gfx950). There is no CPU fallback for this path.')
    # torch ships its own copy of the HIP runtime; it has to be in the process BEFORE this library so that both
    # resolve to the same runtime (and device pointers mean the same thing on both sides)
    import torch  # noqa: F401
    L = C.CDLL(SO_PATH)
    vp, i64, i32, dbl = C.c_void_p, C.c_int64, C.c_int32, C.c_double
    L.rt2d_last_error.restype = C.c_char_p
    L.rt2d_last_error.argtypes = []
    L.rt2d_version.restype = i32
    L.rt2d_version.argtypes = []
    sig = {
        'rt2d_env_create': [C.POINTER(Rt2dEnvDesc), C.POINTER(vp)],
        'rt2d_env_destroy': [vp],
        'rt2d_env_surface_speed': [vp, C.POINTER(dbl)],
        'rt2d_cast_count': [vp, vp, i64, C.POINTER(dbl), C.POINTER(Rt2dParams), vp, vp, C.POINTER(Rt2dRayOut), vp],
        'rt2d_traj_layout': [vp, i64, vp, C.POINTER(i64), vp],
        'rt2d_fan_summary': [vp, vp, vp, i64, vp, vp, vp],
        'rt2d_cast_write': [vp, vp, i64, C.POINTER(dbl), C.POINTER(Rt2dParams), vp, vp, C.POINTER(Rt2dRayOut),
                            vp, vp, vp, vp, vp],
        'rt2d_layout_summary': [vp, vp, vp, i64, vp, vp, i32, vp, vp],
        'rt2d_traj_compact': [vp, vp, i64, vp, vp, vp, vp, vp, vp, vp],
        'rt2d_post': [vp, vp, vp, vp, vp, i64, vp, vp, vp, vp, vp, vp],
        'rt2d_gain': [vp, vp, vp, vp, vp, C.POINTER(Rt2dRayOut), i32, i64, vp, i32, i64, vp, vp, vp, vp],
        'rt2d_dist': [vp, vp, vp, vp, i64, C.POINTER(dbl), vp, vp],
        'rt2d_target_products': [vp, vp, vp, vp, vp, vp, vp, vp, i64, C.POINTER(dbl), vp, i32, vp, vp, vp, vp],
        'rt2d_heatmap_accumulate': [vp, vp, vp, vp, vp, i64, vp, C.POINTER(i32), C.POINTER(i32), vp, vp, vp, vp],
        'rt2d_heatmap_accumulate_noalloc': [vp, vp, vp, vp, vp, i64, vp, C.POINTER(i32), C.POINTER(i32), vp, vp, vp],
        'rt2d_heatmap_accumulate_multi': [vp, vp, vp, i32, vp, vp, i64, vp, C.POINTER(i32), C.POINTER(i32), vp, vp, vp, vp, vp],
        'rt2d_heatmap_finalize_multi': [vp, C.POINTER(i32), i32, dbl, dbl, vp, vp, vp],
        'rt2d_heatmap_finalize': [vp, C.POINTER(i32), dbl, dbl, vp, vp, vp],
        'rt2d_gain_heatmap': [vp, vp, vp, vp, vp, C.POINTER(Rt2dRayOut), i32, i64, dbl, vp, C.POINTER(i32), C.POINTER(i32),
                              vp, vp, vp, vp, vp, vp, vp],
        'rt2d_gain_heatmap_rest': [vp, vp, vp, vp, vp, C.POINTER(Rt2dRayOut), i32, i64, dbl, vp, C.POINTER(i32), C.POINTER(i32),
                                   vp, vp, i64, vp, i64, vp, vp, vp, vp, vp, vp],
        'rt2d_classify_rays': [vp, vp, vp, vp, vp, i64, C.POINTER(i32), C.POINTER(i32), i64, vp, vp, vp, vp, vp],
        'rt2d_heatmap_hints': [vp, vp, vp, i64, C.POINTER(i32), C.POINTER(i32), vp, vp],
        'rt2d_heatmap_accumulate_hinted': [vp, vp, vp, vp, vp, i64, vp, C.POINTER(i32), C.POINTER(i32), vp, vp, vp, i64, vp],
        'rt2d_cast_heatmap': [vp, vp, i64, C.POINTER(dbl), C.POINTER(Rt2dParams), vp, vp, vp, dbl, C.POINTER(i32),
                              C.POINTER(i32), dbl, vp, C.POINTER(Rt2dRayOut), vp, vp, vp, i32, vp],
        'rt2d_rebound_loss': [vp, C.POINTER(Rt2dRayOut), i32, i64, vp, i32, vp, vp],
        'rt2d_alpha_table_error': [vp, dbl, C.POINTER(dbl), C.POINTER(i32)],
        'rt2d_small_fan_heatmap': [vp, vp, vp, vp, vp, C.POINTER(Rt2dRayOut), i32, i64, i64, dbl, vp, C.POINTER(i32), C.POINTER(i32),
                                   vp, vp, dbl, dbl, vp, vp, vp, vp],
    }
    for name, argtypes in sig.items():
        fn = getattr(L, name)
        fn.restype = i32
        fn.argtypes = argtypes
    _lib = L
    return L


def check(rc):
    if rc != 0:
        raise Rt2dError(lib().rt2d_last_error().decode())
