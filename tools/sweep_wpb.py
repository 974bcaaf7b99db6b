"""Development aid: the 1M-ray counting and storing casts for several workgroup sizes (RT2D_CAST_WPB waves per workgroup)."""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rt2d_loader  # noqa: E402
import bench  # noqa: E402

rt = rt2d_loader.load()
from ray_tracer_amd import _lib  # noqa: E402

dev = torch.device('cuda', 0)
if os.environ.get('SWEEP_FLOOR_N'):
    xs = np.linspace(0., 75000., int(os.environ['SWEEP_FLOOR_N']))
    env = rt.Environment2D((xs, bench.c2_floor(xs)), (np.array([0., 75000.]), np.array([0., 0.])), np.array([0., -1501.]), np.array([75000., 1.]))
else:
    env = bench.make_env(rt)
L = _lib.lib()
handle = env.device_handle(0)
stream = torch.cuda.current_stream().cuda_stream
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
angles = torch.from_numpy(-(np.arange(n) + 1) * (0.999 / n)).to(dev)
f64, i32 = dict(dtype=torch.float64, device=dev), dict(dtype=torch.int32, device=dev)
n_pts = torch.empty(n, **i32)
rng = torch.empty((4, n), **f64)
out = _lib.Rt2dRayOut()
out.n_pts, out.range = n_pts.data_ptr(), rng.data_ptr()
p = _lib.Rt2dParams()
p.dx_max, p.dz_max, p.n_steps_max, p.n_rebounds_max, p.backprop, p.rebound_cap = 50., 25., 100000, 20, 1, 0
src = (C.c_double * 2)(0., 0.)
room = 2048
n_tiles = (n + 63) // 64
off0 = torch.arange(n_tiles + 1, dtype=torch.int64, device=dev) * (room * 64)
X0, Z0 = torch.empty(n_tiles * room * 64, **f64), torch.empty(n_tiles * room * 64, **f64)


def timed(fn, reps=3):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for wpb in [int(v) for v in os.environ.get('SWEEP_WPB', '16,8,5,4,3,2,1').split(',')]:
    os.environ['RT2D_CAST_WPB'] = str(wpb)
    a = timed(lambda: _lib.check(L.rt2d_cast_count(handle, angles.data_ptr(), n, src, C.byref(p), None, None, C.byref(out), stream)))
    b = timed(lambda: _lib.check(L.rt2d_cast_write(handle, angles.data_ptr(), n, src, C.byref(p), None, None, C.byref(out),
                                                   off0.data_ptr(), X0.data_ptr(), Z0.data_ptr(), None, stream)))
    print(f'wpb={wpb:2d}: count {a:.3f} ms   store {b:.3f} ms', flush=True)
