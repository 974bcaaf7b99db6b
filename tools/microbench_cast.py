"""Micro-benchmark of the stepping kernel variants (development aid; run on the GPU box)."""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rt2d_loader  # noqa: E402
import bench  # noqa: E402

rt = rt2d_loader.load()
from ray_tracer_amd import _lib  # noqa: E402

dev = torch.device('cuda', 0)
env = bench.make_env(rt)
L = _lib.lib()
handle = env.device_handle(0)
stream = torch.cuda.current_stream().cuda_stream


def run(n, lean, n_steps_max=100000, reps=10):
    angles = torch.from_numpy(-(np.arange(n) + 1) * (0.999 / n)).to(dev)
    n_pts = torch.empty(n, dtype=torch.int32, device=dev)
    rng = torch.empty((4, n), dtype=torch.float64, device=dev)
    out = _lib.Rt2dRayOut()
    out.n_pts = n_pts.data_ptr()
    if not lean:
        out.range = rng.data_ptr()
    p = _lib.Rt2dParams()
    p.dx_max, p.dz_max, p.n_steps_max, p.n_rebounds_max, p.backprop = 50., 25., n_steps_max, 20, 1
    src = (C.c_double * 2)(0., 0.)
    for _ in range(2):
        _lib.check(L.rt2d_cast_count(handle, angles.data_ptr(), n, src, C.byref(p), None, None, C.byref(out), stream))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        _lib.check(L.rt2d_cast_count(handle, angles.data_ptr(), n, src, C.byref(p), None, None, C.byref(out), stream))
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    steps = int((n_pts - 1).sum().item())
    return ms, steps


for n in ((999,) if os.environ.get('RT2D_MB_SHORT') else (999, 16384, 65536, 262144, 1 << 20)):
    for lean in (False, True):
        ms, steps = run(n, lean, reps=10 if n < 100000 else 3)
        print(f'n={n:8d} lean={lean!s:5} {ms:8.3f} ms  {steps / ms / 1e6:8.2f} G steps/s  max-steps/ray={steps / n:.0f}', flush=True)
ms, steps = run(999, True, n_steps_max=1)
print(f'fixed overhead (1 step): {ms:.3f} ms')
ms, steps = run(999, True, n_steps_max=200)
print(f'200 steps: {ms:.3f} ms -> {1e3 * ms / 200:.2f} us/step incl. overhead')
ms2, _ = run(999, True, n_steps_max=400)
print(f'400 steps: {ms2:.3f} ms -> marginal {1e3 * (ms2 - ms) / 200:.2f} us/step')


def run_write(n, post, log, reps=10, room=2048):
    """Storing cast into a uniform provisional layout: which of POST / rebound log costs what per step."""
    angles = torch.from_numpy(-(np.arange(n) + 1) * (0.999 / n)).to(dev)
    f64, i32 = dict(dtype=torch.float64, device=dev), dict(dtype=torch.int32, device=dev)
    n_pts = torch.empty(n, **i32)
    nreb = torch.empty(n, **i32)
    rng = torch.empty((4, n), **f64)
    L_end, T_end, margin = (torch.empty(n, **f64) for _ in range(3))
    out = _lib.Rt2dRayOut()
    out.n_pts, out.n_rebounds = n_pts.data_ptr(), nreb.data_ptr()
    if post:
        out.range, out.L_end, out.T_end, out.margin = rng.data_ptr(), L_end.data_ptr(), T_end.data_ptr(), margin.data_ptr()
    cap = 21
    if log:
        rs = torch.empty((cap, n), **i32); rsf = torch.empty((cap, n), dtype=torch.uint8, device=dev)
        rv, ra, rx, rz = (torch.empty((cap, n), **f64) for _ in range(4))
        out.reb_step, out.reb_surface, out.reb_velocity = rs.data_ptr(), rsf.data_ptr(), rv.data_ptr()
        out.reb_angle, out.reb_pos_x, out.reb_pos_z = ra.data_ptr(), rx.data_ptr(), rz.data_ptr()
    p = _lib.Rt2dParams()
    p.dx_max, p.dz_max, p.n_steps_max, p.n_rebounds_max, p.backprop, p.rebound_cap = 50., 25., 100000, 20, 1, cap
    src = (C.c_double * 2)(0., 0.)
    n_tiles = (n + 63) // 64
    off0 = torch.arange(n_tiles + 1, dtype=torch.int64, device=dev) * (room * 64)
    X0, Z0 = torch.empty(n_tiles * room * 64, **f64), torch.empty(n_tiles * room * 64, **f64)

    def go():
        _lib.check(L.rt2d_cast_write(handle, angles.data_ptr(), n, src, C.byref(p), None, None, C.byref(out),
                                     off0.data_ptr(), X0.data_ptr(), Z0.data_ptr(), None, stream))
    go(); go()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        go()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps, int(n_pts.max().item())


for post in (False, True):
    for log in (False, True):
        ms, mx = run_write(999, post, log)
        print(f'write n=999 post={post!s:5} log={log!s:5}: {ms:.3f} ms -> {1e3 * ms / mx:.2f} us/step of the longest ray ({mx} pts)', flush=True)
