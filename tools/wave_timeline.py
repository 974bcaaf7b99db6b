"""Development aid: when do the waves of the 1M-ray counting cast start and end, and on which CU?  (rt2d_debug_wave_log)"""
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
env = bench.make_env(rt)
L = _lib.lib()
handle = env.device_handle(0)
stream = torch.cuda.current_stream().cuda_stream
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
angles = torch.from_numpy(-(np.arange(n) + 1) * (0.999 / n)).to(dev)
n_pts = torch.empty(n, dtype=torch.int32, device=dev)
rng = torch.empty((4, n), dtype=torch.float64, device=dev)
out = _lib.Rt2dRayOut()
out.n_pts, out.range = n_pts.data_ptr(), rng.data_ptr()
p = _lib.Rt2dParams()
p.dx_max, p.dz_max, p.n_steps_max, p.n_rebounds_max, p.backprop, p.rebound_cap = 50., 25., 100000, 20, 1, 0
src = (C.c_double * 2)(0., 0.)
n_waves = (n + 63) // 64
log = torch.zeros((n_waves, 4), dtype=torch.int64, device=dev)
L.rt2d_debug_wave_log.argtypes = [C.c_void_p, C.c_int64]
L.rt2d_debug_wave_log.restype = None
_lib.check(L.rt2d_cast_count(handle, angles.data_ptr(), n, src, C.byref(p), None, None, C.byref(out), stream))
torch.cuda.synchronize()
L.rt2d_debug_wave_log(log.data_ptr(), n_waves)
_lib.check(L.rt2d_cast_count(handle, angles.data_ptr(), n, src, C.byref(p), None, None, C.byref(out), stream))
torch.cuda.synchronize()
L.rt2d_debug_wave_log(None, 0)
lg = log.cpu().numpy()
t0, t1 = lg[:, 0], lg[:, 1]
base = t0.min()
t0 = (t0 - base) / 100.0      # us
t1 = (t1 - base) / 100.0
dur = t1 - t0
steps = n_pts.cpu().numpy().reshape(-1, 64).max(axis=1) - 1
print(f'waves {n_waves}, kernel span {t1.max():.0f} us; wave duration us: min {dur.min():.0f} median {np.median(dur):.0f} max {dur.max():.0f}')
print(f'us per step of a wave: median {np.median(dur / steps):.3f}, p10 {np.percentile(dur / steps, 10):.3f}, p90 {np.percentile(dur / steps, 90):.3f}')
# resident waves over time
edges = np.linspace(0, t1.max(), 41)
for a, b in zip(edges[:-1], edges[1:]):
    mid = (a + b) / 2
    live = int(((t0 <= mid) & (t1 > mid)).sum())
    print(f'  t={mid:7.0f} us: {live:5d} waves live ({live / 1024:.2f} per SIMD)')
hw = lg[:, 2]
cu = ((hw >> 8) & 0xf) | (((hw >> 13) & 0x7) << 4) | (((hw >> 12) & 0x1) << 7) | ((lg[:, 3] & 0xf) << 8)
print('distinct (xcc, se, sh, cu) ids:', len(np.unique(cu)))
wg = np.arange(n_waves) // 16
for g in (0, 1, 300, 700, 1000):
    sel = wg == g
    print(f'  workgroup {g}: waves start {t0[sel].min():.0f}..{t0[sel].max():.0f}, end {t1[sel].min():.0f}..{t1[sel].max():.0f} us, steps {steps[sel].min()}..{steps[sel].max()}')
