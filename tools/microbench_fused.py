"""Times rt2d_cast_heatmap alone on the C2 environment (1M rays by default).  RT2D_FUSE_DBG / RT2D_FUSE_WAVES vary it."""
import ctypes as C
import os
import sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import rt2d_loader
import bench
rt = rt2d_loader.load()
from ray_tracer_amd import _lib
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
if os.environ.get('SWEEP_FLOOR_N'):
    xs = np.linspace(0., 75000., int(os.environ['SWEEP_FLOOR_N']))
    env = rt.Environment2D((xs, bench.c2_floor(xs)), (np.array([0., 75000.]), np.array([0., 0.])), np.array([0., -1501.]), np.array([75000., 1.]))
else:
    env = bench.make_env(rt)
dev = torch.device('cuda', 0)
angles = torch.from_numpy(-(np.arange(n) + 1) * (0.999 / n)).to(dev)
fan = rt.RayFan(env, np.array([0., 0.]), angles, store=False, **bench.PROP)
shape = (1503, 62)
mask = fan.fused_mask(bench.RES, shape)
H = torch.zeros(shape[0] * shape[1], dtype=torch.float64, device=dev)
viol = torch.zeros(1, dtype=torch.int64, device=dev)
src = (C.c_double * 2)(0., 0.)
p = fan._params(0)
L = _lib.lib()
use_loss = os.environ.get('RT2D_FUSE_LOSS', '1') != '0' and fan.rebound_cap > 0
loss = None
if use_loss:
    loss = torch.empty(fan.rebound_cap * n, dtype=torch.float64, device=dev)
    fr = torch.ones(1, dtype=torch.float64, device=dev)
    _lib.check(L.rt2d_rebound_loss(env.device_handle(0), C.byref(fan._out), fan.rebound_cap, n, fr.data_ptr(), 1, loss.data_ptr(),
                                   torch.cuda.current_stream().cuda_stream))
def run():
    _lib.check(L.rt2d_cast_heatmap(env.device_handle(0), angles.data_ptr(), n, src, C.byref(p), None, None, mask.data_ptr(), 1.0,
                                   (C.c_int32 * 2)(*bench.RES), (C.c_int32 * 2)(*shape), 1.0, H.data_ptr(), None, None,
                                   viol.data_ptr(), loss.data_ptr() if use_loss else None, fan.rebound_cap if use_loss else 0,
                                   torch.cuda.current_stream().cuda_stream))
run(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(3): run()
e1.record(); torch.cuda.synchronize()
steps = int((fan.n_pts[mask != 0] - 1).sum().item())
ms = e0.elapsed_time(e1) / 3
print(f"loss={use_loss} dbg={os.environ.get('RT2D_FUSE_DBG', '0')} waves={os.environ.get('RT2D_FUSE_WAVES', 'auto')} n={n}: {ms:.2f} ms, "
      f"{steps / ms / 1e6:.1f} G ray-steps/s, violations {int(viol.item())}, rays binned {int(mask.sum().item())}")
