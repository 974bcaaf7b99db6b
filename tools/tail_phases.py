"""Where a workgroup of rt2d_small_tail_kernel spends its time (library built with -DRT2D_TAIL_LOG, RT2D_SO pointing at it):
100 MHz timestamps at its phase boundaries, per ray.  python tools/tail_phases.py"""
import os, sys, ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, rt2d_loader, bench
rt = rt2d_loader.load()
from ray_tracer_amd import _lib
L = _lib.lib()
L.rt2d_debug_wave_log.argtypes = [C.c_void_p, C.c_int64]
env = bench.make_env(rt)
angles = np.array([i / 1000 for i in range(-1, -1000, -1)])
n_blocks = 1024
buf = torch.zeros(n_blocks * 8, dtype=torch.int64, device='cuda')
shape = (1503, 62)
for rep in range(3):
    fan = rt.RayFan(env, np.array([0., 0.]), angles, **bench.PROP)
    H = torch.zeros(shape[0] * shape[1], dtype=torch.float64, device='cuda')
    torch.cuda.synchronize()
    buf.zero_()
    L.rt2d_debug_wave_log(buf.data_ptr(), n_blocks * 8)
    fan.heatmap_small(H, bench.RES, shape, freq=1)
    torch.cuda.synchronize()
    L.rt2d_debug_wave_log(None, 0)
t = buf.cpu().numpy().reshape(n_blocks, 8).astype(np.float64)
live = t[:, 6] > 0
if not live.any():
    print('no timestamps: this library was not built with -DRT2D_TAIL_LOG'); sys.exit(0)
t = t[live]
t0 = t[:, 0].min()
names = ['loads+hash fill+losses issued', 'cells+dG', 'scan+sync', 'cand+sync', 'gain', 'pass A', 'pass B']
print(f'{live.sum()} workgroups; kernel span {(t[:, 6].max() - t0) / 100:.1f} us; workgroup start spread {(t[:, 0].max() - t0) / 100:.1f} us')
d = np.diff(t[:, :7], axis=1) / 100.0
for k in range(6):
    print(f'  phase {k} -> {k + 1} ({names[k + 1] if k + 1 < len(names) else ""}): mean {d[:, k].mean():6.2f} us, max {d[:, k].max():6.2f} us')
print(f'  whole workgroup: mean {(t[:, 6] - t[:, 0]).mean() / 100:.2f} us, max {(t[:, 6] - t[:, 0]).max() / 100:.2f} us')
