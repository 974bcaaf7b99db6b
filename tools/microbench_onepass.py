"""Times rt2d_gain_heatmap alone (events) on the 1M-ray C2 fan."""
import os, sys, time, ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, rt2d_loader, bench
rt = rt2d_loader.load()
from ray_tracer_amd import _lib
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
env = bench.make_env(rt)
dev = torch.device('cuda', 0)
angles = torch.from_numpy(-(np.arange(n) + 1) * (0.999 / n)).to(dev)
shape = (1503, 62)
H = torch.zeros(shape[0] * shape[1], dtype=torch.float64, device=dev)
fan = rt.RayFan(env, np.array([0., 0.]), angles, store=True, **bench.PROP)
fan.validate()
m = fan.fused_mask(bench.RES, shape)
L = _lib.lib()
res_c = (C.c_int32 * 2)(*bench.RES); shape_c = (C.c_int32 * 2)(*shape)
viol = torch.zeros(1, dtype=torch.int64, device=dev)
loss_ws = torch.empty(fan.rebound_cap * n, dtype=torch.float64, device=dev)
stream = torch.cuda.current_stream().cuda_stream
def run():
    _lib.check(L.rt2d_gain_heatmap(env.device_handle(0), fan.X.data_ptr(), fan.Z.data_ptr(), fan.tile_off.data_ptr(), fan.n_pts.data_ptr(),
                                   C.byref(fan._out), fan.rebound_cap, n, 1.0, m.data_ptr(), res_c, shape_c, H.data_ptr(), None, None, None,
                                   viol.data_ptr(), loss_ws.data_ptr(), stream))
run(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(3): run()
e1.record(); torch.cuda.synchronize()
print(f"dbg={os.environ.get('RT2D_FUSE_DBG','0')} block={os.environ.get('RT2D_GH_BLOCK','auto')} n={n}: {e0.elapsed_time(e1)/3:.2f} ms (incl. 0.25 ms rebound-loss kernel), violations {int(viol.item())}")
# one clean map of this form (RT2D_GH_LEAN = 1: rt2d_onepass_kernel, 0: rt2d_gain_heatmap_kernel<true, true>) for comparison
H.zero_(); run(); torch.cuda.synchronize()
tag = os.environ.get('RT2D_GH_LEAN', '1')
os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
np.save(os.path.join(ROOT, 'gpurun_out', f'onepass_H_lean{tag}.npy'), H.cpu().numpy())
other = os.path.join(ROOT, 'gpurun_out', f'onepass_H_lean{"0" if tag != "0" else "1"}.npy')
if os.path.exists(other):
    Ho, Hn = np.load(other), H.cpu().numpy()
    both = (Ho != 0) & np.isfinite(Ho)
    print(f'maps lean vs exact: same occupancy {np.array_equal(Ho == 0, Hn == 0)}, worst relative difference '
          f'{(np.abs(Hn - Ho)[both] / np.abs(Ho[both])).max():.3e}, sum {Hn.sum():.9f} vs {Ho.sum():.9f}')
