"""Development aid (needs a library built with -DRT2D_WAVE_LOG, RT2D_SO=...): for the storing cast of the 999-ray headline
fan, how many steps each wave takes in its tight loop and in the general loop, and the time it spends in each."""
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

n = int(sys.argv[1]) if len(sys.argv) > 1 else 999
env = bench.make_env(rt)
L = _lib.lib()
L.rt2d_debug_wave_log.argtypes = [C.c_void_p, C.c_int64]
L.rt2d_debug_wave_log.restype = None
angles = -(np.arange(n) + 1) * (0.999 / n) if n != 999 else -(np.arange(999) + 1) / 1000.
prop = dict(bench.PROP)
kw = {}
if len(sys.argv) > 3:      # python tools/wave_steps.py N lo hi [n_steps_max n_rebounds_max]: a cone of steep rays as the refine loop casts
    angles = -np.linspace(float(sys.argv[2]), float(sys.argv[3]), n)
    if len(sys.argv) > 5:
        prop.update(n_steps_max=int(sys.argv[4]), n_rebounds_max=int(sys.argv[5]))
    if len(sys.argv) > 7:  # ... dx_max dz_max
        prop.update(dx_max=float(sys.argv[6]), dz_max=float(sys.argv[7]))
    kw = dict(target=np.array([30000., -300.]), store=False)
source = np.array([0., 0.])
fan = rt.RayFan(env, source, angles, **prop, **kw)
torch.cuda.synchronize()
cap = 1 << 16
log = torch.zeros((cap, 12), dtype=torch.int64, device='cuda')
L.rt2d_debug_wave_log(log.data_ptr(), cap)
fan = rt.RayFan(env, source, angles, **prop, **kw)
torch.cuda.synchronize()
L.rt2d_debug_wave_log(None, 0)
lg = log.cpu().numpy()
lg = lg[lg[:, 1] > 0]
dur = (lg[:, 1] - lg[:, 0]) / 100.
print(f'{len(lg)} waves, kernel span {(lg[:, 1].max() - lg[:, 0].min()) / 100.:.0f} us, longest wave {dur.max():.0f} us')
order = np.argsort(-dur)
print('wave  dur_us  tight_steps  general_iters  us_tight  us_general  ns/tight_step  ns/general_iter  general: some ray off the common step / some ray steep')
for w in list(order[:8]) + list(order[len(order) // 2: len(order) // 2 + 3]):
    nt, ng, tt, tg = lg[w, 4], lg[w, 5], lg[w, 6] / 100., lg[w, 7] / 100.
    print(f'{w:5d} {dur[w]:7.0f} {nt:11d} {ng:13d} {tt:9.0f} {tg:11.0f} {1e3 * tt / max(nt, 1):13.0f} {1e3 * tg / max(ng, 1):15.0f} {lg[w, 10]:8d} {lg[w, 11]:8d}')
nt, ng, tt, tg = lg[:, 4].sum(), lg[:, 5].sum(), lg[:, 6].sum() / 100., lg[:, 7].sum() / 100.
print(f'tight loop, shader clocks per step: head (update, guesses, LDS, decision) {lg[:, 8].sum() / max(nt, 1):.0f}, tail (commit, division) {lg[:, 9].sum() / max(nt, 1):.0f}')
print(f'rebounds logged by the fan: {int(fan.n_rebounds.sum().item())} ({float(fan.n_rebounds.float().mean().item()):.1f} per ray)')
print(f'all waves: {nt} tight steps in {tt:.0f} wave-us ({1e3 * tt / nt:.0f} ns each), {ng} general iterations in {tg:.0f} wave-us ({1e3 * tg / ng:.0f} ns each)')

