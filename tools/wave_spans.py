"""Development aid (needs a -DRT2D_WAVE_LOG build, RT2D_SO=...): per-wave durations of the count pass and of the fused
kernel of a C2 fan at the reference's default step sizes -- how uneven the waves of one launch are.
python tools/wave_spans.py [rays]"""
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

n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
env = bench.make_env(rt)
L = _lib.lib()
L.rt2d_debug_wave_log.argtypes = [C.c_void_p, C.c_int64]
L.rt2d_debug_wave_log.restype = None
angles = torch.from_numpy(-(np.arange(n) + 1) * (0.999 / n)).cuda()
source = np.array([0., 0.])
cap = 1 << 16
log = torch.zeros((cap, 12), dtype=torch.int64, device='cuda')


def show(name, lo):
    lg = log.cpu().numpy()
    idx = np.nonzero(lg[:, 1] > 0)[0]
    idx = idx[idx >= lo]
    lg = lg[idx]
    dur = (lg[:, 1] - lg[:, 0]) / 100.
    span = (lg[:, 1].max() - lg[:, 0].min()) / 100.
    q = np.percentile(dur, [0, 10, 50, 90, 99, 100])
    print(f'{name}: {len(lg)} waves (slots >= {lo}), span {span / 1e3:.1f} ms, wave durations ms: min {q[0] / 1e3:.1f} p10 {q[1] / 1e3:.1f} '
          f'median {q[2] / 1e3:.1f} p90 {q[3] / 1e3:.1f} p99 {q[4] / 1e3:.1f} max {q[5] / 1e3:.1f}; mean {dur.mean() / 1e3:.1f}')
    order = np.argsort(-dur)[:12]
    print('   longest waves (slot: ms):', ', '.join(f'{int(idx[o])}: {dur[o] / 1e3:.0f}' for o in order))
    return idx, dur


sim = rt.Simulation2D(env, source)
fan = sim.cast(angles, store=False); fan.validate()          # warm
torch.cuda.synchronize()
L.rt2d_debug_wave_log(log.data_ptr(), cap)
sim = rt.Simulation2D(env, source)
fan = sim.cast(angles, store=False); fan.validate()
torch.cuda.synchronize()
L.rt2d_debug_wave_log(None, 0)
idx, dur = show('count pass', 0)
nr = fan.n_rebounds.cpu().numpy().astype(np.int64)
per_wave = np.add.reduceat(nr, np.arange(0, n, 64))
print('   rebounds per wave of 64 rays: median', int(np.median(per_wave)), 'p90', int(np.percentile(per_wave, 90)), 'max', int(per_wave.max()),
      '; correlation of wave duration with its rebounds:', float(np.corrcoef(per_wave[idx], dur)[0, 1]) if len(idx) == len(per_wave) else 'n/a')
if len(idx) == len(per_wave):
    A = np.stack([np.ones(len(dur)), per_wave[idx]], 1)
    coef = np.linalg.lstsq(A, dur, rcond=None)[0]
    print(f'   fit: wave ms = {coef[0] / 1e3:.1f} + {coef[1]:.2f} us x rebounds in the wave')
log.zero_()
L.rt2d_debug_wave_log(log.data_ptr(), cap)
hm = sim.heatmap(resolution=[50, 25], cutoff=.5, as_tensor=True)
torch.cuda.synchronize()
L.rt2d_debug_wave_log(None, 0)
idx2, dur2 = show('fused kernel (slots below 256 may be the side rays\' cast)', 256)
mask = fan.fused_mask().cpu().numpy() if hasattr(fan, 'fused_mask') else None
if mask is not None:
    nrm = np.where(mask != 0, nr, 0)
    pw = np.add.reduceat(nrm, np.arange(0, n, 64))
    A = np.stack([np.ones(len(dur2)), pw[idx2]], 1)
    coef = np.linalg.lstsq(A, dur2, rcond=None)[0]
    print(f'   fit: wave ms = {coef[0] / 1e3:.1f} + {coef[1]:.2f} us x rebounds of its binned rays; correlation {float(np.corrcoef(pw[idx2], dur2)[0, 1]):.3f}')
