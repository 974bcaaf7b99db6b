"""Which rays of the 1M-ray bench fan does the tile heat-map kernel hand to the general kernel, and why?
Run on the GPU box: python tools/diag_irregular.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rt2d_loader
import bench
rt = rt2d_loader.load()
env = bench.make_env(rt)
n = 1 << 18
angles = -(np.arange(n) + 1) * (0.999 / n)
fan = rt.RayFan(env, [0., 0.], angles, **bench.PROP)
mono = fan.monotone.cpu().numpy() != 0
stop = fan.host('stop_code')
rng = fan.range.cpu().numpy()
print('rays', n, 'non-monotone', int((~mono).sum()))
for c in range(9):
    m = stop == c
    if m.any():
        print(f'stop {c}: {int(m.sum())} rays, non-monotone {int((m & ~mono).sum())}')
print('out of map by range: x<=-1', int((rng[0] <= -1).sum()), 'z>=1', int((rng[3] >= 1).sum()))
bad = np.nonzero(~mono)[0]
tiles = np.unique(bad // 64)
print('tiles with a non-monotone ray', len(tiles), 'of', (n + 63) // 64, '; rays per such tile', len(bad) / max(1, len(tiles)))
print('angle range of non-monotone rays', angles[bad].min(), angles[bad].max())
hist = {}
for i in bad[:: max(1, len(bad) // 400)]:
    xz = rt.Ray2D._from_fan(fan, int(i)).XZ
    back = np.nonzero(np.diff(xz[:, 0]) < 0)[0]
    key = (len(back), 'last' if len(back) and back[-1] == len(xz) - 2 else 'inner', int(stop[i]))
    hist[key] = hist.get(key, 0) + 1
print('sample of non-monotone rays: (backward steps, where, stop) -> count')
for k, v in sorted(hist.items(), key=lambda kv: -kv[1])[:12]:
    print(' ', k, v)

# how far behind its own column frontier does a ray step, and how many columns does one step jump?
res_x = 50
cnt = {'col-monotone': 0, 'P&P1': 0, 'other': 0}
worst_back = 0
for i in bad[:: max(1, len(bad) // 2000)]:
    xz = rt.Ray2D._from_fan(fan, int(i)).XZ
    col = np.trunc(xz[:, 0]).astype(np.int64) // res_x
    front = np.maximum.accumulate(col)
    behind = int((front - col).max())
    jump = int(np.abs(np.diff(col)).max())
    worst_back = max(worst_back, behind)
    if behind == 0:
        cnt['col-monotone'] += 1
    elif behind <= 1 and jump <= 1:
        cnt['P&P1'] += 1
    else:
        cnt['other'] += 1
print('column pattern of sampled non-monotone rays:', cnt, 'worst columns behind the frontier', worst_back)
