"""One storing cast of the 1M-ray (or --rays) C2 fan, for `rocprofv3 --pmc WRITE_SIZE` passes under the stepping kernel's
switches (RT2D_CAST_TIGHT, RT2D_CAST_TAB): VERDICT r2 item 4, where do the bytes beyond 16 B per point come from.
Prints the algorithmic bytes, the bytes of the padded-to-the-tile's-longest-ray layout and of the rebound log entries."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, rt2d_loader, bench
rt = rt2d_loader.load()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
env = bench.make_env(rt)
angles = torch.from_numpy(-(np.arange(n) + 1) * (0.999 / n)).cuda()
for rep in range(2):
    fan = rt.RayFan(env, np.array([0., 0.]), angles, **bench.PROP)
    fan.validate()
torch.cuda.synchronize()
npts = fan.n_pts.to(torch.int64)
pad = (-n) % 64
tiles = torch.cat([npts, npts.new_zeros(pad)]).view(-1, 64)
alg = int(npts.sum().item()) * 16
tile_rows = int(tiles.amax(dim=1).sum().item()) * 64 * 16
reb = int(torch.minimum(fan.n_rebounds, torch.full_like(fan.n_rebounds, fan.rebound_cap)).sum().item())
print(f'rays {n} passes {fan.passes}: points {int(npts.sum().item())}, algorithmic {alg / 1e9:.3f} GB, whole rows of every tile up to its '
      f'longest ray {tile_rows / 1e9:.3f} GB, rebound log entries {reb} x 37 B = {reb * 37 / 1e9:.3f} GB, per-ray results {n * 61 / 1e9:.3f} GB')
