"""The count pass (cast without trajectories) of a C2 fan at the reference's default step sizes, for counter passes:
python tools/count_default.py [--rays 65536] [--steps 10000] [--reps 3].  16 384 rays = one wave per CU, 65 536 = four."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch          # noqa: E402
import rt2d_loader    # noqa: E402
import bench          # noqa: E402

rt = rt2d_loader.load()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--rays', type=int, default=65536)
    ap.add_argument('--steps', type=int, default=10000)
    ap.add_argument('--reps', type=int, default=3)
    ap.add_argument('--store', type=int, default=0)
    ap.add_argument('--dx', type=float, default=None)
    ap.add_argument('--dz', type=float, default=None)
    ap.add_argument('--rebounds', type=int, default=None)
    args = ap.parse_args()
    env = bench.make_env(rt)
    n = args.rays
    angles = torch.from_numpy(-(np.arange(n) + 1) * (0.999 / n)).cuda()
    out = []
    for _ in range(args.reps + 1):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        sim = rt.Simulation2D(env, np.array([0., 0.]))
        kw = {}
        if args.dx: kw['dx_max'] = args.dx
        if args.dz: kw['dz_max'] = args.dz
        if args.rebounds is not None: kw['n_rebounds_max'] = args.rebounds
        fan = sim.cast(angles, store=bool(args.store), n_steps_max=args.steps, **kw)
        fan.validate()
        torch.cuda.synchronize()
        out.append(time.perf_counter() - t0)
    steps = bench.ray_steps_of(fan.n_pts, fan.status)
    print(json.dumps({'rays': n, 'n_steps_max': args.steps, 's_per_cast': min(out[1:]), 'ray_steps': steps,
                      'us_per_wave_step': min(out[1:]) / args.steps * 1e6, 'passes': fan.passes}))


if __name__ == '__main__':
    main()
