"""Fans at the reference's DEFAULT step sizes (dx_max = 10, dz_max = 0.1, 100 000 steps, no rebound limit:
preferences.py:6-9) on the GPU box: (a) the 10-ray fan of tests_3.ipynb (the reference's second published timing),
(b) a C2 fan of --rays rays through Simulation2D.cast(store=False) + heatmap(), by each of the paths that keep no
trajectories (fused kernel behind a counting pass; chunked: cast with trajectories -> bin -> drop), with lane
utilisation.  python tools/default_res.py [--rays 65536] [--reps 3]"""
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
from ray_tracer_amd import rays as rays_mod   # noqa: E402


def lane_utilisation(n_pts):
    n = int(n_pts.numel())
    pad = (-n) % 64
    p = torch.cat([n_pts, n_pts.new_zeros(pad)]).view(-1, 64)
    return float(n_pts.sum().item()) / float(64 * p.amax(dim=1).sum().item())


def timed(fn, reps):
    fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps):
        out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps, out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--rays', type=int, default=65536)
    ap.add_argument('--reps', type=int, default=3)
    ap.add_argument('--paths', default='default,fused,chunked')
    args = ap.parse_args()
    res = {}
    # (a) tests_3.ipynb
    xs = np.linspace(0., 100000., 4001)
    env3 = rt.Environment2D((xs, -1000. + 0.001 * xs), (np.array([0., 100000.]), np.array([0., 0.])),
                            np.array([0., -1000.]), np.array([100000., 0.]))
    src3 = np.array([0., -200.])
    ang3 = np.linspace(-0.5, 0.3, 10)

    def t3_cast():
        sim = rt.Simulation2D(env3, src3)
        fan = sim.cast(ang3)
        fan.validate()
        return sim, fan

    def t3_map():
        sim, fan = t3_cast()
        return fan, sim.heatmap(resolution=[50, 25], as_tensor=True)
    t_cast, (_, fan3) = timed(t3_cast, args.reps)
    t_map, _ = timed(t3_map, args.reps)
    res['t3_fan10'] = {'ms_per_cast': t_cast * 1e3, 'ms_per_cast_plus_heatmap': t_map * 1e3, 'ray_steps': fan3.ray_steps(),
                       'passes': fan3.passes, 'lane_utilisation': lane_utilisation(fan3.n_pts)}
    print(json.dumps(res['t3_fan10']), flush=True)

    # (b) C2 fan at defaults, nothing stored
    env = bench.make_env(rt)
    source = np.array([0., 0.])
    n = args.rays
    angles = torch.from_numpy(-(np.arange(n) + 1) * (0.999 / n)).cuda()

    def run(path):
        old = (rays_mod.CHUNKED_MIN_RAYS,)
        if path == 'fused':
            rays_mod.CHUNKED_MIN_RAYS = 1 << 40
        elif path == 'chunked':
            rays_mod.CHUNKED_MIN_RAYS = 1
        try:
            sim = rt.Simulation2D(env, source)
            fan = sim.cast(angles, store=False)
            hm = sim.heatmap(resolution=[50, 25], cutoff=.5, as_tensor=True)
            return fan, hm
        finally:
            rays_mod.CHUNKED_MIN_RAYS, = old
    def count_only():
        sim = rt.Simulation2D(env, source)
        fan = sim.cast(angles, store=False)
        fan.validate()
        return fan
    t_count, _ = timed(count_only, args.reps)
    print(json.dumps({'count_pass_s': t_count}), flush=True)
    res['count_pass_s'] = t_count
    ref = None
    for path in args.paths.split(','):
        t, (fan, hm) = timed(lambda: run(path), args.reps)
        steps = bench.ray_steps_of(fan.n_pts, fan.status)
        d = {'path': path, 'rays': n, 's_per_heatmap': t, 'ray_steps': steps, 'ray_steps_per_s': steps / t, 'passes': fan.passes,
             'chunk_rays': fan._chunk_rays() if path != 'fused' else None,
             'lane_utilisation': lane_utilisation(fan.n_pts), 'slow_rays': int(fan.last_slow_rays),
             'fp64_frac_at_48_flop': steps * 48 / t / 1e12 / 78.6,
             'stop_codes': torch.bincount(fan.stop_code, minlength=9).tolist()}
        if ref is None:
            ref = hm
        else:
            d['max_abs_diff_vs_first'] = float((hm - ref).abs().max().item())
        res[path] = d
        print(json.dumps(d), flush=True)
        del fan, hm
    os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
    json.dump(res, open(os.path.join(ROOT, 'gpurun_out', 'default_res.json'), 'w'), indent=1)


if __name__ == '__main__':
    main()
