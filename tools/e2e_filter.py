"""Wall time of the eigenray products through the product API: EigenraySim2D.target_products (T_target, Tmult_target[F]
per ray) and get_filter_NEW on a synthetic signal.  Run on the GPU box: python tools/e2e_filter.py [n_rays] [n_freqs]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rt2d_loader
import bench
rt = rt2d_loader.load()
env = bench.make_env(rt)
n_rays = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
F = int(sys.argv[2]) if len(sys.argv) > 2 else 64
sim = rt.EigenraySim2D(env, np.array([0., 0.]), np.array([40000., -600.]), n_rays, 20, n_rays_scan=1 << 18,
                       scan_angle_min=-0.999, scan_angle_max=-0.001, dx_max=50., dz_max=25.)
freqs = list(np.linspace(1, 20000, F))
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ang, T_t, Tm_t = sim.target_products(*freqs)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    print(f'run {rep}: target_products of {len(ang)} rays x {F} frequencies: {1e3 * (t1 - t0):9.2f} ms; '
          f'T_target {T_t.min():.4f} .. {T_t.max():.4f} s', flush=True)
