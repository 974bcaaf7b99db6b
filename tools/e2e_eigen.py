"""Wall time of BASELINE config 5 through the product API: EigenraySim2D sweep of 1M rays, the 1000 best kept, three
refine iterations of 5 rays per cone.  Run on the GPU box: python tools/e2e_eigen.py [n_scan]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rt2d_loader
import bench
rt = rt2d_loader.load()
env = bench.make_env(rt)
n_scan = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    sim = rt.EigenraySim2D(env, np.array([0., 0.]), np.array([40000., -600.]), 1000, 20, n_rays_scan=n_scan,
                           scan_angle_min=-0.999, scan_angle_max=-0.001, dx_max=50., dz_max=25.)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    d0 = sim.dist_avg
    sim.refine(iterations=3, n_rays=5, force_refine=True)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f'run {rep}: sweep of {n_scan} rays + keep 1000: {1e3 * (t1 - t0):8.2f} ms (mean distance {d0:.4g} m); '
          f'3 refines: {1e3 * (t2 - t1):8.2f} ms (mean distance {sim.dist_avg:.4g} m)', flush=True)
