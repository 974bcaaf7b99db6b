import os, sys, cProfile, pstats
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, rt2d_loader, bench
rt = rt2d_loader.load()
env = bench.make_env(rt)
sim = rt.EigenraySim2D(env, np.array([0., 0.]), np.array([40000., -600.]), 1000, 20, n_rays_scan=1 << 18,
                       scan_angle_min=-0.999, scan_angle_max=-0.001, dx_max=50., dz_max=25.)
pr = cProfile.Profile(); pr.enable()
sim.refine(iterations=2, n_rays=5, force_refine=True)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(28)
