"""What the refine cones of config 5 cost with and without the post-pass in the stepping loop: the cone's angles cast as a
fan with a target (distance to target, travel time: the whole post-pass rides along) and without one.
python tools/refine_cone_cost.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, rt2d_loader, bench
rt = rt2d_loader.load()
env = bench.make_env(rt)
src, tgt = np.array([0., 0.]), np.array([40000., -600.])
sim = rt.EigenraySim2D(env, src, tgt, 1000, 20, n_rays_scan=1 << 18, scan_angle_min=-0.999, scan_angle_max=-0.001, dx_max=50., dz_max=25.)
sim.refine(iterations=1, n_rays=5, force_refine=True)
fan = sim.last_refine_fan if hasattr(sim, 'last_refine_fan') else None
angles = None
for name, pack in sim.raypacks.items():
    if name.startswith('refine'):
        segs = pack.segments()
        angles = segs[0][0].angles_dev.clone()
        kw = dict(segs[0][0].kw)
print('cone rays', int(angles.numel()), kw)
for label, target in (('with target', tgt), ('without', None)):
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        f = rt.RayFan(env, src, angles, target=target, store=False, **kw)
        f.validate(); torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    npts = f.host('n_pts')
    print(f'{label}: {1e3 * dt:.1f} ms; longest ray {int(npts.max())} points, rays over 50 000 points: {int((npts > 50000).sum())}, rebounds of the longest: {int(f.host("n_rebounds")[int(npts.argmax())])}, monotone: {int(f.host("monotone")[int(npts.argmax())])}')
