"""Worker of tests/test_gpu_distributed.py: one rank of a 2-rank job on ONE GPU (gloo carries the collectives, both
ranks compute on cuda:0).  Exercises the product's distributed mode end to end: shard -> cast -> C1 -> bin -> C2, and the
sharded sweep (C3).  Usage: python dist_worker.py RANK WORLD PORT OUT_DIR"""

import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))


def main():
    rank, world, port, out_dir = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = port
    os.environ.setdefault('RT2D_HEATMAP_FAST_MIN_RAYS', '0')
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import rt2d_loader
    rt = rt2d_loader.load()
    import common
    out = {}
    for mode in ('interleave', 'block'):
        g = common.load('c2_fan999')
        sim = rt.Simulation2D(common.product_env(rt, g), g['fan_source'], distributed=True, shard=mode)
        fan = sim.cast(g['fan_angles'], **common.prop_kwargs(g))
        H, dropped = sim.heatmap_full(resolution=[50, 25])
        out[f'{mode}_H'] = H.cpu().numpy()
        out[f'{mode}_n_local'] = np.int64(fan.n)
        out[f'{mode}_plot'] = sim.heatmap(resolution=[50, 25], cutoff=.1)
    # one rank's share holds an angle twice (positions 3 and 5 of the interleaved deal are both rank 1's): that rank cannot
    # take the one-read path (the pack keeps the first ray of an angle only) and says so in C1's max; BOTH ranks then take
    # the general path, with its collectives in its order
    g = common.load('c2_fan999')
    dup = np.array(g['fan_angles'][:200], dtype=np.float64)
    dup[5] = dup[3]
    sim = rt.Simulation2D(common.product_env(rt, g), g['fan_source'], distributed=True, shard='interleave')
    sim.cast(dup, sort=False, **common.prop_kwargs(g))
    out['dup_H'] = sim.heatmap_full(resolution=[50, 25])[0].cpu().numpy()
    # a ray slips through the fused binner's mask on ONE rank (its mask is made to vouch for rays that turn back): the
    # kernel counts it, the verdict is shared, and BOTH ranks bin again -- the same collectives in the same order on both
    g = common.load('c2_fan999')
    from ray_tracer_amd import rays as rays_mod
    honest = rays_mod.RayFan.fused_mask
    if rank == 1:
        def vouch_for_all(self, res, shape, ray_mask=None):
            ok = self.status == 0
            if ray_mask is not None:
                ok = ok & (ray_mask != 0)
            return ok.to(torch.uint8)
        rays_mod.RayFan.fused_mask = vouch_for_all
    sim = rt.Simulation2D(common.product_env(rt, g), g['fan_source'], distributed=True, shard='interleave')
    fan = sim.cast(g['fan_angles'], store=False, **common.prop_kwargs(g))
    out['viol_H'] = sim.heatmap_full(resolution=[50, 25])[0].cpu().numpy()
    out['viol_redone'] = np.int64(1 if fan.__dict__.get('_exact_binning') else 0)
    rays_mod.RayFan.fused_mask = honest
    g = common.load('c5_eigen')
    sim = rt.EigenraySim2D(common.product_env(rt, g), g['source'], g['target'], int(g['n_rays']), int(g['n_rebounds_max']),
                           n_rays_scan=int(g['n_rays_scan']), scan_angle_min=float(g['scan_angle_min']),
                           scan_angle_max=float(g['scan_angle_max']), dx_max=float(g['dx_max']), dz_max=float(g['dz_max']),
                           distributed=True)
    out['scan_selected'] = np.array([r.angle for r in sim.raypacks['scan'].rays])
    out['scan_n_local'] = np.int64(sim.scan_fan.n)
    sim.refine(iterations=2, n_rays=5)
    out['final_angles'] = np.array([r.angle for r in sim.raypacks[sim.pack_default].rays])
    out['final_dist_avg'] = np.float64(sim.dist_avg)
    np.savez(os.path.join(out_dir, f'rank{rank}.npz'), **out)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
