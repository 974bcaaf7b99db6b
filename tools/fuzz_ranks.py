"""The distributed mode on random environments: two ranks on one card (gloo carries the collectives) cast their shares of
a random fan -- interleaved or in blocks, stored or not -- and combine plot range and heat map; every rank then computes
the same map alone.  python tools/fuzz_ranks.py [first] [last]  (starts its own two ranks)"""
import os, sys, subprocess, socket
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))


def rank_main(rank, world, first, last):
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import rt2d_loader
    import test_gpu_random as t
    rt = rt2d_loader.load()
    bad = []
    for seed in range(first, last):
        rng = np.random.default_rng(11000 + seed)
        env, source, target, angles, kw = t._random_case(rt, rng)
        kw['n_steps_max'] = min(kw['n_steps_max'], 2000)
        angles = np.linspace(float(np.min(angles)), float(np.max(angles)), int(rng.integers(40, 700)))
        res = [int(v) for v in rng.choice([[50, 25], [20, 10], [10, 5]])]
        freq = float(rng.choice([1., 500.]))
        mode = str(rng.choice(['interleave', 'block']))
        store = bool(rng.integers(0, 2))
        try:
            sim = rt.Simulation2D(env, source, distributed=True, shard=mode)
            fan = sim.cast(angles, store=store, **kw)
            Hd = sim.heatmap_full(resolution=res, freq=freq)[0].cpu().numpy()
            nd = sim.heatmap(resolution=res, freq=freq)
            one = rt.Simulation2D(env, source)
            one.cast(angles, **kw)
            Hs = one.heatmap_full(resolution=res, freq=freq, stream=False)[0].cpu().numpy()
            ns = one.heatmap(resolution=res, freq=freq)
            assert Hd.shape == Hs.shape, (Hd.shape, Hs.shape)
            assert np.array_equal(Hd == 0, Hs == 0) and np.array_equal(np.isfinite(Hd), np.isfinite(Hs)), 'occupancy'
            m = np.isfinite(Hs) & (Hs != 0)
            rel = float((np.abs(Hd - Hs)[m] / np.abs(Hs[m])).max(initial=0))
            assert rel < 1e-11, rel
            assert nd.shape == ns.shape and np.array_equal(np.isnan(nd), np.isnan(ns)) and float(np.nanmax(np.abs(nd - ns), initial=0)) < 1e-9
            if rank == 0:
                print(f'seed {seed}: {len(angles)} rays ({fan.n} here), {mode}, store={store}, map {Hs.shape}, cells to {rel:.1e}', flush=True)
        except AssertionError as e:
            bad.append(seed); print(f'seed {seed}: MISMATCH on rank {rank}: {str(e)[:200]}', flush=True)
    flag = torch.tensor([float(len(bad))])
    dist.all_reduce(flag)
    if rank == 0:
        print(f'{last - first} seeds, {int(flag.item())} mismatches {bad}', flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
    if 'RANK' in os.environ:
        rank_main(int(os.environ['RANK']), int(os.environ['WORLD_SIZE']), int(sys.argv[1]), int(sys.argv[2]))
    else:
        first = sys.argv[1] if len(sys.argv) > 1 else '0'
        last = sys.argv[2] if len(sys.argv) > 2 else '30'
        with socket.socket() as sk:
            sk.bind(('127.0.0.1', 0)); port = sk.getsockname()[1]
        procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), first, last],
                                  env=dict(os.environ, RANK=str(r), WORLD_SIZE='2', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port)))
                 for r in range(2)]
        sys.exit(max(p.wait() for p in procs))
