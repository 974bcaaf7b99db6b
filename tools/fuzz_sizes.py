"""Fan sizes around every threshold of the product path (rays per wave and table placement of the cast, one-pass map from
2^17 rays, chunked map from 2^18) on the C2 and the bathymetry environment: the map of cast() + heatmap_full() -- stored
and store=False -- against the gain array + binning kernels (stream=False), which the tests pin to the oracle.
python tools/fuzz_sizes.py"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch, rt2d_loader, common
rt = rt2d_loader.load()
sizes = [63, 65, 255, 257, 4095, 4097, 16383, 16385, 32767, 32769, 65535, 65537, 131071, 131072, 131073, 262143, 262144, 262145, 300001]
bad = 0
for name in ('c2_fan999', 'c3_bathy999'):
    g = common.load(name)
    env = common.product_env(rt, g)
    kw = common.prop_kwargs(g)
    for n in sizes:
        angles = -(np.arange(n) + 1) * (0.999 / n)
        ref = rt.Simulation2D(env, g['fan_source']); ref.cast(angles, **kw)
        Hr = ref.heatmap_full(resolution=[50, 25], stream=False)[0].cpu().numpy()
        del ref
        out = []
        for store in (True, False):
            sim = rt.Simulation2D(env, g['fan_source'])
            fan = sim.cast(angles, store=store, **kw)
            H = sim.heatmap_full(resolution=[50, 25])[0].cpu().numpy()
            ok = H.shape == Hr.shape and np.array_equal(H == 0, Hr == 0) and np.array_equal(np.isfinite(H), np.isfinite(Hr))
            m = np.isfinite(Hr) & (Hr != 0)
            rel = float((np.abs(H - Hr)[m] / np.abs(Hr[m])).max(initial=0)) if ok else float('inf')
            ok = ok and rel < 1e-11
            bad += 0 if ok else 1
            out.append(f"{'stored' if store else 'store=False'} {fan.passes} rel {rel:.1e}{'' if ok else ' MISMATCH'}")
            del sim, fan
        torch.cuda.empty_cache()
        print(f'{name} n={n}: ' + '; '.join(out), flush=True)
print(f'{bad} mismatches')
