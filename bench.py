"""bench.py -- the reference's headline workload on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one fan: cast (count pass, tile layout, store pass) ->
gain at 1 Hz -> heat-map accumulation -> normalise/clip/power, everything on the device, inputs (the angle
array and the environment tables) resident in HBM before the timed region.  At N = 1 the fan is BASELINE
config 1: 999 rays (angles -i/1000), 20 rebounds, dx_max 50 / dz_max 25, floor max(-e^{0.0005x}-500,
-1500+0.02x) over 75 km.  At N > 1 every rank casts 999 rays of the N-times denser fan -(j+1)/(1000 N)
(ray j on rank j mod N, weak scaling); the plot range and the heat map are combined with the two RCCL
all-reduces of ray-tracer_amd/parallel.py.

Prints ONE JSON line on rank 0: metric ray-steps/s (whole job), per-stage kernel times from HIP events,
`roofline` of the dominant kernel (the storing cast: 48 flop per ray-step against the FP64 vector ALU first, 16 B
per ray-step against HBM second; the PMC traffic only from a committed profile of THIS build), a `large_fan` block
(the same pipeline on --large-rays rays, where the GPU is actually full), `large_fan_fused` (the same map without
keeping trajectories) and `cpu_baseline` (oracle/rt2d_oracle.c, scalar, 1 core, same fan).
"""

import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
FP64_VECTOR_PEAK_TFLOPS = 78.6 # vendor figure
BYTES_PER_STEP = 16            # x, z as f64 per appended point, storing cast
FLOP_PER_STEP = 48


def c2_floor(x):
    return np.maximum(-1 * np.exp(0.0005 * x) - 500, -1500 + 0.02 * x)


def make_env(rt):
    xs = np.linspace(0., 75000., 3001)
    return rt.Environment2D((xs, c2_floor(xs)), (np.array([0., 75000.]), np.array([0., 0.])),
                            np.array([0., -1501.]), np.array([75000., 1.]))


PROP = dict(dx_max=50., dz_max=25., n_rebounds_max=20)
RES = (50, 25)
CUTOFF = .5      # BASELINE config 2 "raypower cutoff=0.5": the heat map's saturation level (simulation.py:134-136)


class Stages:
    """HIP-event timing of the device stages of one step (events on torch's current stream, which is the
    stream every rt2d_* launch uses)."""
    NAMES = ('cast_count', 'layout', 'cast_store', 'layout_compact', 'gain', 'heatmap', 'finalize')

    def __init__(self, torch):
        self.torch = torch
        self.records = []

    def new_step(self):
        ev = [self.torch.cuda.Event(enable_timing=True) for _ in range(len(self.NAMES) + 1)]
        self.records.append(ev)
        return ev

    def summary(self):
        out = {}
        for k, name in enumerate(self.NAMES):
            out[name] = float(np.mean([ev[k].elapsed_time(ev[k + 1]) for ev in self.records]))
        return out


def one_step(rt, torch, env, source, angles_dev, stages=None, collectives=False, device=None):
    """The hot path once.  Returns (ray_steps, heat map (nz, nx) device tensor)."""
    import torch.distributed  # noqa: F401
    from ray_tracer_amd import _lib, parallel, rays
    L = _lib.lib()
    stream = torch.cuda.current_stream().cuda_stream
    ev = stages.new_step() if stages is not None else None
    n = int(angles_dev.shape[0])
    f64 = dict(dtype=torch.float64, device=device)
    i32 = dict(dtype=torch.int32, device=device)
    handle = env.device_handle(device.index)
    n_pts, stop, nreb, status = (torch.empty(n, **i32) for _ in range(4))
    rng = torch.empty((4, n), **f64)
    out = _lib.Rt2dRayOut()
    out.n_pts, out.stop_code, out.n_rebounds, out.status = n_pts.data_ptr(), stop.data_ptr(), nreb.data_ptr(), status.data_ptr()
    out.range = rng.data_ptr()
    mono = torch.empty(n, dtype=torch.uint8, device=device)
    out.monotone = mono.data_ptr()
    smax = torch.empty(n, **f64)
    out.slope_max = smax.data_ptr()
    src = (C.c_double * 2)(float(source[0]), float(source[1]))
    p = _lib.Rt2dParams()
    p.dx_max, p.dz_max, p.n_steps_max, p.n_rebounds_max, p.backprop = PROP['dx_max'], PROP['dz_max'], 100000, PROP['n_rebounds_max'], 1
    cap = PROP['n_rebounds_max'] + 1
    p.rebound_cap = cap
    out.reb_step = None
    n_tiles = (n + 63) // 64
    tile_off = torch.empty(n_tiles + 1, dtype=torch.int64, device=device)
    total = C.c_int64(0)
    reb_step = torch.empty((cap, n), **i32)
    reb_surface = torch.empty((cap, n), dtype=torch.uint8, device=device)
    reb_vel, reb_ang, reb_px, reb_pz = (torch.empty((cap, n), **f64) for _ in range(4))

    def with_log(o):
        o.reb_step, o.reb_surface, o.reb_velocity = reb_step.data_ptr(), reb_surface.data_ptr(), reb_vel.data_ptr()
        o.reb_angle, o.reb_pos_x, o.reb_pos_z = reb_ang.data_ptr(), reb_px.data_ptr(), reb_pz.data_ptr()
        return o

    summ = torch.empty(8, **f64)

    def summarise(offsets):
        """rt2d_fan_summary: [points, longest tile, log overflow, error-free rays, plot range x4] in one 64-byte read"""
        _lib.check(L.rt2d_fan_summary(status.data_ptr(), rng.data_ptr(), None, n, offsets.data_ptr(), summ.data_ptr(), stream))
        return summ.tolist()

    sm = None
    room = 4096
    speculative = n_tiles * room * 64 * 16 <= (1 << 30)          # same policy as RayFan (ray-tracer_amd/rays.py)
    padded = (not speculative and n_tiles * room * 64 * 16 <= rays.PADDED_BYTES
              and n_tiles * room * 64 * 16 * 1.6 <= torch.cuda.mem_get_info(device)[0])
    done = False
    if ev: ev[0].record()
    if padded:
        # big fan: ONE storing pass into a padded layout that is kept (288 GB of HBM: untouched pages cost nothing)
        if ev: ev[1].record(); ev[2].record()
        tile_off = torch.arange(n_tiles + 1, dtype=torch.int64, device=device) * (room * 64)
        total = n_tiles * room * 64
        X, Z = torch.empty(total, **f64), torch.empty(total, **f64)
        _lib.check(L.rt2d_cast_write(handle, angles_dev.data_ptr(), n, src, C.byref(p), None, None, C.byref(with_log(out)),
                                     tile_off.data_ptr(), X.data_ptr(), Z.data_ptr(), None, stream))
        if ev: ev[3].record()
        exact = torch.empty(n_tiles + 1, dtype=torch.int64, device=device)
        _lib.check(L.rt2d_traj_layout(n_pts.data_ptr(), n, exact.data_ptr(), None, stream))
        sm = summarise(exact)
        done = sm[1] <= room * 64 and not sm[2]
        if not done:
            del X, Z
            tile_off = torch.empty(n_tiles + 1, dtype=torch.int64, device=device)
            total = C.c_int64(0)
    if speculative:
        # small fan: one storing pass into a provisional layout, exact layout, compaction copy
        if ev: ev[1].record(); ev[2].record()
        off0 = torch.arange(n_tiles + 1, dtype=torch.int64, device=device) * (room * 64)
        X0, Z0 = torch.empty(n_tiles * room * 64, **f64), torch.empty(n_tiles * room * 64, **f64)
        _lib.check(L.rt2d_cast_write(handle, angles_dev.data_ptr(), n, src, C.byref(p), None, None, C.byref(with_log(out)),
                                     off0.data_ptr(), X0.data_ptr(), Z0.data_ptr(), None, stream))
        if ev: ev[3].record()
        _lib.check(L.rt2d_traj_layout(n_pts.data_ptr(), n, tile_off.data_ptr(), None, stream))
        sm = summarise(tile_off)
        if sm[1] <= room * 64 and not sm[2]:
            total = int(sm[0])
            X, Z = torch.empty(total, **f64), torch.empty(total, **f64)
            _lib.check(L.rt2d_traj_compact(off0.data_ptr(), tile_off.data_ptr(), n, X0.data_ptr(), Z0.data_ptr(), None,
                                           X.data_ptr(), Z.data_ptr(), None, stream))
            done = True
    if not done:
        cnt = _lib.Rt2dRayOut()
        cnt.n_pts, cnt.n_rebounds = n_pts.data_ptr(), nreb.data_ptr()
        _lib.check(L.rt2d_cast_count(handle, angles_dev.data_ptr(), n, src, C.byref(p), None, None, C.byref(cnt), stream))
        if ev: ev[1].record()
        _lib.check(L.rt2d_traj_layout(n_pts.data_ptr(), n, tile_off.data_ptr(), C.byref(total), stream))
        total = int(total.value)
        if ev: ev[2].record()
        X, Z = torch.empty(total, **f64), torch.empty(total, **f64)
        _lib.check(L.rt2d_cast_write(handle, angles_dev.data_ptr(), n, src, C.byref(p), None, None, C.byref(with_log(out)),
                                     tile_off.data_ptr(), X.data_ptr(), Z.data_ptr(), None, stream))
        if ev: ev[3].record()
        sm = summarise(tile_off)
    if ev: ev[4].record()
    # RT2D_BENCH_ONE_PASS=1: gain + binning in ONE pass over the trajectories (RayFan.heatmap_accumulate_stream) for big fans;
    # slower than the two kernels today (17.5 vs 15.1 ms per 1M rays), kept as the form that needs no gain array
    one_pass = n > 16384 and bool(os.environ.get('RT2D_BENCH_ONE_PASS'))
    if not one_pass:
        G = torch.empty(total, **f64)
        freq = torch.ones(1, **f64)
        loss_ws = torch.empty(cap * n, **f64)
        _lib.check(L.rt2d_gain(handle, X.data_ptr(), Z.data_ptr(), tile_off.data_ptr(), n_pts.data_ptr(), C.byref(out), cap, n,
                               freq.data_ptr(), 1, total, G.data_ptr(), None, loss_ws.data_ptr(), stream))
    if ev: ev[5].record()
    # plot range of the rays that did not fail (RayPack2D.add fold, origin included) -> heat-map shape: already in the
    # summary read above; with several ranks one max-all-reduce of the four numbers (C1) and one more read
    okm = (status == 0)
    if collectives:
        ext = summ[4:8].clone()
        parallel._all_reduce(ext, torch.distributed.ReduceOp.MAX)              # C1
        ext = ext.cpu().numpy()
    else:
        ext = np.array(sm[4:8])
    lo, hi = -ext[:2], ext[2:]
    shape = parallel.heatmap_shape(lo, hi, RES)
    H = torch.zeros(int(shape[0]) * int(shape[1]), **f64)
    res_c = (C.c_int32 * 2)(*RES)
    shape_c = (C.c_int32 * 2)(int(shape[0]), int(shape[1]))
    if one_pass:
        # 1 = the one-pass kernel bins the ray (x never decreases, inside the map, <= 8 rows per column: RayFan.fused_mask);
        # 2 = it only stores the ray's gain and the general kernel bins it afterwards
        inside = ((rng[0] > -1) & (rng[3] < 1) & (torch.div(torch.trunc(rng[2]), RES[0], rounding_mode='floor') < int(shape[0]))
                  & (torch.div(-torch.trunc(rng[1]), RES[1], rounding_mode='floor') < int(shape[1])))
        regular = okm & inside & (mono != 0) & (torch.floor(smax * (RES[0] / RES[1])) + 2 <= 8)
        rest = okm & ~regular
        mask = regular.to(torch.uint8) + 2 * rest.to(torch.uint8)
        n_rest = int(rest.sum().item())
        G = torch.empty(total, **f64) if n_rest else None
        viol = torch.zeros(1, dtype=torch.int64, device=device)
        _lib.check(L.rt2d_gain_heatmap(handle, X.data_ptr(), Z.data_ptr(), tile_off.data_ptr(), n_pts.data_ptr(), C.byref(out), cap,
                                       n, 1.0, mask.data_ptr(), res_c, shape_c, H.data_ptr(), G.data_ptr() if n_rest else None,
                                       None, None, viol.data_ptr(), torch.empty(cap * n, **f64).data_ptr(), stream))
        if n_rest:
            _lib.check(L.rt2d_heatmap_accumulate(X.data_ptr(), Z.data_ptr(), G.data_ptr(), tile_off.data_ptr(), n_pts.data_ptr(), n,
                                                 rest.to(torch.uint8).data_ptr(), res_c, shape_c, H.data_ptr(), None, None, stream))
    else:
        # ray_mask with the hint bit: x never decreases (from the cast) and the ray's bounding box is inside the map
        mask = okm.to(torch.uint8)
        if n > 6000:           # the tile kernel's range (rt2d_heatmap_accumulate); smaller fans use the general kernel
            inside = ((rng[0] > -1) & (rng[3] < 1) & (torch.div(torch.trunc(rng[2]), RES[0], rounding_mode='floor') < int(shape[0]))
                      & (torch.div(-torch.trunc(rng[1]), RES[1], rounding_mode='floor') < int(shape[1])))
            mask = mask * (1 + 2 * (inside & (mono != 0)).to(torch.uint8))
        work = torch.empty(n + 2, **i32)
        _lib.check(L.rt2d_heatmap_accumulate(X.data_ptr(), Z.data_ptr(), G.data_ptr(), tile_off.data_ptr(), n_pts.data_ptr(), n,
                                             mask.data_ptr(), res_c, shape_c, H.data_ptr(), None, work.data_ptr(), stream))
    if collectives:
        parallel.allreduce_heatmap(H)                                          # C2
    if ev: ev[6].record()
    outp = torch.empty(int(shape[0]) * int(shape[1]), **f64)
    scratch = torch.empty(2, **f64)
    _lib.check(L.rt2d_heatmap_finalize(H.data_ptr(), shape_c, CUTOFF, .5, outp.data_ptr(), scratch.data_ptr(), stream))
    if ev: ev[7].record()
    return total, n_pts, status, outp.view(int(shape[1]), int(shape[0]))


def product_step(rt, env, source, angles, distributed=False):
    """The hot path once THROUGH THE PRODUCT API, as a user of the reference's Simulation2D would write it
    (simulation.py:69-94 cast, then the heat map): returns (fan, (nz, nx) device tensor)."""
    sim = rt.Simulation2D(env, source, distributed=distributed)
    fan = sim.cast(angles, **PROP)
    hm = sim.heatmap(resolution=list(RES), cutoff=CUTOFF, reduction_power=.5, as_tensor=True)
    return fan, hm


def pmc_traffic(large):
    """HBM bytes per launch of the storing cast from the newest committed PMC profile (profiles/r*/*_traffic.json,
    written by tools/summarize_profile.py from separate FETCH_SIZE / WRITE_SIZE passes of this same command, with the
    gfx950 corrections of MI355X_MICROARCH.md).  The 999-ray and the 1M-ray launches differ by grid size."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*', '*_traffic.json')))
    if not files:
        return None, 'no PMC profile committed under profiles/'
    with open(files[-1]) as f:
        prof = json.load(f)
    stale = profile_is_stale(prof)
    if stale:
        return None, f'{os.path.relpath(files[-1], ROOT)}: {stale}'
    k = prof['kernels']
    # the launch is told apart by its grid: the 1M-ray fan is one thread per ray; the 999-ray fan runs as waves of 4 rays (one
    # wave per CU on 256 CUs: launch_cast2), i.e. 250 waves = 16 000 threads.  (Other blocks of this file launch the same
    # kernel with other grids -- the 10-ray default-step fan, the refine cones -- and the profile passes skip the bathymetry
    # fan, whose launches have the 1M-ray fan's grid and 1.44 x its points: --profile-pass.)
    want = large if isinstance(large, int) and large > 1 else ((1 << 20) if large else 16000)
    cand = [v for name, v in k.items() if name.startswith('rt2d_cast_kernel<true') and ', 1, false, true, 0>' in name.split('|')[0]
            and int(name.split('grid=')[1].split('|')[0]) == want and v.get('hbm_bytes') is not None]
    if not cand:
        return None, f'{os.path.relpath(files[-1], ROOT)} has no FETCH_SIZE / WRITE_SIZE pass of the storing cast with grid {want}'
    return cand[0]['hbm_bytes'], os.path.relpath(files[-1], ROOT)


def profile_is_stale(prof):
    """None when the committed profile measured THIS build (hash of the kernel sources, stamped by tools/profile_gpu.sh),
    else the reason its counters are not reported."""
    import rt2d_loader
    here = rt2d_loader.load()._lib.source_hash()
    b = prof.get('build')
    if not b or not b.get('source_sha256'):
        return 'profile carries no build stamp (taken before round 2)'
    if b['source_sha256'] != here:
        return (f"profile measured kernel sources {b['source_sha256']} (commit {str(b.get('git_head'))[:10]}), "
                f"this build is {here}: counters not carried over")
    return None


def pmc_binning():
    """Per-launch HBM bytes and L2 atomic requests of the two binning kernels on the 1M-ray fan, from the newest committed
    PMC passes (profiles/r*/*_traffic.json, *_counters.json: FETCH_SIZE / WRITE_SIZE / TCC_ATOMIC_sum passes of this command)."""
    import glob
    out = {'hbm_bytes': None, 'atomics': None, 'source': None}
    tf = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*', '*_traffic.json')))
    cf = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*', '*_counters.json')))
    def big(k, names):
        rows = [v for name, v in k.items() if name.split('|')[0] in names and int(name.split('grid=')[1].split('|')[0]) >= (1 << 20)]
        return rows
    names = ('rt2d_heatmap_fast_kernel<1>', 'rt2d_heatmap_kernel')
    if tf:
        prof = json.load(open(tf[-1]))
        stale = profile_is_stale(prof)
        if stale:
            out['source'] = f'{os.path.relpath(tf[-1], ROOT)}: {stale}'
            return out
        rows = big(prof['kernels'], names)
        if rows and all(r.get('hbm_bytes') is not None for r in rows):
            out['hbm_bytes'] = sum(r['hbm_bytes'] for r in rows)
            out['source'] = os.path.relpath(tf[-1], ROOT)
    if cf:
        rows = big(json.load(open(cf[-1]))['kernels'], names)
        if rows and all('TCC_ATOMIC_sum' in r for r in rows):
            out['atomics'] = sum(r['TCC_ATOMIC_sum'] for r in rows)
    return out


def roofline_block(kernel, steps, ms, traffic, traffic_src, note=None):
    """The stepping kernel is priced against the FP64 vector ALU first (48 flop per ray-step, no MFMA shape in
    it), the HBM roof of its storing instance second (16 B per ray-step).  `traffic`: HBM bytes per launch from the PMC passes
    of the committed profile of THIS build, else null with the reason in traffic_source."""
    t = ms * 1e-3
    tf = steps * FLOP_PER_STEP / t / 1e12
    gb = steps * BYTES_PER_STEP / t / 1e9
    r = {'kernel': kernel, 'bound': 'fp64_vector', 'achieved': tf, 'peak': FP64_VECTOR_PEAK_TFLOPS, 'unit': 'TFLOP/s',
         'frac': tf / FP64_VECTOR_PEAK_TFLOPS, 'algorithmic_flop': steps * FLOP_PER_STEP, 'flop_per_ray_step': FLOP_PER_STEP,
         'traffic': traffic, 'traffic_source': traffic_src,
         'secondary': {'bound': 'hbm', 'achieved': gb, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': gb / HBM_PEAK_GBS,
                       'algorithmic_bytes': steps * BYTES_PER_STEP, 'bytes_per_ray_step': BYTES_PER_STEP},
         'launch_ms': ms}
    if note:
        r['note'] = note
    return r


def one_step_fused(rt, torch, env, source, angles_dev, ev=None, device=None):
    """The same product, trajectories never stored: count pass (per-ray scalars, plot range) -> fused cast+gain+binning
    for the rays it can bin exactly, stored path for the others -> normalise.  This is Simulation2D.cast(store=False)
    + heatmap() spelled out so that the stages can be timed."""
    from ray_tracer_amd import _lib, parallel
    L = _lib.lib()
    f64 = dict(dtype=torch.float64, device=device)
    if ev: ev[0].record()
    fan = rt.RayFan(env, source, angles_dev, store=False, device=device, **PROP)
    if ev: ev[1].record()
    okm = fan.status == 0
    big = torch.finfo(torch.float64).max
    rng = fan.range
    lo = torch.minimum(torch.where(okm, rng[:2], torch.full_like(rng[:2], big)).amin(dim=1), torch.zeros(2, **f64))
    hi = torch.maximum(torch.where(okm, rng[2:], torch.full_like(rng[2:], -big)).amax(dim=1), torch.zeros(2, **f64))
    shape = parallel.heatmap_shape(lo.cpu().numpy(), hi.cpu().numpy(), RES)
    H = torch.zeros(int(shape[0]) * int(shape[1]), **f64)
    fan.heatmap_accumulate_fused(H, RES, shape, freq=1)
    if ev: ev[2].record()
    outp = torch.empty(int(shape[0]) * int(shape[1]), **f64)
    scratch = torch.empty(2, **f64)
    shape_c = (C.c_int32 * 2)(int(shape[0]), int(shape[1]))
    _lib.check(L.rt2d_heatmap_finalize(H.data_ptr(), shape_c, CUTOFF, .5, outp.data_ptr(), scratch.data_ptr(),
                                       torch.cuda.current_stream().cuda_stream))
    if ev: ev[3].record()
    return fan, outp.view(int(shape[1]), int(shape[0]))


def ray_steps_of(n_pts, status):
    ok = status == 0
    return int((n_pts[ok] - 1).sum().item())


def lane_utilisation(n_pts):
    """sum(n_pts) / sum over 64-ray tiles of 64 * max(n_pts): the share of a wave's lane-steps that belong to a ray
    which is still alive (a wave lives until its longest ray ends)."""
    import torch
    n = int(n_pts.numel())
    if n == 0:
        return None
    p = torch.cat([n_pts, n_pts.new_zeros((-n) % 64)]).view(-1, 64).to(torch.int64)
    return float(p.sum().item()) / float(64 * p.amax(dim=1).sum().item())


def default_resolution_block(rt, torch, env_c2, device, rays, reps=2):
    """Fans at the reference's DEFAULT step sizes (preferences.py:6-9: dx_max 10, dz_max 0.1, 100 000 steps, no rebound
    limit).  (a) the 10-ray fan of tests_3.ipynb cells 2-3, 14, 16 -- the reference's second published timing, 2 m 10.7 s --
    pinned to the reference by tests/golden/t3_default10.npz; (b) a C2 fan of `rays` rays through
    Simulation2D.cast(store=False) + heatmap(): the default path, and the other no-trajectory path beside it."""
    from ray_tracer_amd import rays as rays_mod
    out = {'step_sizes': 'dx_max 10, dz_max 0.1, n_steps_max 100000, n_rebounds_max -1 (preferences.py:6-9)'}

    def timed(fn, k):
        fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(k):
            r = fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / k, r

    xs = np.linspace(0., 100000., 4001)
    env3 = rt.Environment2D((xs, -1000. + 0.001 * xs), (np.array([0., 100000.]), np.array([0., 0.])),
                            np.array([0., -1000.]), np.array([100000., 0.]))
    src3, ang3 = np.array([0., -200.]), np.linspace(-0.5, 0.3, 10)

    def t3(with_map):
        sim = rt.Simulation2D(env3, src3)
        fan = sim.cast(ang3)
        if with_map:
            sim.heatmap(resolution=list(RES), as_tensor=True)
        fan.validate()
        return fan
    t_cast, fan3 = timed(lambda: t3(False), reps)
    t_map, _ = timed(lambda: t3(True), reps)
    ref_s = None
    gold = os.path.join(ROOT, 'tests', 'golden', 't3_default10.npz')
    if os.path.exists(gold):
        ref_s = float(np.load(gold)['ref_seconds'])
    out['tests3_fan10'] = {
        'workload': 'tests_3.ipynb: floor -1000 + 0.001 x over 100 km, source (0, -200), 10 rays linspace(-0.5, 0.3), defaults',
        'ray_steps': fan3.ray_steps(), 'ms_per_cast': t_cast * 1e3, 'ms_per_cast_plus_heatmap': t_map * 1e3,
        'passes': fan3.passes, 'lane_utilisation': lane_utilisation(fan3.n_pts),
        'reference_s_published': 130.7, 'reference_s_published_source': 'tests_3.ipynb cell 16 (author\'s machine, 1 core)',
        'reference_s_build_container': ref_s, 'reference_s_build_container_source': 'oracle/gen_golden.py t3 (ref_seconds in the fixture)',
        'note': '10 rays = one wave of 10 lanes: 100 000 dependent steps of a lone wave (latency, not throughput)'}

    n = int(rays)
    angles = torch.from_numpy(-(np.arange(n) + 1) * (0.999 / n)).to(device)
    source = np.array([0., 0.])

    def run(path):
        old = rays_mod.CHUNKED_MIN_RAYS
        if path == 'fused':
            rays_mod.CHUNKED_MIN_RAYS = 1 << 60
        elif path == 'chunked':
            rays_mod.CHUNKED_MIN_RAYS = 1
        try:
            sim = rt.Simulation2D(env_c2, source)
            fan = sim.cast(angles, store=False)
            hm = sim.heatmap(resolution=list(RES), cutoff=CUTOFF, reduction_power=.5, as_tensor=True)
            return fan, hm
        finally:
            rays_mod.CHUNKED_MIN_RAYS = old
    t_def, (fan, hm) = timed(lambda: run('default'), reps)
    steps = ray_steps_of(fan.n_pts, fan.status)
    default_is_fused = n < rays_mod.CHUNKED_MIN_RAYS
    other = 'chunked' if default_is_fused else 'fused'
    del fan3
    torch.cuda.empty_cache()          # (the chunked path wants a third of the FREE memory per chunk: give back what the blocks above cached)
    t_other, (fan_o, hm_o) = timed(lambda: run(other), 1)
    blk = {
        'workload': f'C2 environment, {n} rays -(j+1) 0.999/{n} rad, defaults, Simulation2D.cast(store=False) + heatmap()',
        'rays': n, 'ray_steps': steps, 's_per_heatmap': t_def, 'ray_steps_per_s': steps / t_def, 'passes': fan.passes,
        'path': ('fused kernel (rt2d_cast_heatmap: count pass, then cast -> gain -> binning with nothing stored; the rays it '
                 'cannot bin go through the stored path beside it)' if default_is_fused else
                 'chunked (cast with trajectories -> one-pass gain + bin -> drop)'),
        'why': 'a ray has 100 001 points at these step sizes: trajectories are 1.6 MB per ray and every point is written and '
               'read once (chunked), against no trajectory traffic at all (fused); measured both ways below',
        'chunks': 0 if default_is_fused else -(-n // fan._chunk_rays()),
        'rays_through_the_stored_side_path': int(fan.last_slow_rays),
        'lane_utilisation': lane_utilisation(fan.n_pts),
        'fp64_vector_frac_at_48_flop_per_step': steps * FLOP_PER_STEP / t_def / 1e12 / FP64_VECTOR_PEAK_TFLOPS,
        'stop_code_histogram': torch.bincount(fan.stop_code.to(torch.int64), minlength=9).tolist(),
        'other_path': {'path': other, 's_per_heatmap': t_other, 'chunks': -(-n // fan_o._chunk_rays()) if other == 'chunked' else 0,
                       'max_abs_diff_of_the_maps': float((hm - hm_o).abs().max().item())}}
    out['c2_fan'] = blk
    return out


def cpu_baseline(seconds=12.0):
    """oracle/rt2d_oracle.c (scalar port of the reference's algorithm) on this host, 1 core, same C2 fan."""
    from oracle import oracle as orc
    xs = np.linspace(0., 75000., 3001)
    oenv = orc.OracleEnv(orc.LinearFloor(xs, c2_floor(xs)), orc.LinearFloor([0., 75000.], [0., 0.]),
                         [0., -1501.], [75000., 1.])
    angles = np.array([i / 1000 for i in range(-1, -1000, -1)])
    p = orc.make_params(**PROP)
    orc.run_fan(oenv, [0., 0.], angles[:50], p)       # warm
    reps, steps, t0 = 0, 0, time.perf_counter()
    while True:
        s, H, nerr = orc.run_fan(oenv, [0., 0.], angles, p, freq=1, res=RES)
        steps += s; reps += 1
        dt = time.perf_counter() - t0
        if dt >= seconds:
            break
    # run_fan makes two propagation passes (range, then binning), like the GPU's count + store passes
    res = {'value': steps / dt, 'unit': 'ray-steps/s', 'cores': 1, 'kind': 'port',
           'sample': f'{reps} x the full 999-ray C2 fan (cast x2 + post + gain@1Hz + heat map) in {dt:.2f} s, '
                     f'{s} ray-steps per fan', 's_per_heatmap': dt / reps}
    # the same port on all the cores this job may use: the fan cut into one block of rays per thread (ctypes releases
    # the GIL); every block bins into its own map, the final sum of the maps is not included
    from concurrent.futures import ThreadPoolExecutor
    threads = max(1, min(16, int(os.environ.get('OMP_NUM_THREADS', '0')) or (os.cpu_count() or 1)))
    chunks = [angles[k::threads] for k in range(threads)]
    with ThreadPoolExecutor(threads) as pool:
        reps_t, t0 = 0, time.perf_counter()
        while True:
            list(pool.map(lambda a: orc.run_fan(oenv, [0., 0.], a, p, freq=1, res=RES), chunks))
            reps_t += 1
            dtt = time.perf_counter() - t0
            if dtt >= seconds / 3:
                break
    # the 10-ray fan of tests_3.ipynb at the reference's default step sizes (bench block default_resolution), same port, 1 core
    xs3 = np.linspace(0., 100000., 4001)
    oenv3 = orc.OracleEnv(orc.LinearFloor(xs3, -1000. + 0.001 * xs3), orc.LinearFloor([0., 100000.], [0., 0.]),
                          [0., -1000.], [100000., 0.])
    p3 = orc.make_params()
    t0 = time.perf_counter()
    s3, _, _ = orc.run_fan(oenv3, [0., -200.], np.linspace(-0.5, 0.3, 10), p3, freq=1, res=RES)
    res['tests3_fan10_default_steps'] = {'s_per_heatmap': time.perf_counter() - t0, 'ray_steps': int(s3), 'cores': 1,
                                         'sample': 'the whole fan once (cast x2 + post + gain@1Hz + heat map)'}
    res['all_cores'] = {'value': s * reps_t / dtt, 'unit': 'ray-steps/s', 'cores': threads, 's_per_heatmap': dtt / reps_t,
                        'sample': f'{reps_t} x the same fan, {threads} threads, one block of rays each, in {dtt:.2f} s'}
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--large-rays', type=int, default=1 << 20, help='rays of the large-fan block (0 = skip)')
    ap.add_argument('--large-steps', type=int, default=3)
    ap.add_argument('--default-rays', type=int, default=65536, help='rays of the default-step-size C2 fan (0 = skip the block)')
    ap.add_argument('--no-cpu', action='store_true')
    ap.add_argument('--profile-pass', action='store_true',
                    help='tools/profile_gpu.sh: skip the blocks whose launches share kernel name AND grid size with the C2 fans '
                         '(the bathymetry fan) or only add long latency-bound launches (default step sizes, sweep + refine), '
                         'so that per-(kernel, grid) averages of the profile are of one workload')
    ap.add_argument('--force-collectives', action='store_true',
                    help='run the two all-reduces of the step even in a group of one rank (rehearses RCCL on a 1-GPU box)')
    ap.add_argument('--dry-run', action='store_true', help='ranks print their torch.distributed environment and exit (launcher test)')
    args = ap.parse_args()

    if args.dry_run and 'WORLD_SIZE' in os.environ:
        print(json.dumps({'rank': int(os.environ['RANK']), 'local_rank': int(os.environ['LOCAL_RANK']),
                          'world': int(os.environ['WORLD_SIZE']), 'port': os.environ['MASTER_PORT']}), flush=True)
        sys.exit(int(os.environ.get('RT2D_BENCH_DRY_FAIL_RANK', '-1')) == int(os.environ['RANK']))
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        # plain `python bench.py --gpus N`: this process becomes the launcher of N ranks and never touches the GPU
        sys.exit(launch_ranks(args.gpus))

    # ONE line on stdout, the JSON: libraries write there too (gloo announces its connections on stdout, from C++), so the
    # descriptor itself is pointed at stderr for the length of the run and the JSON line goes to the saved one at the end
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world != args.gpus and rank == 0:
        print(f'bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: the launcher decides, running {world} ranks', file=sys.stderr)
    n_dev = torch.cuda.device_count()                       # (does not initialise the GPU)
    assert n_dev > 0, 'bench.py needs a GPU'
    dev_index = local_rank % n_dev                          # one rank per GPU; wraps only when rehearsed on fewer GPUs
    torch.cuda.set_device(dev_index)
    device = torch.device('cuda', dev_index)
    backend = None
    if world > 1 or args.force_collectives:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29533')
        # "nccl" is RCCL on ROCm.  RCCL wants one GPU per rank: with more ranks than GPUs (a rehearsal on a 1-GPU box) the
        # collectives go over gloo instead, and the JSON line says so
        backend = os.environ.get('RT2D_DIST_BACKEND', 'nccl' if world <= n_dev else 'gloo')
        if backend == 'nccl':
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
        assert dist.get_world_size() == world
    collectives = world > 1 or args.force_collectives

    import rt2d_loader
    rt = rt2d_loader.load()
    env = make_env(rt)
    source = np.array([0., 0.])
    n_fan = 999
    j = np.arange(rank, n_fan * world, world)                       # ray j of the N-times denser fan on rank j % N
    angles = -(j + 1) / (1000.0 * world)
    angles_dev = torch.from_numpy(angles).to(device)
    env.device_handle(device.index)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # The timed step goes through the product classes (Simulation2D.cast + .heatmap): every rank hands the WHOLE fan of
    # 999 N rays to cast(), which keeps this rank's share (ray j on rank j % N); heatmap() combines the ranks (C1, C2).
    all_angles = -(np.arange(n_fan * world) + 1) / (1000.0 * world)
    use_dist = world > 1 or args.force_collectives
    if args.force_collectives and world == 1:
        os.environ['RT2D_FORCE_COLLECTIVES'] = '1'
    for _ in range(args.warmup):
        product_step(rt, env, source, all_angles, distributed=use_dist)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        fan_p, hm = product_step(rt, env, source, all_angles, distributed=use_dist)
    barrier()
    dt = time.perf_counter() - t0
    n_pts, status = fan_p.n_pts, fan_p.status
    steps_rank = ray_steps_of(n_pts, status)
    # the same step spelled out call by call on the C ABI, for the per-stage HIP-event times (not the headline)
    for _ in range(min(args.warmup, 2)):
        one_step(rt, torch, env, source, angles_dev, collectives=collectives, device=device)
    stages = Stages(torch)
    barrier()
    t0r = time.perf_counter()
    for _ in range(args.steps):
        _, n_pts_r, status_r, hm_r = one_step(rt, torch, env, source, angles_dev, stages=stages, collectives=collectives, device=device)
    barrier()
    dt_raw = time.perf_counter() - t0r
    assert ray_steps_of(n_pts_r, status_r) == steps_rank and tuple(hm_r.shape) == tuple(hm.shape)
    t = torch.tensor([dt, float(steps_rank)], dtype=torch.float64, device=device)
    if world > 1:
        from ray_tracer_amd import parallel
        tmax = parallel._all_reduce(t.clone(), dist.ReduceOp.MAX)
        tsum = parallel._all_reduce(t.clone(), dist.ReduceOp.SUM)
        dt, steps_all = float(tmax[0].item()), float(tsum[1].item())
    else:
        steps_all = float(steps_rank)
    st = stages.summary()

    result = None
    if rank == 0:
        ms_per_step = dt / args.steps * 1e3
        value = steps_all * args.steps / dt
        store_ms = st['cast_store']
        traffic, traffic_src = pmc_traffic(large=False)
        result = {
            'metric': 'ray-steps/sec (node) + s per 999-ray 20-rebound heatmap', 'value': value, 'unit': 'ray-steps/s',
            'n_gpus': world, 'ranks': dist.get_world_size() if backend else 1,
            'collective_backend': None if not backend else ('nccl (RCCL)' if backend == 'nccl' else backend), 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': ms_per_step,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
            'config': {'workload': 'c2_fan999: 999 rays/GPU (-(j+1)/(1000 N) rad), 20 rebounds, dx_max 50, dz_max 25, '
                                   'floor max(-e^{0.0005x}-500,-1500+0.02x) over 75 km, gain @1 Hz, heat map 50x25 m, cutoff 0.5, reduction power 0.5',
                       'rays_per_gpu': n_fan, 'ray_steps_per_gpu': steps_rank, 'heatmap_shape': list(hm.shape)},
            's_per_heatmap': dt / args.steps,
            'timed_path': 'product API: Simulation2D.cast + Simulation2D.heatmap (one device -> host read per heat map)',
            'product_api_ms': ms_per_step, 'raw_abi_ms': dt_raw / args.steps * 1e3,
            'stage_ms': st,
            'roofline': roofline_block('rt2d_cast_kernel<WRITE=true>', steps_rank, store_ms, traffic, traffic_src,
                                       note='999 rays = 250 wavefronts of 4 rays, one per CU: bound by the dependent-instruction '
                                            'latency of a lone wave (DESIGN.md section 6); see large_fan for the full chip'),
        }

    # N > 1: the 1M-ray fan of the single-GPU block below, SHARDED over the ranks (strong scaling: north_star's "1M-ray
    # heat map on 8 GPUs").  Rays are dealt in blocks of 64 (one tile) round-robin, so every rank sees the whole angular
    # range; C1 / C2 as in the headline step.  Timed like the headline: barrier, K steps, barrier, max over ranks.
    # N > 1: the literal BASELINE metric -- the 999-ray 20-rebound fan ITSELF dealt over the N ranks (strong scaling: ray j on
    # rank j % N), C1 and C2 included, through the product API; timed like the headline (barrier, K steps, barrier, max
    # over ranks).  999 rays are latency-bound on one GPU already: more GPUs shorten nothing but the longest wave's queue.
    if world > 1:
        angles999 = -(np.arange(n_fan) + 1) / 1000.0
        for _ in range(max(1, min(args.warmup, 3))):
            product_step(rt, env, source, angles999, distributed=True)
        barrier(); t0 = time.perf_counter()
        for _ in range(args.steps):
            fan_s9, hm_s9 = product_step(rt, env, source, angles999, distributed=True)
        barrier(); dt9 = time.perf_counter() - t0
        from ray_tracer_amd import parallel
        t9 = torch.tensor([dt9, float(ray_steps_of(fan_s9.n_pts, fan_s9.status))], dtype=torch.float64, device=device)
        t9_max = parallel._all_reduce(t9.clone(), dist.ReduceOp.MAX)
        t9_sum = parallel._all_reduce(t9.clone(), dist.ReduceOp.SUM)
        if rank == 0:
            dt9 = float(t9_max[0].item())
            result['fan999_strong'] = {
                'workload': 'c2_fan999 (999 rays, 20 rebounds) dealt over the ranks, ray j on rank j % N; C1 + C2 included',
                'rays_total': n_fan, 'rays_rank0': int(fan_s9.n), 'scaling': 'strong', 'ray_steps_total': float(t9_sum[1].item()),
                's_per_heatmap': dt9 / args.steps, 'value': float(t9_sum[1].item()) * args.steps / dt9, 'unit': 'ray-steps/s',
                'heatmap_shape': list(hm_s9.shape)}
        del fan_s9, hm_s9

    if args.large_rays > 0 and world > 1:
        nL = args.large_rays
        tiles = np.arange(rank, (nL + 63) // 64, world)
        jj = (tiles[:, None] * 64 + np.arange(64)[None, :]).reshape(-1)
        jj = jj[jj < nL]
        shard_angles = torch.from_numpy(-(jj + 1) * (0.999 / nL)).to(device)
        one_step(rt, torch, env, source, shard_angles, collectives=True, device=device)
        stages_s = Stages(torch)
        barrier(); t0 = time.perf_counter()
        for _ in range(args.large_steps):
            _, n_pts_s, status_s, hm_s = one_step(rt, torch, env, source, shard_angles, stages=stages_s, collectives=True, device=device)
        barrier(); dts = time.perf_counter() - t0
        from ray_tracer_amd import parallel
        ts = torch.tensor([dts, float(ray_steps_of(n_pts_s, status_s))], dtype=torch.float64, device=device)
        ts_max = parallel._all_reduce(ts.clone(), dist.ReduceOp.MAX)
        ts_sum = parallel._all_reduce(ts.clone(), dist.ReduceOp.SUM)
        if rank == 0:
            dts = float(ts_max[0].item())
            result['large_fan_sharded'] = {
                'rays_total': nL, 'rays_per_gpu': int(len(jj)), 'scaling': 'strong', 'ray_steps_total': float(ts_sum[1].item()),
                'value': float(ts_sum[1].item()) * args.large_steps / dts, 'unit': 'ray-steps/s',
                's_per_heatmap': dts / args.large_steps, 'stage_ms_rank0': stages_s.summary(), 'heatmap_shape': list(hm_s.shape),
                'lane_utilisation_rank0': lane_utilisation(n_pts_s),
                'roofline_rank0': roofline_block('rt2d_cast_kernel<WRITE=true>', ray_steps_of(n_pts_s, status_s),
                                                 stages_s.summary()['cast_store'], None,
                                                 'no PMC pass of a sharded run is committed (the 1-GPU profile is of the unsharded fan)',
                                                 note='rank 0\'s storing cast of its share of the fan')}
        del shard_angles

    # the same pipeline where the GPU is full (single-GPU block, rank 0 only prints it)
    if args.large_rays > 0 and world == 1:
        nL = args.large_rays
        big_angles = torch.from_numpy(-(np.arange(nL) + 1) * (0.999 / nL)).to(device)
        one_step(rt, torch, env, source, big_angles, device=device)
        stages_l = Stages(torch)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(args.large_steps):
            _, n_pts_l, status_l, hm_l = one_step(rt, torch, env, source, big_angles, stages=stages_l, device=device)
        torch.cuda.synchronize(); dtl = time.perf_counter() - t0
        sl = ray_steps_of(n_pts_l, status_l)
        stl = stages_l.summary()
        traffic_l, traffic_src_l = pmc_traffic(large=True) if nL == (1 << 20) else (None, 'the committed profile is of the 1 048 576-ray fan')
        result['large_fan'] = {
            'rays': nL, 'ray_steps': sl, 'value': sl * args.large_steps / dtl, 'unit': 'ray-steps/s',
            's_per_heatmap': dtl / args.large_steps, 'stage_ms': stl, 'lane_utilisation': lane_utilisation(n_pts_l),
            'roofline': roofline_block('rt2d_cast_kernel<WRITE=true>', sl, stl['cast_store'], traffic_l, traffic_src_l),
        }
        if nL == (1 << 20):
            pb = pmc_binning()
            t_bin = stl['heatmap'] * 1e-3
            result['large_fan']['binning'] = {
                'kernels': 'rt2d_heatmap_fast_kernel (tile) + rt2d_heatmap_kernel (general)', 'ms': stl['heatmap'],
                'algorithmic_bytes': sl * 24,                    # x, z, gain of every point read once
                'achieved_GBps_algorithmic': sl * 24 / t_bin / 1e9, 'frac_of_hbm_peak': sl * 24 / t_bin / 1e9 / HBM_PEAK_GBS,
                'hbm_bytes_pmc': pb['hbm_bytes'], 'l2_atomics_pmc': pb['atomics'],
                'atomics_per_s': None if pb['atomics'] is None else pb['atomics'] / t_bin,
                'points_per_atomic': None if not pb['atomics'] else sl / pb['atomics'], 'pmc_source': pb['source']}
        # the stored fan through the product API (its default for a fan this big: the one-pass map, queued without waiting)
        # (warm-up: TWO fans alive at once, as in the loop below, where the previous fan is only released when the next one
        # has been built -- otherwise the loop's first step may pay for a fresh 68 GB of address space from the driver)
        warm_a = product_step(rt, env, source, big_angles)
        warm_b = product_step(rt, env, source, big_angles)
        del warm_a, warm_b
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(args.large_steps):
            fan_p2, hm_p2 = product_step(rt, env, source, big_angles)
        torch.cuda.synchronize(); dtp = time.perf_counter() - t0
        result['large_fan']['product_api'] = {
            's_per_heatmap': dtp / args.large_steps, 'path': 'Simulation2D.cast + heatmap(): storing cast, one-pass gain + binning '
            '(rt2d_gain_heatmap_rest), nothing read back before the end', 'passes': fan_p2.passes,
            'max_abs_diff_vs_three_kernel_map': float((hm_p2 - hm_l).abs().max().item())}
        # the block's own figure is the product API's, like the headline's; the raw-ABI loop of three kernels (cast, gain,
        # binning: the path of fans below 2^17 rays and the source of the per-stage times) is kept beside it
        result['large_fan']['raw_abi_three_kernels'] = {'s_per_heatmap': result['large_fan']['s_per_heatmap'],
                                                        'value': result['large_fan']['value'], 'unit': 'ray-steps/s'}
        result['large_fan']['s_per_heatmap'] = dtp / args.large_steps
        result['large_fan']['value'] = sl * args.large_steps / dtp
        result['large_fan']['timed_path'] = 'product API (see product_api.path); stage_ms and roofline.launch_ms come from the raw-ABI loop'
        del fan_p2, hm_p2
        # The same product WITHOUT keeping trajectories, through the product API: Simulation2D.cast(store=False) is lazy for a
        # fan this big, and heatmap() then casts it chunk by chunk WITH trajectories, bins each chunk through the stored
        # pipeline and drops it (RayFan.heatmap_accumulate_chunked): no counting pass, per-ray results as a by-product,
        # never more than a chunk's trajectories in HBM.
        from ray_tracer_amd import rays as rays_mod

        def frugal_step():
            sim_f = rt.Simulation2D(env, source)
            fan_f = sim_f.cast(big_angles, store=False, **PROP)
            hm = sim_f.heatmap(resolution=list(RES), cutoff=CUTOFF, reduction_power=.5, as_tensor=True)
            return fan_f, hm
        frugal_step()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(args.large_steps):
            fan_f, hm_f = frugal_step()
        torch.cuda.synchronize(); dtf = time.perf_counter() - t0
        sf = ray_steps_of(fan_f.n_pts, fan_f.status)
        result['large_fan_fused'] = {
            'rays': nL, 'ray_steps': sf, 'value': sf * args.large_steps / dtf, 'unit': 'ray-steps/s',
            's_per_heatmap': dtf / args.large_steps,
            'path': 'Simulation2D.cast(store=False) + heatmap(): lazy fan, cast-with-trajectories -> one-pass gain+bin -> drop, in chunks '
                    'of as many rays as a third of the free HBM holds (no counting pass, no gain array)',
            'chunk_rays': fan_f._chunk_rays(), 'passes': fan_f.passes,
            'rays_binned_by_the_general_kernel': int(fan_f.last_slow_rays),
            'max_abs_diff_vs_stored_map': float((hm_f - hm_l).abs().max().item())}
        del fan_f, hm_f
        # ... and the fused KERNEL (rt2d_cast_heatmap: cast -> gain -> binning in registers, nothing stored at all) behind a
        # counting pass, which the chunked path replaced as the default for big fans
        _, hm_k = one_step_fused(rt, torch, env, source, big_angles, device=device)
        recs = []
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(args.large_steps):
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
            fan_k, hm_k = one_step_fused(rt, torch, env, source, big_angles, ev=ev, device=device)
            recs.append(ev)
        torch.cuda.synchronize(); dtk = time.perf_counter() - t0
        names = ('cast_count', 'fused_cast_gain_bin+rest', 'finalize')
        stf = {nm: float(np.mean([e[k].elapsed_time(e[k + 1]) for e in recs])) for k, nm in enumerate(names)}
        sk = ray_steps_of(fan_k.n_pts, fan_k.status)
        result['large_fan_fused']['fused_kernel'] = {
            's_per_heatmap': dtk / args.large_steps, 'stage_ms': stf,
            'rays_binned_by_the_stored_path': int(fan_k.last_slow_rays),
            'fused_violations': int(fan_k.last_fused_violations.item()),
            'max_abs_diff_vs_stored_map': float((hm_k - hm_l).abs().max().item()),
            'fp64_valu_frac': sk * (FLOP_PER_STEP + 30) / (stf['fused_cast_gain_bin+rest'] * 1e-3) / 1e12 / FP64_VECTOR_PEAK_TFLOPS,
            # the counting pass stores nothing: stepping alone, the FP64 vector figure of the cast kernel
            'count_pass_steps_per_s': sk / (stf['cast_count'] * 1e-3),
            'count_pass_fp64_valu_frac': sk * FLOP_PER_STEP / (stf['cast_count'] * 1e-3) / 1e12 / FP64_VECTOR_PEAK_TFLOPS}
        del fan_k, hm_k
        # the same fan over a bathymetry cut (BASELINE configs 3 / 5: the floor is a table cut out of an ESRI grid;
        # the reference's data/asc file is absent, so the synthetic grid of the parity fixtures stands in)
        asc = os.path.join(ROOT, 'tests', 'golden', 'synth_bathy.asc')
        if os.path.exists(asc) and not args.profile_pass:
            bmap = rt.Map.from_asc(asc)
            b0, b1 = (3.05, 41.7), (3.9, 41.75)
            length = float(bmap.cut_length(b0, b1))
            env_b = rt.Environment2D(bmap.floor_table(b0, b1, npoints=751), (np.array([0., length]), np.array([0., 0.])),
                                     np.array([0., -2501.]), np.array([length, 1.]))
            env_b.device_handle(device.index)
            src_b = np.array([0., -50.])
            one_step(rt, torch, env_b, src_b, big_angles, device=device)
            stages_b = Stages(torch)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(args.large_steps):
                _, n_pts_b, status_b, hm_b = one_step(rt, torch, env_b, src_b, big_angles, stages=stages_b, device=device)
            torch.cuda.synchronize(); dtb = time.perf_counter() - t0
            sb = ray_steps_of(n_pts_b, status_b)
            product_step(rt, env_b, src_b, big_angles)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(args.large_steps):
                fan_pb, hm_pb = product_step(rt, env_b, src_b, big_angles)
            torch.cuda.synchronize(); dtpb = time.perf_counter() - t0
            diff_b = float((hm_pb - hm_b).abs().max().item()) if tuple(hm_pb.shape) == tuple(hm_b.shape) else None
            del fan_pb, hm_pb
            result['large_fan_bathymetry'] = {
                'rays': nL, 'floor': 'Map.from_asc(tests/golden/synth_bathy.asc).floor_table((3.05, 41.7), (3.9, 41.75), 751): '
                                     f'{length / 1e3:.1f} km cut, depths {float(env_b.floor(np.linspace(0, length, 751)).max()):.0f} '
                                     f'to {float(env_b.floor(np.linspace(0, length, 751)).min()):.0f} m',
                'ray_steps': sb, 'rays_with_error_status': int((status_b != 0).sum().item()),
                'lane_utilisation': lane_utilisation(n_pts_b),
                'value': sb * args.large_steps / dtpb, 'unit': 'ray-steps/s', 's_per_heatmap': dtpb / args.large_steps,
                'timed_path': 'product API: Simulation2D.cast + heatmap() (storing cast, one-pass gain + binning)',
                'max_abs_diff_vs_three_kernel_map': diff_b,
                'raw_abi_three_kernels': {'s_per_heatmap': dtb / args.large_steps, 'value': sb * args.large_steps / dtb, 'unit': 'ray-steps/s'},
                'stage_ms': stages_b.summary(), 'heatmap_shape': list(hm_b.shape)}
        del big_angles
        # BASELINE config 5 through the product API: EigenraySim2D sweep of --large-rays rays (scalars only), the 1000 best
        # kept (device top-k), three refine iterations of 5 rays per cone (cast without a rebound limit, as the reference does)
        def eigen():
            torch.cuda.synchronize(); t0 = time.perf_counter()
            sim_e = rt.EigenraySim2D(env, source, np.array([40000., -600.]), 1000, 20, n_rays_scan=nL, scan_angle_min=-0.999,
                                     scan_angle_max=-0.001, dx_max=50., dz_max=25.)
            torch.cuda.synchronize(); t1 = time.perf_counter()
            sim_e.refine(iterations=3, n_rays=5, force_refine=True)
            torch.cuda.synchronize(); t2 = time.perf_counter()
            return sim_e, t1 - t0, t2 - t1
        if not args.profile_pass:
            eigen()
        sim_e, t_sweep, t_refine = eigen()
        cone = sim_e.last_cone_fan
        result['config5_sweep_refine'] = {
            'rays_scanned': nL, 'rays_kept': 1000, 'target': [40000., -600.],
            'sweep_plus_topk_plus_recast_ms': t_sweep * 1e3, 'refine_x3_ms': t_refine * 1e3,
            'sweep_ray_steps': ray_steps_of(sim_e.scan_fan.n_pts, sim_e.scan_fan.status),
            'sweep_lane_utilisation': lane_utilisation(sim_e.scan_fan.n_pts),
            'last_refine_cones': {'rays': int(cone.n), 'ray_steps': ray_steps_of(cone.n_pts, cone.status),
                                  'lane_utilisation': lane_utilisation(cone.n_pts),
                                  'longest_ray_points': int(cone.n_pts.max().item())}}
        del sim_e, cone
        if args.default_rays > 0 and not args.profile_pass:
            result['default_resolution'] = default_resolution_block(rt, torch, env, device, args.default_rays)
    if rank == 0 and world == 1:
        # "raypower cutoff=0.5" read the other way (SURVEY trap 2): rays END when their power falls below the cutoff --
        # not in the reference's propagate; the opt-in CUTOFF instance of the cast kernel, reported apart from the headline
        fans = []
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10):
            fans.append(rt.RayFan(env, source, angles, power_cutoff=.5, cutoff_freq=1., **PROP))
        torch.cuda.synchronize(); dtc = time.perf_counter() - t0
        result['power_cutoff_termination'] = {
            'power_cutoff': .5, 'freq_hz': 1., 'rays': n_fan, 'ms_per_stored_cast': dtc / 10 * 1e3,
            'ray_steps': fans[-1].ray_steps(), 'rays_ended_by_the_cutoff': int((fans[-1].stop_code == 8).sum().item())}
        del fans
    if rank == 0:
        if world == 1 and not args.no_cpu:
            result['cpu_baseline'] = cpu_baseline()
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(result) + '\n').encode())
    if backend:
        dist.destroy_process_group()


def launch_ranks(n):
    """`python bench.py --gpus N` without a launcher around it: start N copies of this script, one rank per GPU, with
    the torch.distributed environment set (what `python -m torch.distributed.run --nproc-per-node N` would do), BEFORE
    anything in this process has touched the GPU.  Rank 0 prints the JSON line on the inherited stdout.  Returns the
    exit status: non-zero as soon as any rank fails (the others are then ended by PID)."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        e = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                 MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
        e.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=e))
    rc = 0
    alive = list(procs)
    while alive:
        time.sleep(0.05)
        for p in list(alive):
            code = p.poll()
            if code is None:
                continue
            alive.remove(p)
            if code != 0 and rc == 0:
                rc = code if code > 0 else 1
                for q in alive:          # a rank died: the others would wait in a collective for ever
                    q.terminate()
    return rc


if __name__ == '__main__':
    main()
