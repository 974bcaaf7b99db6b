"""Generate tests/golden/*.npz by running the REFERENCE ITSELF (imported from /root/reference).

Runs only in the build container (the reference does not travel to the GPU box); the fixtures it
writes are data only: inputs (environment tables, fan definition, parameters) and the reference's
outputs (per-ray integers, sub-sampled trajectories, end-of-ray scalars, gains, heat maps).

    python oracle/gen_golden.py [c1 c2 c3 c4 c5 t2 t3 f5]      (default: all)

Floors / ceilings are handed to the reference as oracle.LinearFloor callables, i.e. the very
piecewise-linear tables the device reads, so both sides integrate over one sea floor
(SURVEY.md section 7, step 3).
"""

import os
import sys
import time

os.environ.setdefault('MPLBACKEND', 'Agg')
_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)
REF = '/root/reference'
sys.path.insert(0, REF)
sys.path.insert(0, _ROOT)

import numpy as np                                      # noqa: E402

from environment import Environment2D                   # noqa: E402  (reference)
from ray import Ray2D, RayPack2D                        # noqa: E402  (reference)
from simulation import EigenraySim2D                    # noqa: E402  (reference)
from functions import coords_to_mask_2d                 # noqa: E402  (reference)
from oracle.oracle import LinearFloor                   # noqa: E402

GOLD = os.path.join(_ROOT, 'tests', 'golden')
STOP = {None: 0, 'exit-xmin': 1, 'exit-xmax': 2, 'exit-zmin': 3, 'exit-zmax': 4, 'max-rebounds': 5,
        'backprop': 6, 'max-iter': 7}
N_SAMPLES = 48


def tabulated(fn, x0, x1, n):
    xs = np.linspace(x0, x1, n)
    return LinearFloor(xs, np.asarray(fn(xs), dtype=float) * np.ones_like(xs))


def run_fan(env, source, angles, freqs, target=None, heatmap=True, res=(50, 25), **prop):
    """Cast the fan ray by ray exactly as Simulation2D.cast does (simulation.py:89-94), catching the
    per-ray ValueError of SURVEY trap 6; returns a dict of arrays."""
    n = len(angles)
    reb_cap = 64
    out = dict(
        steps=np.zeros(n, np.int32), stop_code=np.zeros(n, np.int32), n_rebounds=np.zeros(n, np.int32),
        error=np.zeros(n, np.int32), range_min=np.zeros((n, 2)), range_max=np.zeros((n, 2)),
        L_end=np.full(n, np.nan), T_end=np.full(n, np.nan), dist_to_target=np.full(n, np.nan),
        step_target=np.full(n, -1, np.int32), T_target=np.full(n, np.nan),
        reb_step=np.full((n, reb_cap), -1, np.int32), reb_surface=np.full((n, reb_cap), -1, np.int8),
        reb_velocity=np.full((n, reb_cap), np.nan), reb_angle=np.full((n, reb_cap), np.nan),
        reb_pos=np.full((n, reb_cap, 2), np.nan),
        sample_idx=np.zeros((n, N_SAMPLES), np.int32), sample_xz=np.full((n, N_SAMPLES, 2), np.nan),
        sample_T=np.full((n, N_SAMPLES), np.nan),
        G_end=np.full((len(freqs), n), np.nan), sample_G=np.full((len(freqs), n, N_SAMPLES), np.nan),
    )
    pack = RayPack2D()
    rays = []
    t0 = time.time()
    for i, angle in enumerate(angles):
        kw = dict(prop)
        if target is not None:
            kw['target'] = np.asarray(target, dtype=float)
        ray = Ray2D(env, np.asarray(source, dtype=float), float(angle), **kw)
        try:
            ray.propagate(**kw)
        except ValueError:
            out['error'][i] = 1
        rebs = ray._Ray2D__rebounds
        npts = ray.XZ.shape[0]
        out['steps'][i] = npts
        out['stop_code'][i] = STOP[ray.stop_reason]
        out['n_rebounds'][i] = ray.n_rebounds
        for k, rb in enumerate(rebs[:reb_cap]):
            out['reb_step'][i, k] = rb['step']
            out['reb_surface'][i, k] = 0 if rb['surface'] == 'sediment' else 1
            out['reb_velocity'][i, k] = float(rb['velocity'])
            out['reb_angle'][i, k] = float(rb['angle'])
            out['reb_pos'][i, k] = rb['position']
        idx = np.unique(np.round(np.linspace(0, npts - 1, N_SAMPLES)).astype(np.int32))
        idx = np.concatenate((idx, np.full(N_SAMPLES - idx.shape[0], idx[-1], np.int32)))
        out['sample_idx'][i] = idx
        out['sample_xz'][i] = ray.XZ[idx]
        if out['error'][i]:
            rays.append(None)
            continue
        out['range_min'][i], out['range_max'][i] = ray.range_min, ray.range_max
        out['L_end'][i], out['T_end'][i] = ray.L[-1], ray.T[-1]
        out['sample_T'][i] = ray.T[idx]
        if target is not None:
            out['dist_to_target'][i] = ray.dist_to_target
            out['step_target'][i] = ray.step_target
            out['T_target'][i] = ray.T_target
        ray.populate_NEW(*freqs)                              # populate() raises without a target (SURVEY trap 5)
        for f, fr in enumerate(freqs):
            out['G_end'][f, i] = ray.G[fr][-1]
            out['sample_G'][f, i] = ray.G[fr][idx]
        pack.add(ray)
        rays.append(ray)
    out['ref_seconds'] = time.time() - t0
    out['range_min_plot'], out['range_max_plot'] = pack.range_min_plot, pack.range_max_plot

    if heatmap:
        # simulation.py:108-138 restated verbatim over the reference's own rays and its own
        # coords_to_mask_2d; vals from populate_NEW(1) because populate() raises (SURVEY trap 5).
        resa = np.array(res)
        shape = np.ceil(pack.size_plot / resa).astype(int) + 1
        full = np.zeros(shape)
        t0 = time.time()
        for ray in pack.rays:
            rx = ray.XZ.copy().astype(int)
            rx[:, 0] //= resa[0]
            rx[:, 1] //= -1 * resa[1]
            if 1 not in ray.G:
                ray.populate_NEW(1)
            vals = np.power(np.power(10, ray.G[1] / 10), -0.1)
            full += coords_to_mask_2d(shape, rx, vals)
        out['heatmap_seconds'] = time.time() - t0
        out['heatmap_res'] = resa
        out['heatmap_full'] = full
        for cutoff in (.02, .1, .5):
            norm = full / np.max(full)
            norm[norm > cutoff] = cutoff
            out[f'heatmap_plot_cut{cutoff}'] = np.power(norm, .5).T
    return out, rays


def save(name, env_spec, fan, out):
    d = {}
    d.update({f'env_{k}': v for k, v in env_spec.items()})
    d.update({f'fan_{k}': v for k, v in fan.items()})
    d.update(out)
    path = os.path.join(GOLD, name + '.npz')
    np.savez_compressed(path, **d)
    print(f'wrote {path}: {os.path.getsize(path) / 1e6:.2f} MB; ref {out.get("ref_seconds", 0):.1f} s; '
          f'total pts {int(out["steps"].sum())}; errors {int(out["error"].sum())}')


def env_spec(floor, ceil, rmin, rmax, res_x=100, res_z=10):
    return dict(floor_x=floor.xs, floor_z=floor.zs, ceil_x=ceil.xs, ceil_z=ceil.zs,
                range_min=np.asarray(rmin, float), range_max=np.asarray(rmax, float),
                res_x=np.int64(res_x), res_z=np.int64(res_z))


def make_env(spec):
    fl = LinearFloor(spec['floor_x'], spec['floor_z'])
    ce = LinearFloor(spec['ceil_x'], spec['ceil_z'])
    return Environment2D(fl, ce, spec['range_min'], spec['range_max'], res_x=int(spec['res_x']), res_z=int(spec['res_z']))


def fan_dict(source, angles, freqs, target=None, **prop):
    d = dict(source=np.asarray(source, float), angles=np.asarray(angles, float), freqs=np.asarray(freqs, float))
    d['has_target'] = np.int64(0 if target is None else 1)
    d['target'] = np.asarray(target if target is not None else [np.nan, np.nan], float)
    full = dict(dx_max=10., dz_max=0.1, n_steps_max=100000, n_rebounds_max=-1, backprop=True)
    full.update(prop)
    for k, v in full.items():
        d[k] = np.float64(v) if isinstance(v, float) else np.int64(v)
    return d


# ---- configurations ---------------------------------------------------------
def c1():
    """250-ray fan, flat floor, 10 rebounds (BASELINE config 0)."""
    fl = LinearFloor([0., 50000.], [-1000., -1000.])
    ce = LinearFloor([0., 50000.], [0., 0.])
    spec = env_spec(fl, ce, [0., -1001.], [50000., 1.])
    angles = np.linspace(-0.5, 0.5, 250)
    prop = dict(dx_max=50., dz_max=25., n_rebounds_max=10)
    out, _ = run_fan(make_env(spec), [0., -50.], angles, [1, 1000.], target=[30000., -600.], **prop)
    save('c1_flat250', spec, fan_dict([0., -50.], angles, [1, 1000.], target=[30000., -600.], **prop), out)


C2_FLOOR = lambda x: np.maximum(-1 * np.exp(0.0005 * x) - 500, -1500 + 0.02 * x)     # deprecated.ipynb cell 0


def c2_spec():
    fl = tabulated(C2_FLOOR, 0., 75000., 3001)
    ce = LinearFloor([0., 75000.], [0., 0.])
    return env_spec(fl, ce, [0., -1501.], [75000., 1.])


def c2():
    """999-ray fan, 20 rebounds (BASELINE config 1): angles i/1000, i = -1..-999 (deprecated.ipynb cell 2)."""
    spec = c2_spec()
    angles = np.array([i / 1000 for i in range(-1, -1000, -1)])
    prop = dict(dx_max=50., dz_max=25., n_rebounds_max=20)
    out, _ = run_fan(make_env(spec), [0., 0.], angles, [1, 1000.], **prop)
    save('c2_fan999', spec, fan_dict([0., 0.], angles, [1, 1000.], **prop), out)


def write_synth_asc(path, ncols=96, nrows=64, cellsize=0.01, xll=3.0, yll=41.4):
    """Synthetic ESRI ASCII grid in the format Map.from_asc expects (bathymetry.py:51-67):
    smooth seeded depth field with a NODATA land patch."""
    rng = np.random.default_rng(0)
    jj, ii = np.meshgrid(np.arange(ncols), np.arange(nrows))
    depth = -1400 + 500 * np.sin(jj / 11.0) * np.cos(ii / 7.0) + 250 * np.sin(jj / 3.7 + 1.0) + 0.5 * jj
    depth += rng.normal(scale=5.0, size=depth.shape)
    depth = np.clip(depth, -2500, -50)
    depth[:6, :8] = -99999.0
    with open(path, 'w') as f:
        f.write(f'NCOLS {ncols}\nNROWS {nrows}\nXLLCORNER {xll}\nYLLCORNER {yll}\nCELLSIZE {cellsize}\nNODATA_VALUE -99999\n')
        for row in depth:
            f.write(' '.join(f'{v:.3f}' for v in row) + '\n')


def c3():
    """Fan over a bathymetry cut (BASELINE config 2).  The real .asc is absent (SURVEY trap 3) so a synthetic
    grid is cut with the reference's own Map; x follows the reference-faithful axis x = res*t."""
    from topography.bathymetry import Map                                           # reference
    asc = os.path.join(GOLD, 'synth_bathy.asc')
    write_synth_asc(asc)
    m = Map.from_asc(asc)
    x, z = m.cut((3.05, 41.7), (3.9, 41.75), npoints=100)
    np.savez_compressed(os.path.join(GOLD, 'c3_cut.npz'), start=np.array([3.05, 41.7]), stop=np.array([3.9, 41.75]),
                        npoints=np.int64(100), x=x, z=z)
    # stretch the cut over 60 km so the fan actually sees relief (x = d*t, the corrected axis), and keep
    # the faithful axis in c3_cut.npz for the Map parity test
    d = x[-1] * (len(x) - 1)
    fl = LinearFloor(np.linspace(0., 60000., len(z)), z)
    ce = LinearFloor([0., 60000.], [0., 0.])
    zmin = float(np.floor(z.min())) - 1.
    spec = env_spec(fl, ce, [0., zmin], [60000., 1.])
    spec['cut_length_m'] = np.float64(d)
    angles = np.array([i / 1000 for i in range(-1, -1000, -1)])          # all 999 angles of config 3
    prop = dict(dx_max=50., dz_max=25., n_rebounds_max=20)
    out, _ = run_fan(make_env(spec), [0., 0.], angles, [1], **prop)
    save('c3_bathy999', spec, fan_dict([0., 0.], angles, [1], **prop), out)


def c4():
    """Multi-frequency gain (BASELINE config 3): 64 frequencies 1..20000 Hz on a 40-ray subset of the c2 fan."""
    spec = c2_spec()
    angles = np.array([i / 1000 for i in range(-5, -1000, -25)])
    freqs = np.linspace(1, 20000, 64)
    prop = dict(dx_max=50., dz_max=25., n_rebounds_max=20)
    out, _ = run_fan(make_env(spec), [0., 0.], angles, list(freqs), heatmap=False, **prop)
    save('c4_multifreq', spec, fan_dict([0., 0.], angles, freqs, **prop), out)


def c5():
    """Sweep + refine (BASELINE config 4, reduced): EigenraySim2D scan of 200 rays, keep 5, refine twice."""
    fl = LinearFloor([0., 50000.], [-1000., -1000.])
    ce = LinearFloor([0., 50000.], [0., 0.])
    spec = env_spec(fl, ce, [0., -1001.], [50000., 1.])
    env = make_env(spec)
    source, target = np.array([0., -50.]), np.array([12000., -600.])
    sim = EigenraySim2D(env, source, target, 5, 4, n_rays_scan=200, scan_angle_min=-0.4, scan_angle_max=0.4,
                        dx_max=50., dz_max=25., verbose=True, verbose_depth_max=0)
    d = dict(scan_angles=np.array(sorted(sim.raypacks['__iter0-scan'].angles.keys())),
             scan_dist=np.array([sim.raypacks['__iter0-scan'].angles[a].dist_to_target
                                 for a in sorted(sim.raypacks['__iter0-scan'].angles.keys())]),
             scan_selected_angles=np.array([r.angle for r in sim.raypacks['scan'].rays]),
             scan_selected_dist=np.array([r.dist_to_target for r in sim.raypacks['scan'].rays]),
             scan_dist_avg=np.float64(sim.dist_avg))
    sim.refine(iterations=2, n_rays=5)
    for it in (1, 2):
        pk = sim.raypacks.get(f'refine-{it}')
        if pk is None:
            continue
        d[f'refine{it}_angles'] = np.array([r.angle for r in pk.rays])
        d[f'refine{it}_dist'] = np.array([r.dist_to_target for r in pk.rays])
    d['final_dist_avg'] = np.float64(sim.dist_avg)
    d['n_refines'] = np.int64(sim.n_refines)
    dd = {f'env_{k}': v for k, v in spec.items()}
    dd.update(source=source, target=target, n_rays=np.int64(5), n_rebounds_max=np.int64(4), n_rays_scan=np.int64(200), scan_angle_min=np.float64(-0.4), scan_angle_max=np.float64(0.4),
              dx_max=np.float64(50.), dz_max=np.float64(25.))
    dd.update(d)
    np.savez_compressed(os.path.join(GOLD, 'c5_eigen.npz'), **dd)
    print('wrote c5_eigen.npz', d['scan_selected_angles'], d['final_dist_avg'])


def t2():
    """tests_2.ipynb ray: the only known-answer numbers the reference itself holds
    (T_target = 6.945020272254603, Tmult[:4] = 0.693612709, 0.616751167, 0.4324197, 0.229686171)."""
    fl = tabulated(lambda x: 50 * np.sin(0.005 * x + 37) - 2000, 0., 10000., 4001)
    ce = LinearFloor([0., 10000.], [0., 0.])
    spec = env_spec(fl, ce, [0., -5001.], [10000., 1.])
    env = make_env(spec)
    freqs = list(np.linspace(1, 20000, 500)[:4])
    out, rays = run_fan(env, [0., 0.], [-1.], freqs, heatmap=False)
    ray = rays[0]
    ray.populate_NEW(*np.linspace(1, 20000, 500))
    T_target, Tmult_target = ray.gen_filter(np.array([4000., -1000.]))
    out['gen_filter_T_target'] = np.float64(T_target)
    out['gen_filter_Tmult'] = Tmult_target
    print('T_target', T_target, 'Tmult[:4]', Tmult_target[:4])
    save('t2_single', spec, fan_dict([0., 0.], [-1.], freqs), out)


def t3():
    """tests_3.ipynb cells 2-3, 14, 16: the 10-ray fan at the reference's DEFAULT step sizes (dx_max = 10, dz_max = 0.1,
    100 000 steps, no rebound limit: preferences.py:6-9), the fan behind its second published timing ("2 m 10.7 s").
    Float source (the notebook's integer source truncates the stored trajectory, SURVEY trap 7)."""
    fl = tabulated(lambda x: -1000. + 0.001 * x, 0., 100000., 4001)
    ce = LinearFloor([0., 100000.], [0., 0.])
    spec = env_spec(fl, ce, [0., -1000.], [100000., 0.])
    angles = np.linspace(-0.5, 0.3, 10)
    out, _ = run_fan(make_env(spec), [0., -200.], angles, [1, 1000.])
    save('t3_default10', spec, fan_dict([0., -200.], angles, [1, 1000.]), out)


def synth_signal(n=4096, sample_rate=8000.):
    """Deterministic test signal for the eigenray filter (the reference's WAV inputs are absent from its tree): a
    linear chirp 50 Hz -> 3 kHz plus seeded noise, Hann-tapered."""
    rng = np.random.default_rng(7)
    t = np.arange(n) / sample_rate
    chirp = np.sin(2 * np.pi * (50. * t + 0.5 * (2950. / t[-1]) * t * t))
    return sample_rate, (chirp + 0.1 * rng.normal(size=n)) * np.hanning(n)


def f5():
    """EigenraySim2D.get_filter_NEW (simulation.py:410-480) on the c5 environment: per-ray target products
    (gen_filter, ray.py:382-405) and the filtered, time-offset, summed signal."""
    fl = LinearFloor([0., 50000.], [-1000., -1000.])
    ce = LinearFloor([0., 50000.], [0., 0.])
    spec = env_spec(fl, ce, [0., -1001.], [50000., 1.])
    env = make_env(spec)
    source, target = np.array([0., -50.]), np.array([12000., -600.])
    sim = EigenraySim2D(env, source, target, 5, 4, n_rays_scan=200, scan_angle_min=-0.4, scan_angle_max=0.4,
                        dx_max=50., dz_max=25., verbose=True, verbose_depth_max=0)
    freqs = np.linspace(10., 4000., 40)
    filt = sim.get_filter_NEW(*freqs)
    sr, sig = synth_signal()
    sr_out, out = filt(sr, sig)
    pack = sim.raypacks[sim.pack_default]
    angles = np.sort(list(pack.angles.keys()))
    T_t, Tm_t = [], []
    for a in angles:
        T, Tm = pack.angles[a].gen_filter(target)
        T_t.append(T); Tm_t.append(Tm)
    dd = {f'env_{k}': v for k, v in spec.items()}
    dd.update(source=source, target=target, n_rays=np.int64(5), n_rebounds_max=np.int64(4), n_rays_scan=np.int64(200),
              scan_angle_min=np.float64(-0.4), scan_angle_max=np.float64(0.4), dx_max=np.float64(50.), dz_max=np.float64(25.),
              freqs=freqs, sample_rate=np.float64(sr), signal=sig, filtered=out, angles=angles,
              T_target=np.array(T_t), Tmult_target=np.array(Tm_t))
    np.savez_compressed(os.path.join(GOLD, 'f5_filter.npz'), **dd)
    print('wrote f5_filter.npz', out.shape, np.abs(out).max(), np.array(T_t))


if __name__ == '__main__':
    os.makedirs(GOLD, exist_ok=True)
    which = sys.argv[1:] or ['c1', 'c2', 'c3', 'c4', 'c5', 't2', 't3', 'f5']
    for w in which:
        t = time.time()
        globals()[w]()
        print(f'{w}: {time.time() - t:.1f} s')
