"""Fan orchestration, heat maps and eigenray search on the GPU.

Host-side mirror of the reference's ``Simulation2D`` / ``EigenraySim2D`` (simulation.py:29-314): same
constructor arguments, ``cast(*angles, pack=, target=, + propagate kwargs)``, ``raypacks``, ``plot`` and the
restored ``heatmap(resolution=[50, 25], reduction_power=.5, cutoff=.02, pack=)`` whose body the reference
ships commented out (simulation.py:96-138).  ``cast`` hands the whole angle list to ONE batched device
cast instead of looping ``for angle in angles`` (simulation.py:89-94).
"""

import ctypes as C
import os

import numpy as np

from .rays import RayFan, RayPack2D, _torch, _raw_stream
from . import _lib
from . import rays as rays_mod
from . import parallel


class Simulation2D:

    def __init__(self, env, source, **kwargs):
        """
        :param env: Environment2D
        :param source: source point (x, z)
        :param kwargs/pack_default: default pack name ('main'; None = create packs by hand)
        :param kwargs/verbose, verbose_depth, verbose_depth_max: accepted as in the reference
        :param kwargs/distributed: (build-specific) one process per GPU under torch.distributed: ``cast`` keeps this
            rank's share of every fan (``shard='interleave'|'block'``, parallel.shard_angles) and ``heatmap`` combines
            the ranks with the two all-reduces of parallel.py (plot range, then the map); every rank returns the
            whole map.  Without an initialised process group it changes nothing.
        """
        self.distributed = bool(kwargs.get('distributed', False))
        self.shard = kwargs.get('shard', 'interleave')
        self.verbose = kwargs.get('verbose', False)
        self.verbose_depth_max = kwargs.get('verbose_depth_max', 0)
        self.verbose_depth = kwargs.get('verbose_depth', 0)
        # the reference only sets this under verbose=True and then reads it unconditionally (SURVEY trap 4);
        # here it is always defined so cast() also works with verbose=False
        self.verbose_next = self.verbose and ((self.verbose_depth < self.verbose_depth_max) or (self.verbose_depth_max == -1))
        self._vi = '\t' * self.verbose_depth + self.__class__.__name__ + ': '

        self.env = env
        self.source = source
        self.raypacks = dict()
        self.pack_default = kwargs.get('pack_default', 'main')
        if self.pack_default:
            self.raypacks[self.pack_default] = RayPack2D()

        self.range_min = self.env.range_min
        self.range_max = self.env.range_max
        self.size = self.env.size

    def __repr__(self):
        return self.__class__.__name__ + ' object'

    def cast(self, *angles, **kwargs):
        """Propagate rays at the given angles and add them to a pack (reference: simulation.py:69-94).

        ``angles`` may also be one array.  Propagate kwargs (backprop, dx_max, dz_max, n_steps_max,
        n_rebounds_max) and ``target`` are forwarded.  ``store=False`` keeps only per-ray scalars
        (termination, rebounds, distance to target) -- enough for sweeps.  ``strict=True`` raises
        ValueError if any ray leaves the slope tables, as the reference's loop would.
        Returns the RayFan.
        """
        pack = kwargs.get('pack', self.pack_default)
        if pack not in self.raypacks:
            self.raypacks[pack] = RayPack2D()
        raypack = self.raypacks[pack]
        if len(angles) == 1 and (np.ndim(angles[0]) == 1 or hasattr(angles[0], 'data_ptr')):
            angles = angles[0]
        on_device = hasattr(angles, 'data_ptr') and getattr(angles, 'is_cuda', False)      # a device tensor is used where it is
        if on_device and self.distributed and parallel.active():
            angles, on_device = angles.cpu().numpy(), False
        if not on_device:
            angles = np.asarray(angles, dtype=np.float64)
        if self.distributed and parallel.active():
            angles, _ = parallel.shard_angles(angles, mode=kwargs.get('shard', self.shard))
            raypack.sharded = True            # this pack holds one rank's share: heatmap() combines the ranks
        # A wave holds 64 consecutive rays: neighbours in angle step, rebound and end together, unrelated rays make the wave
        # take the general step almost every time and wait for its longest ray (1M shuffled rays: 3x the time of the same
        # rays in order).  Angles that are not a monotonic sweep are therefore cast in sorted order; the pack gets the rays
        # in the caller's order (ray.py:451-465 keeps the first ray of every angle, in the order given), and the returned
        # fan says where each of its rays came from (fan.order[k] = position of fan ray k among the caller's angles).
        order = None
        if not on_device and kwargs.get('sort', True) and angles.ndim == 1 and angles.size > 64:
            step = np.diff(angles)
            if not (bool((step >= 0).all()) or bool((step <= 0).all())):
                torch = _torch()                          # (sorted on the device: a host argsort of 1M angles takes 80 ms)
                angles, order = torch.sort(torch.from_numpy(np.ascontiguousarray(angles)).cuda(), stable=True)
        store = kwargs.get('store', True)
        # a big fan without trajectories is cast lazily: if a heat map is the first thing asked of it, the chunked heat
        # map casts it on its way (RayFan.heatmap_accumulate_chunked) and no counting pass runs
        lazy = (not store) and kwargs.get('target', None) is None and len(angles) >= rays_mod.CHUNKED_MIN_RAYS
        fan = RayFan(self.env, self.source, angles, target=kwargs.get('target', None), store=store, lazy=lazy,
                     **{k: kwargs[k] for k in RayFan.PROPAGATE_DEFAULTS if k in kwargs})
        if kwargs.get('strict', False) and bool((fan.status == _lib.RAY_ERR_INTERP).any().item()):
            raise ValueError('A value in x_new is outside the interpolation range.')
        fan.order = order
        if order is None:
            raypack.add_fan(fan)
        else:
            back = _torch().empty_like(order)
            back[order] = _torch().arange(len(order), device=order.device)
            fan._whole_in_pack = True                   # every ray of the fan, permuted: heat maps may take the fan as it is
            raypack.add_fan(fan, back)
        return fan

    @staticmethod
    def _resolution(kwargs):
        res = np.asarray(kwargs.get('resolution', np.array([50, 25])))
        if not np.issubdtype(res.dtype, np.integer):
            if np.any(res != np.round(res)):
                raise TypeError('heatmap resolution must be integral (the reference floor-divides integer coordinates)')
            res = res.astype(np.int64)
        return res

    def _heatmap_one_fan(self, fan, res, kwargs, finalize, dist_on=False):
        """Heat map of a pack that is ONE whole fan, without waiting for the device before the end: the map is binned
        into the shape of the environment's rectangle (an upper bound of the plot range, known on the host), normalised
        there, and cropped to ceil(size_plot / res) + 1 (simulation.py:117) once the fan's summary -- plot range, "every
        ray fitted its room" -- has been read, in the one device -> host read of the whole cast + heat map.  Valid when
        no ray starts left of x = 0 or rises above z = 1 (then every point falls inside both maps, at the same cell);
        returns None otherwise, or when the fan had to be cast again, and the caller takes the general path.

        dist_on (the fan is this rank's share of a sharded cast): the same with the two collectives queued on the stream --
        the ranks' maps are summed in the rectangle's shape (C2) and the plot range is combined (C1) AFTER the binning, which
        does not need it here -- so that a rank still reads the device once per heat map.  Every rank must take the same
        decisions: whatever would make one rank leave this path (a fan cast again, repeated angles, an empty share) travels
        as a flag in C1's max, and then all ranks leave it together.  Only what is the same on all ranks (environment,
        resolution) may return before the collectives."""
        torch = _torch()
        env = self.env
        lo = np.minimum(0., np.asarray(env.range_min, dtype=np.float64))
        hi = np.maximum(0., np.asarray(env.range_max, dtype=np.float64))
        lazy = bool(fan.__dict__.get('_uncast'))
        if lo[0] <= -1 or hi[1] >= 1 + 1e-9:
            return None
        bad = fan.n == 0                                   # (local reasons to leave: returned at once alone, flagged among ranks)
        if not dist_on and (bad or not (fan._pending or lazy)):
            return None
        shape_b = parallel.heatmap_shape(lo, hi, res)
        if int(shape_b[0]) * int(shape_b[1]) >= (1 << 28):
            return None
        device = fan.device
        with torch.cuda.device(device):
            # a pack keeps the first ray of every angle (ray.py:454): a strictly monotonic sweep has no repeats.  Checked on
            # the host copy of the angles when there is one (while the device is busy with the cast), else on the device
            distinct = None
            if fan.n > 1:
                if fan._angles_host is not None:
                    step = np.diff(fan._angles_host)
                    if not (bool((step > 0).all()) or bool((step < 0).all())):
                        if not dist_on:
                            return None
                        bad = True
                else:
                    a = fan.angles_dev
                    distinct = (a[1:] > a[:-1]).all() | (a[1:] < a[:-1]).all()
            cells = int(shape_b[0]) * int(shape_b[1])
            zeroed = torch.zeros(cells + 1, dtype=torch.float64, device=device)      # the map and the dropped-points counter: one fill
            H, dropped = zeroed[:cells], zeroed[cells:].view(torch.int64)
            out_b, scratch, finalized = None, None, False
            if bad:
                pass
            elif lazy:
                # no trajectories kept and nothing cast yet: chunk by chunk (cast with trajectories -> bin -> drop)
                if not fan.heatmap_accumulate_chunked(H, res, shape_b, freq=kwargs.get('freq', 1), dropped=dropped):
                    if not dist_on:
                        return None
                    bad = True
            elif (kwargs.get('fast', True) and kwargs.get('small', True) and kwargs.get('stream', None) is None
                  and fan.small_tail_room(kwargs.get('freq', 1))):
                # a small fan: gain + binning of every ray in ONE launch (no gain array), and -- when nothing has to be
                # exchanged between ranks first -- max + normalisation queued by the same call
                if finalize and not dist_on:
                    out_b = torch.empty((int(shape_b[1]), int(shape_b[0])), dtype=torch.float64, device=device)
                    scratch = torch.empty(2, dtype=torch.float64, device=device)
                    finalized = True
                fan.heatmap_small(H, res, shape_b, freq=kwargs.get('freq', 1), dropped=dropped, out=out_b if finalized else None,
                                  scratch=scratch if finalized else None, cutoff=float(kwargs.get('cutoff', .02)),
                                  reduction_power=float(kwargs.get('reduction_power', .5)))
            else:
                mask = fan._ok_mask if getattr(fan, '_ok_mask', None) is not None else (fan.status == 0).to(torch.uint8)
                fan.heatmap_accumulate(H, res, shape_b, freq=kwargs.get('freq', 1), ray_mask=mask, dropped=dropped,
                                       fast=kwargs.get('fast', True), hint=kwargs.get('hint', True), stream=kwargs.get('stream', None),
                                       defer=True)
            ext = None
            if dist_on:
                # C1 and the flag in one max-all-reduce of five numbers, C2 (with the dropped-point count as a sixth map cell):
                # both on the stream, nothing read yet
                f64 = dict(dtype=torch.float64, device=device)
                ext = torch.zeros(5, **f64)
                sd = getattr(fan, '_summary_dev', None)
                if sd is not None and not bad:
                    ext[:4] = sd[4:8]
                    if fan._pending:
                        ext[4] = ((sd[1] > float(fan._room * 64)) | (sd[2] != 0) | (sd[8] != 0)).to(torch.float64)
                    if distinct is not None:
                        ext[4] = torch.maximum(ext[4], (~distinct).to(torch.float64))
                    if fan.__dict__.get('_viol_pending') is not None:         # (a ray the one pass could not bin: all ranks start over)
                        ext[4] = torch.maximum(ext[4], (fan._viol_pending[0] != 0).to(torch.float64))
                else:
                    ext[4] = 1.0
                parallel._all_reduce(ext, parallel._dist().ReduceOp.MAX)
                zeroed[cells] = dropped[0].to(torch.float64)                      # (the counter travels as a double in the same sum)
                parallel.allreduce_heatmap(zeroed)
                dropped = zeroed[cells:].round().to(torch.int64)
            if finalize and not finalized:
                out_b = torch.empty((int(shape_b[1]), int(shape_b[0])), dtype=torch.float64, device=device)
                scratch = torch.empty(2, dtype=torch.float64, device=device)
                shape_c = (C.c_int32 * 2)(int(shape_b[0]), int(shape_b[1]))
                _lib.check(_lib.lib().rt2d_heatmap_finalize(H.data_ptr(), shape_c, float(kwargs.get('cutoff', .02)),
                                                            float(kwargs.get('reduction_power', .5)), out_b.data_ptr(),
                                                            scratch.data_ptr(), _raw_stream()))
            # -- the one read --
            if dist_on:
                got = ext.tolist()
                fan.validate()                                # (a fan that must be cast again is, before every rank starts over)
                self._violated([fan])                         # (marks this rank's fan; the verdict itself travelled in got[4])
                if got[4] != 0:
                    return None
                rmin, rmax = -np.array(got[0:2]), np.array(got[2:4])
                if rmin[0] <= -1 or rmax[1] >= 1:
                    return None                               # (the combined range: the same verdict on every rank)
                shape = parallel.heatmap_shape(rmin, rmax, res)
                return H.view(int(shape_b[0]), int(shape_b[1])), out_b, shape, dropped
            if fan.validate() or getattr(fan, '_onepass_overflow', False):
                fan.__dict__.pop('_viol_pending', None)
                return None                                   # cast again with counted sizes, or a one-pass map without rows
                                                              # enough for its side rays: start over
            if self._violated([fan]):
                return None                                   # a ray the one pass could not bin exactly: start over, the exact way
            if distinct is not None and not bool(distinct.item()):
                return None                                   # repeated angles: the pack keeps the first of each
        summ = fan._summary
        rmin, rmax = -np.array(summ[4:6]), np.array(summ[6:8])
        if rmin[0] <= -1 or rmax[1] >= 1:
            return None
        shape = parallel.heatmap_shape(rmin, rmax, res)
        return H.view(int(shape_b[0]), int(shape_b[1])), out_b, shape, dropped

    def _violated(self, fans, dist_on=False, device=None):
        """True when a one-pass or fused binning queued by this call met a ray that its binner could not bin exactly (the
        masks are meant to keep such rays out: RayFan.fused_mask, rt2d_classify_rays; the kernels count what slips through,
        RayFan.last_fused_violations).  The map is then incomplete: the fans are marked so that their rays take the gain
        array + binning kernels from now on, and the caller starts over.  Among ranks the verdict is shared (a maximum), so
        that all of them start over together."""
        torch = _torch()
        flag = 0
        for fan in fans:
            v = fan.__dict__.pop('_viol_pending', None)
            if v is not None and int(v.item()) != 0:
                fan._exact_binning = True
                flag = 1
        if dist_on:
            t = torch.tensor([float(flag)], dtype=torch.float64, device=device)
            parallel._all_reduce(t, parallel._dist().ReduceOp.MAX)
            flag = int(t.item())
        return bool(flag)

    def heatmap_full(self, **kwargs):
        """Un-normalised heat map (nx, nz) as a device tensor, plus the number of dropped points."""
        return self._heatmap(kwargs, finalize=False)

    def _heatmap(self, kwargs, finalize):
        torch = _torch()
        res = self._resolution(kwargs)
        freq = kwargs.get('freq', 1)
        raypack = self.raypacks[kwargs.get('pack', self.pack_default)]
        dist_on = getattr(raypack, 'sharded', False) and parallel.active()
        fan = raypack._sole_pending_fan()
        one_fan_dist = os.environ.get('RT2D_DIST_ONE_FAN', '1') != '0'            # (0: the general path for sharded packs, as before round 2)
        if fan is not None and (fan.store or fan.__dict__.get('_uncast')) and not kwargs.get('stream', None) and (one_fan_dist or not dist_on):
            got = self._heatmap_one_fan(fan, res, kwargs, finalize, dist_on=dist_on)
            if got is not None:
                Hb, out_b, shape, dropped = got
                nx, nz = int(shape[0]), int(shape[1])
                return (out_b[:nz, :nx] if finalize else Hb[:nx, :nz]), dropped
        segs = raypack.segments()
        device = segs[0][0].device if segs else torch.device('cuda', torch.cuda.current_device())
        if dist_on:                                                       # C1: the plot range of ALL ranks' rays
            lo, hi = parallel.allreduce_plot_range(raypack.range_min_plot, raypack.range_max_plot, device=device)
            shape = parallel.heatmap_shape(lo, hi, res)
        else:
            shape = np.ceil(raypack.size_plot / res).astype(int) + 1     # simulation.py:117
        H = torch.zeros(int(shape[0]) * int(shape[1]), dtype=torch.float64, device=device)
        dropped = torch.zeros(1, dtype=torch.int64, device=device)
        for fan, idx in segs:
            mask = None
            if len(idx) != fan.n:
                mask = torch.zeros(fan.n, dtype=torch.uint8, device=device)
                mask[torch.as_tensor(np.asarray(idx, dtype=np.int64), device=device)] = 1
            if (kwargs.get('fast', True) and kwargs.get('small', True) and kwargs.get('stream', None) is None
                    and fan.store and not fan._pending and fan.small_tail_room(freq)):
                fan.heatmap_small(H, res, shape, freq=freq, ray_mask=mask, dropped=dropped)      # one launch: gain + binning
                continue
            fan.heatmap_accumulate(H, res, shape, freq=freq, ray_mask=mask, dropped=dropped, fast=kwargs.get('fast', True),
                                   hint=kwargs.get('hint', True), stream=kwargs.get('stream', None))
        if self._violated([fan for fan, _ in segs], dist_on, device):
            return self._heatmap(kwargs, finalize)                       # (the fans now take the exact binning: once more)
        if dist_on:                                                       # C2: sum of the ranks' maps
            parallel.allreduce_heatmap(H)
            parallel._all_reduce(dropped, parallel._dist().ReduceOp.SUM)
        H = H.view(int(shape[0]), int(shape[1]))
        if not finalize:
            return H, dropped
        nx, nz = H.shape
        out = torch.empty((nz, nx), dtype=torch.float64, device=H.device)
        scratch = torch.empty(2, dtype=torch.float64, device=H.device)
        shape_c = (C.c_int32 * 2)(int(nx), int(nz))
        with torch.cuda.device(H.device):
            _lib.check(_lib.lib().rt2d_heatmap_finalize(H.data_ptr(), shape_c, float(kwargs.get('cutoff', .02)),
                                                        float(kwargs.get('reduction_power', .5)), out.data_ptr(),
                                                        scratch.data_ptr(), _raw_stream()))
        return out, dropped

    def heatmap(self, **kwargs):
        """Ray-power heat map (reference: commented block simulation.py:96-138).

        :param kwargs/resolution: cell size (x, z) in metres, integers (default [50, 25])
        :param kwargs/reduction_power: exponent applied after normalisation (default .5)
        :param kwargs/cutoff: saturation of the normalised map (default .02)
        :param kwargs/pack: pack name
        :param kwargs/freq: frequency whose gain weights the rays (default 1, as the reference)
        :param kwargs/as_tensor: return the device tensor instead of a NumPy array
        Returns the (nz, nx) map, i.e. ``heatmap_plot.T``.
        """
        out, _ = self._heatmap(kwargs, finalize=True)
        return out if kwargs.get('as_tensor', False) else out.cpu().numpy()

    def heatmaps(self, freqs, **kwargs):
        """One heat map per frequency (BASELINE config 4: an array of frequencies, frequency-dependent rebound loss).
        The gains of all frequencies come from ONE launch (eight frequencies per lane share the six absorption table
        look-ups of a point), all maps are binned in ONE pass over the trajectories (the cell of a point does not depend
        on the frequency: rt2d_heatmap_accumulate_multi), normalised by one launch and copied back once.
        ``batched=False``: one ``heatmap(freq=f)`` per frequency instead.  Returns {freq: (nz, nx) array}."""
        torch = _torch()
        freqs = list(freqs)
        pack = self.raypacks[kwargs.get('pack', self.pack_default)]
        segs = pack.segments()
        dist_on = getattr(pack, 'sharded', False) and parallel.active()
        if not kwargs.get('batched', True) or not segs or not all(fan.store for fan, _ in segs) or not freqs:
            if all(fan.store for fan, _ in segs):
                pack.populate(*freqs)
            return {f: self.heatmap(freq=f, **kwargs) for f in freqs}
        res = self._resolution(kwargs)
        device = segs[0][0].device
        if dist_on:
            lo, hi = parallel.allreduce_plot_range(pack.range_min_plot, pack.range_max_plot, device=device)
            shape = parallel.heatmap_shape(lo, hi, res)
        else:
            shape = np.ceil(pack.size_plot / res).astype(int) + 1
        nx, nz, F = int(shape[0]), int(shape[1]), len(freqs)
        with torch.cuda.device(device):
            H = torch.zeros((F, nx * nz), dtype=torch.float64, device=device)
            for fan, idx in segs:
                mask = None
                if len(idx) != fan.n:
                    mask = torch.zeros(fan.n, dtype=torch.uint8, device=device)
                    mask[torch.as_tensor(np.asarray(idx, dtype=np.int64), device=device)] = 1
                fan.heatmap_accumulate_multi(H, res, shape, freqs, ray_mask=mask)
            if dist_on:
                parallel.allreduce_heatmap(H)
            out = torch.empty((F, nz, nx), dtype=torch.float64, device=device)
            scratch = torch.empty(2 * F, dtype=torch.float64, device=device)
            _lib.check(_lib.lib().rt2d_heatmap_finalize_multi(H.data_ptr(), (C.c_int32 * 2)(nx, nz), F, float(kwargs.get('cutoff', .02)),
                                                              float(kwargs.get('reduction_power', .5)), out.data_ptr(),
                                                              scratch.data_ptr(), _raw_stream()))
        if kwargs.get('as_tensor', False):
            return {f: out[k] for k, f in enumerate(freqs)}
        # one copy of all maps through page-locked memory (38 MB for 64 maps of 1500 x 51: 10 ms pageable, 2 ms pinned)
        pinned = torch.empty(out.shape, dtype=out.dtype, pin_memory=True)
        pinned.copy_(out, non_blocking=True)
        torch.cuda.current_stream().synchronize()
        host = pinned.numpy()                     # (the arrays keep the page-locked buffer alive)
        return {f: host[k] for k, f in enumerate(freqs)}

    def plot_heatmap(self, ax=None, floor=True, **kwargs):
        """heatmap(**kwargs) drawn with metre ticks and the sea floor painted over it (deprecated.ipynb cell 4)."""
        from . import presentation
        res = kwargs.get('resolution', np.array([50, 25]))
        show = {k: kwargs.pop(k) for k in ('cmap', 'vmin', 'vmax', 'aspect') if k in kwargs}
        return presentation.plot_heatmap(self.heatmap(**kwargs), res, env=self.env if floor else None, ax=ax, **show)

    def plot(self, fig, **kwargs):
        pack = kwargs.pop('pack', self.pack_default)
        for ray in self.raypacks[pack].rays:
            ray.plot(fig, **kwargs)
        self.env.plot(fig, c='red')


class EigenraySim2D(Simulation2D):
    """Sweep + refine search of the rays that connect source and target (reference: simulation.py:158-314), and the
    eigenray filter built from them (get_filter_NEW, simulation.py:410-480)."""

    _PROP_KEYS = tuple(RayFan.PROPAGATE_DEFAULTS)

    def __init__(self, env, source, target, n_rays, n_rebounds_max, **kwargs):
        self.init_kwargs = kwargs
        self.pack_temp_scan = '__iter0-scan'
        self.pack_scan = 'scan'
        self.gen_pack_temp_refine = lambda n_refines, ray_id: f'__iter{n_refines}-refine-ray{ray_id}'
        self.gen_pack_refine = lambda n_refines: f'refine-{n_refines}'

        super(EigenraySim2D, self).__init__(env, source, pack_default=None, **kwargs)
        self.target = target
        self.n_rebounds_max = n_rebounds_max
        self.n_rays = n_rays
        self.precision_cutoff = kwargs.get('precision_cutoff', 0.1)
        self.dist = dict()
        self.dist_avg = None
        self.n_refines = 0
        self._angular_precision = None

        self.n_rays_scan = kwargs.get('n_rays_scan', self.n_rays * 10)
        self._scan(kwargs.get('scan_angle_min', None), kwargs.get('scan_angle_max', None))

    def _prop(self):
        return {k: self.init_kwargs[k] for k in self._PROP_KEYS if k in self.init_kwargs}

    def _scan(self, angle_min=None, angle_max=None):
        # aperture from mean depths and the rebound budget (simulation.py:218-228)
        Dz = abs(self.env.ceil_avg - self.env.floor_avg)
        Dx = abs(self.target[0] - self.source[0])
        Dz_sc = abs(self.source[1])
        Dz_tc = abs(self.target[1])
        N = self.n_rebounds_max + 1
        if N % 2 == 0:
            angle_min = angle_min if angle_min else -1 * np.arcsin(Dx / (N * Dz - Dz_sc + Dz_tc))
            angle_max = angle_max if angle_max else np.arcsin(Dx / (N * Dz + Dz_sc - Dz_tc))
        else:
            angle_min = angle_min if angle_min else -1 * np.arcsin(Dx / ((N + 1) * Dz - Dz_sc - Dz_tc))
            angle_max = angle_max if angle_max else np.arcsin(Dx / ((N - 1) * Dz + Dz_sc + Dz_tc))

        if self.verbose:
            print(f'{self._vi}Scanning using {self.n_rays_scan} rays between angles {angle_min} and {angle_max}')
        self._angular_precision = (angle_max - angle_min) / (self.n_rays_scan - 1)

        # sweep: scalars only, one batched cast (simulation.py:237-238)
        angles = np.linspace(angle_min, angle_max, self.n_rays_scan)
        kw = self._prop()
        kw['n_rebounds_max'] = self.n_rebounds_max
        torch = _torch()
        dist_on = self.distributed and parallel.active()
        shard_idx = None
        if dist_on:
            angles, shard_idx = parallel.shard_angles(angles, mode=self.shard)
        sweep = RayFan(self.env, self.source, angles, target=self.target, store=False, **kw)
        self.scan_fan = sweep
        # keep the rays whose distance is among the n_rays smallest distinct distances (simulation.py:241-249):
        # top-k on the device, ties kept
        dist = sweep.dist.clone()
        dist[sweep.status != 0] = float('nan')
        if dist_on:
            # C3: every rank contributes the rays within its own n_rays smallest distinct distances; the merged list applies
            # the single-process rule below to them (ties kept, equal distances in fan order) and is the same on every rank
            best_d, best_a = parallel.allgather_topk(dist, sweep.angles_dev, self.n_rays, index_local=shard_idx)
            if best_d.numel() == 0:
                raise KeyError('no scanning ray reaches the target abscissa')
            sel_angles = best_a.cpu().numpy()
        else:
            valid = ~torch.isnan(dist)
            uniq = torch.unique(dist[valid])                       # sorted ascending
            if uniq.numel() == 0:
                raise KeyError('no scanning ray reaches the target abscissa')
            kth = uniq[min(self.n_rays, uniq.numel()) - 1]
            sel = torch.nonzero(valid & (dist <= kth)).flatten()
            sel = sel[torch.argsort(dist[sel], stable=True)]
            sel_angles = angles[sel.cpu().numpy()]
        # re-cast the selected rays with trajectories so they are full Ray2D objects
        self.raypacks[self.pack_scan] = RayPack2D()
        fan = RayFan(self.env, self.source, sel_angles, target=self.target, store=True, **kw)
        self.raypacks[self.pack_scan].add_fan(fan)
        self.pack_default = self.pack_scan
        self.dist_avg = np.mean(self.raypacks[self.pack_scan].dist_sorted)
        if self.verbose:
            print(f'{self._vi}Scan mean distance: {self.dist_avg}')

    def refine(self, **kwargs):
        """Refine each kept ray inside a cone of +-cone_half_angle (reference: simulation.py:255-314).
        All cones of one iteration go into ONE batched cast."""
        cone_half_angle = kwargs.get('cone_half_angle', self._angular_precision)
        n_rays = kwargs.get('n_rays', 5)
        iterations = kwargs.get('iterations', 1)
        force_refine = kwargs.get('force_refine', False)
        torch = _torch()

        for _ in range(iterations):
            if self.dist_avg <= self.precision_cutoff and not force_refine:
                if self.verbose:
                    print(f'{self._vi}Target precision reached')
                break
            raypack_prev = self.raypacks[self.pack_default]
            self.n_refines += 1
            pack_refine = self.gen_pack_refine(self.n_refines)
            self.raypacks[pack_refine] = RayPack2D()
            raypack_refine = self.raypacks[pack_refine]

            prev = list(raypack_prev.rays)
            k = len(prev)
            # note: like the reference (simulation.py:298), refine casts WITHOUT n_rebounds_max
            cones = np.stack([np.linspace(r.angle - cone_half_angle, r.angle + cone_half_angle, n_rays) for r in prev])
            # Only the distance to the target decides which ray of a cone is kept (simulation.py:272-296), so the cones are
            # cast WITHOUT trajectories (one pass, per-ray scalars); the winners' trajectories are cast when something
            # asks for them (RayFan.subset): the kernel is deterministic, the second cast of an angle gives the same ray.
            # (Without a rebound limit some cone rays run to n_steps_max; storing all of them meant a provisional pass,
            # then a counting and a storing pass of 100 000 steps each, per iteration.)
            fan = RayFan(self.env, self.source, cones.reshape(-1), target=self.target, store=False, **self._prop())
            self.last_cone_fan = fan                              # (diagnostics: lane utilisation of the cones, bench.py)
            dist = fan.dist.clone()
            dist[fan.status != 0] = float('nan')
            dist = dist.view(k, n_rays).cpu().numpy()
            picks = []                                            # per parent: -1 = the parent stays, else index into the cone fan
            for ray_id, ray in enumerate(prev):
                # candidates in the reference's order: the cone rays (duplicate angles skipped), then the parent
                cand_d, cand = [], []
                seen = set()
                for j in range(n_rays):
                    a = float(cones[ray_id, j])
                    if a in seen:
                        continue
                    seen.add(a)
                    cand_d.append(dist[ray_id, j]); cand.append(j)
                if ray.angle not in seen:
                    cand_d.append(np.nan if ray.dist_to_target is None else ray.dist_to_target); cand.append(-1)
                cand_d = np.asarray(cand_d, dtype=np.float64)
                if np.all(np.isnan(cand_d)):
                    best = 0                                      # np.sort puts NaN last; dist_sorted[0] is NaN
                else:
                    best = int(np.nanargmin(cand_d))              # first of the smallest distance
                picks.append(-1 if cand[best] == -1 else ray_id * n_rays + cand[best])
            won = [p for p in picks if p >= 0]
            winners = fan.subset(won) if won else None            # scalars now, trajectories if and when they are asked for
            pos = 0
            for ray, p in zip(prev, picks):                       # the pack in the parents' order, as the reference builds it
                if p < 0:
                    raypack_refine.add(ray)
                else:
                    raypack_refine.add_fan(winners, [pos])
                    pos += 1
            self._angular_precision /= n_rays
            self.dist_avg = np.mean(raypack_refine.dist_sorted)
            if self.verbose:
                print(f'{self._vi}Refine #{self.n_refines} mean distance: {self.dist_avg}')
            self.pack_default = pack_refine

    # -- eigenray filter (reference: simulation.py:410-480) ------------------------------------------------
    def target_products(self, *freqs, **kwargs):
        """Per ray of the pack, in increasing angle: travel time to the target and transmittance at every frequency
        (Ray2D.gen_filter, ray.py:382-405).  Returns (angles (n,), T_target (n,), Tmult_target (n, F))."""
        raypack = self.raypacks[kwargs.get('pack', self.pack_default)]
        if kwargs.get('batched', True):
            out = self._target_products_batched(raypack, freqs)
            if out is not None:
                return out
        angles_sorted = np.sort(list(raypack.angles.keys()))
        T_t, Tm_t = [], []
        for angle in angles_sorted:
            ray = raypack.angles[angle]
            ray.populate_NEW(*freqs)                      # one batched gain launch per fan, then views
            T_target, Tmult_target = ray.gen_filter(self.target)
            T_t.append(T_target)
            Tm_t.append(Tmult_target)
        return angles_sorted, np.asarray(T_t, dtype=np.float64), np.asarray(Tm_t, dtype=np.float64)

    def _target_products_batched(self, raypack, freqs):
        """target_products for packs made of whole-fan segments whose rays carry exactly these frequencies: one
        rt2d_target_products launch per fan (one wave per ray) instead of a host loop over rays and frequencies.
        Returns None when the pack does not qualify (loose rays, rays populated at other frequencies too)."""
        torch = _torch()
        segs = raypack.rays._segments
        if not segs or any(fan is None for fan, _ in segs):
            return None
        fl = sorted(dict.fromkeys(freqs))
        ang, T_all, Tm_all = [], [], []
        tgt = (C.c_double * 2)(float(self.target[0]), float(self.target[1]))
        for fan, idx in segs:
            if any(f not in fl for f in fan.G_end):
                return None                               # gen_filter uses every frequency a ray has: leave it to the loop
            fan.populate(*fl)
            if any(f not in fan.G for f in fl):
                return None
            T = fan.post()['T']
            with torch.cuda.device(fan.device):
                ptrs = torch.tensor([fan.G[f].data_ptr() for f in fl], dtype=torch.int64, device=fan.device)
                fr = torch.tensor(fl, dtype=torch.float64, device=fan.device)
                ids = torch.as_tensor(np.asarray(idx, dtype=np.int32), device=fan.device)
                k = len(idx)
                step = torch.empty(k, dtype=torch.int32, device=fan.device)
                T_t = torch.empty(k, dtype=torch.float64, device=fan.device)
                Tm = torch.empty((len(fl), k), dtype=torch.float64, device=fan.device)
                _lib.check(_lib.lib().rt2d_target_products(
                    self.env.device_handle(fan.device.index), fan.X.data_ptr(), fan.Z.data_ptr(), T.data_ptr(), ptrs.data_ptr(),
                    fan.tile_off.data_ptr(), fan.n_pts.data_ptr(), ids.data_ptr(), k, tgt, fr.data_ptr(), len(fl),
                    step.data_ptr(), T_t.data_ptr(), Tm.data_ptr(), _raw_stream()))
            ang.append(np.asarray(fan.angles, dtype=np.float64)[np.asarray(idx, dtype=np.int64)])
            T_all.append(T_t.cpu().numpy())
            Tm_all.append(Tm.cpu().numpy().T)
        ang, T_all, Tm_all = np.concatenate(ang), np.concatenate(T_all), np.concatenate(Tm_all, axis=0)
        order = np.argsort(ang, kind='stable')
        return ang[order], T_all[order], Tm_all[order]

    def get_filter_NEW(self, *freqs, **kwargs):
        """Filter that turns an emitted signal into the signal received at the target: every eigenray contributes the
        signal weighted per frequency by its transmittance, delayed by its travel time (reference: simulation.py:410-480).
        The returned ``filter(sample_rate, signal)`` runs on the GPU: one forward FFT, the rays' spectra as one
        (n_rays, n_samples) batch, one batched inverse FFT (rocFFT through torch.fft), then the delayed sum in the
        reference's order.  Returns (sample_rate, filtered signal) like the reference."""
        torch = _torch()
        rf = np.array(freqs, dtype=np.float64)
        _, T_t, Tm_t = self.target_products(*freqs, **kwargs)
        T_target_min, T_target_max = np.min(T_t), np.max(T_t)
        T_stretch_total = T_target_max - T_target_min
        device = torch.device('cuda', torch.cuda.current_device())
        xp = torch.from_numpy(np.concatenate((-1 * rf[::-1], rf))).to(device)                    # simulation.py:470
        fp = torch.from_numpy(np.concatenate((Tm_t[:, ::-1], Tm_t), axis=1)).to(device)          # (n_rays, 2F)

        def filter(sample_rate, signal):
            signal = np.asarray(signal)
            n_samples = signal.shape[0]
            data_add = int(np.ceil(T_stretch_total * sample_rate))
            yf = torch.fft.fft(torch.from_numpy(np.ascontiguousarray(signal, dtype=np.float64)).to(device))
            xf = torch.from_numpy(np.fft.fftfreq(n_samples, 1 / sample_rate)).to(device)
            # np.interp(xf, xp, fp) for every ray at once: end values outside [xp[0], xp[-1]]
            hi = torch.searchsorted(xp, xf, right=True).clamp(1, xp.numel() - 1)
            lo = hi - 1
            w = ((xf - xp[lo]) / (xp[hi] - xp[lo])).clamp(0., 1.)
            interp = fp[:, lo] + (fp[:, hi] - fp[:, lo]) * w                                     # (n_rays, n_samples)
            sig_rays = torch.fft.ifft(interp * yf.unsqueeze(0), dim=1).real * 0.01               # simulation.py:474-475
            out = torch.zeros(n_samples + data_add, dtype=torch.float64, device=device)
            offsets = np.ceil((T_t - T_target_min) * sample_rate).astype(int)
            for r in range(sig_rays.shape[0]):                                                   # the reference's order
                out[int(offsets[r]):int(offsets[r]) + n_samples] += sig_rays[r]
            return sample_rate, out.cpu().numpy()

        return filter
