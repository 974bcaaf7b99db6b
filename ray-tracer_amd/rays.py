"""Ray fans on the GPU and the reference-shaped views over them.

``RayFan`` is the native form: one batched cast of n rays whose results live in torch tensors on
the device (per-ray scalars, rebound log, trajectories in 64-ray tile layout, gains).  ``Ray2D`` and
``RayPack2D`` keep the reference's names, constructor arguments and attributes (ray.py:23-489) and
are thin, lazy views over a fan, so code written against the reference keeps working:

    ray = Ray2D(env, source, angle, target=target); ray.propagate(dx_max=50, dz_max=25)
    ray.XZ, ray.T, ray.steps, ray.stop_reason, ray.n_rebounds, ray.dist_to_target, ...
    ray.populate_NEW(1, 1000.); ray.G[1000.]

Every number comes from the HIP kernels behind include/rt2d.h; there is no CPU fallback.
"""

import ctypes as C
import os

import numpy as np

from .preferences import (DX_MAX_DEFAULT, DZ_MAX_DEFAULT, N_STEPS_MAX, N_REBOUNDS_MAX, REBOUND_LOG_CAP_DEFAULT)
from . import _lib
from ._lib import STOP_REASONS, RAY_ERR_INTERP

TILE = 64
SUMMARY_LEN = 9                  # the 8 doubles of rt2d_fan_summary + one flag word (RayFan._queue_summary)
_UNIFORM_LAYOUTS = {}            # (device, tiles, room) -> tile offsets of a uniform provisional layout (reused between casts)
HEATMAP_FAST_MIN_RAYS = int(os.environ.get('RT2D_HEATMAP_FAST_MIN_RAYS', 6000))    # same switch as the library's
SPECULATIVE_ROOM = 4096          # points per ray a speculative single-pass cast reserves
SPECULATIVE_BYTES = 1 << 30      # ... as long as the provisional X+Z stay below this
# Big fans: 288 GB of HBM make the counting pass optional.  The same single storing pass goes into a PADDED layout
# (SPECULATIVE_ROOM points of room per ray) that is then kept as the fan's layout -- untouched pages cost nothing -- as
# long as X+Z stay below this and a gain array still fits beside them; post() / several frequencies compact it first.
PADDED_BYTES = int(os.environ.get('RT2D_PADDED_BYTES', 96 << 30))
STREAM_MIN_RAYS = int(os.environ.get('RT2D_STREAM_MIN_RAYS', 1 << 17))    # fans from here on bin through the one-pass kernel
CHUNK_RAYS = int(os.environ.get('RT2D_CHUNK_RAYS', 0))               # rays cast with trajectories at a time by the chunked heat map;
                                                                     # 0: as many as a third of the free HBM holds (RayFan._chunk_rays)
CHUNKED_MIN_RAYS = int(os.environ.get('RT2D_CHUNKED_MIN_RAYS', 1 << 18))   # store=False fans from here on use it (smaller: fused kernel)
ONEPASS_MIN_ROWS, ONEPASS_ROWS_DIV = 4096, 8      # rows of the compact gain array of a one-pass map queued without waiting: max(.., n / ..)
CHUNK_STREAMS = int(os.environ.get('RT2D_CHUNK_STREAMS', 2))
MULTI_GROUP = int(os.environ.get('RT2D_MULTI_GROUP', 32))         # frequencies per group of heatmap_accumulate_multi's pipeline (0: off)
MULTI_PIPELINE_MIN = 1 << 20                                      # rays x frequencies from which it is used
MULTI_MINOR = os.environ.get('RT2D_MULTI_MINOR', '1') != '0'      # several maps per pass through the map-minor scratch (rt2d_heatmap_accumulate_multi)
SIDE_STREAM = os.environ.get('RT2D_SIDE_STREAM', '1') != '0'      # side fans of the fused heat map on a second stream
SIDE_FAN_RAYS = 1 << 16          # rays per stored side fan when the fused heat map hands rays to the stored path
LOSS_WS_BYTES = int(os.environ.get('RT2D_LOSS_WS_BYTES', 4 << 30))      # rebound-loss workspace of rt2d_gain, at most


_TORCH = None


def _torch():
    global _TORCH
    if _TORCH is None:                    # (checked once: torch.cuda.is_available() costs ~2 us, and a cast asks five times)
        import torch
        if not torch.cuda.is_available():
            raise _lib.Rt2dError('no GPU visible to torch: this package computes on MI355X only (no CPU fallback)')
        _TORCH = torch
    return _TORCH


def _ptr(t):
    return None if t is None else t.data_ptr()


_SIDE_STREAMS = {}


def _free_bytes(device):
    """HBM this process can still allocate through torch: what the driver reports free plus what torch's caching
    allocator holds without using it (after a big fan has been dropped that is most of the card)."""
    torch = _torch()
    return torch.cuda.mem_get_info(device)[0] + max(0, torch.cuda.memory_reserved(device) - torch.cuda.memory_allocated(device))


def _side_stream(device, which=0):
    """A second (third, ...) stream of a device (small side fans run on it beside the big kernel of the main stream)."""
    torch = _torch()
    key = (device.type, device.index, which)
    if key not in _SIDE_STREAMS:
        _SIDE_STREAMS[key] = torch.cuda.Stream(device=device)
    return _SIDE_STREAMS[key]


_FREQ_TENSORS = {}
_ANGLE_TENSORS = {}


def _raw_stream():
    """hipStream_t of torch's current stream on the current device, as an integer.  torch.cuda.current_stream() costs the
    host ~10 us (a Python object per call); the raw query behind it a fraction of one."""
    import torch
    try:
        return torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice())
    except AttributeError:                      # (a torch without the private query)
        return torch.cuda.current_stream().cuda_stream


def _angles_on_device(host, device):
    """Device copy of a host angle array.  A copy from pageable memory waits for the transfer (~15 us even for 8 KB, on
    the critical path in front of every cast): the copies of the last few small arrays are kept and found again by
    content (hash of the bytes, then an equality check), e.g. the same sweep cast again and again.  Nothing ever writes
    to a fan's angles."""
    import torch
    if host.size > (1 << 16):
        return torch.from_numpy(host).to(device)
    key = (str(device), host.size, hash(host.tobytes()))
    hit = _ANGLE_TENSORS.get(key)
    if hit is not None and np.array_equal(hit[0], host):
        return hit[1]
    t = torch.from_numpy(host).to(device)
    if len(_ANGLE_TENSORS) >= 16:
        _ANGLE_TENSORS.pop(next(iter(_ANGLE_TENSORS)))
    _ANGLE_TENSORS[key] = (host.copy(), t)
    return t


def _freq_tensor(freqs, device):
    """Device tensor of a list of frequencies.  Kept between calls: creating it is a host -> device copy from pageable
    memory, which waits for the stream -- the cast a heat map is queued behind -- before the gain can even be queued."""
    key = (str(device), tuple(float(f) for f in freqs))
    t = _FREQ_TENSORS.get(key)
    if t is None:
        import torch
        t = torch.tensor(list(key[1]), dtype=torch.float64, device=device)
        if len(_FREQ_TENSORS) < 256:
            _FREQ_TENSORS[key] = t
    return t


class RayFan:
    """n rays cast together.  Device tensors; host copies of the per-ray scalars are made on demand."""

    # power_cutoff / cutoff_freq: opt-in termination on ray power (NOT in the reference, whose propagate never looks at
    # the gain): 0 = off; > 0 ends a ray with the first point whose 10**(G[cutoff_freq]/10) is below it ('power-cutoff')
    PROPAGATE_DEFAULTS = dict(backprop=True, dx_max=DX_MAX_DEFAULT, dz_max=DZ_MAX_DEFAULT,
                              n_steps_max=N_STEPS_MAX, n_rebounds_max=N_REBOUNDS_MAX,
                              power_cutoff=0., cutoff_freq=1.)

    # what _cast produces: asking an uncast (lazy) fan for any of these casts it first
    _CAST_FIELDS = frozenset(('range', 'slope_max', 'n_pts', 'stop_code', 'n_rebounds', 'status', 'monotone', '_scalars', 'dist',
                              'step_target', 'T_target', '_init', 'rebound_cap', 'tile_off', 'X', 'Z', 'Cpt', 'total', 'padded',
                              'passes', '_out', 'reb_velocity', 'reb_angle', 'reb_pos_x', 'reb_pos_z', 'reb_step', 'reb_surface'))

    def __getattr__(self, name):
        d = self.__dict__
        if d.get('_uncast') and name in RayFan._CAST_FIELDS:
            d['_uncast'] = False
            self._cast(d.get('_init_host'), False)
            return getattr(self, name)
        raise AttributeError(f'{type(self).__name__!r} object has no attribute {name!r}')

    def __init__(self, env, source, angles, target=None, store=True, device=None, init=None, store_c=False, lazy=False, **kwargs):
        torch = _torch()
        self.env = env
        self.source = np.asarray(source, dtype=np.float64)
        self.target = None if target is None else np.asarray(target, dtype=np.float64)
        self.device = torch.device('cuda', torch.cuda.current_device()) if device is None else torch.device(device)
        self.kw = {k: kwargs.get(k, v) for k, v in self.PROPAGATE_DEFAULTS.items()}
        self.store = bool(store)
        self.keep_log = bool(kwargs.get('keep_log', True))      # store=False: keep the rebound log for the fused heat map
        if torch.is_tensor(angles):
            self.angles_dev = angles.to(device=self.device, dtype=torch.float64).contiguous()
            self._angles_host = None
        else:
            self._angles_host = np.ascontiguousarray(angles, dtype=np.float64).reshape(-1)
            self.angles_dev = _angles_on_device(self._angles_host, self.device)
        self.n = int(self.angles_dev.shape[0])
        self._host = {}
        self.G = {}            # freq -> device tensor (total,) in tile layout
        self.G_end = {}        # freq -> device tensor (n,)
        self._post = None
        self._pending = False      # a single-pass cast whose summary has not been read yet (validate())
        self._force_exact = False
        self._summary = None
        self._init_host = init
        self._lazy_store = False
        self._ok_mask = None       # (status == 0) as uint8, when the cast's summary launch produced it
        # lazy (store=False only): nothing is cast until something asks for a per-ray result -- a heat map asked for first
        # (heatmap_accumulate_chunked) then produces them on its way, and no counting pass ever runs
        self._uncast = bool(lazy) and not self.store and self.n > 0
        if not self._uncast:
            self._cast(init, store_c)

    # -- casting ----------------------------------------------------------------------------------
    def _params(self, rebound_cap):
        p = _lib.Rt2dParams()
        p.dx_max, p.dz_max = float(self.kw['dx_max']), float(self.kw['dz_max'])
        p.n_steps_max, p.n_rebounds_max = int(self.kw['n_steps_max']), int(self.kw['n_rebounds_max'])
        p.backprop = 1 if self.kw['backprop'] else 0
        p.has_target = 0 if self.target is None else 1
        if self.target is not None:
            p.target[:] = [float(self.target[0]), float(self.target[1])]
        p.rebound_cap = int(rebound_cap)
        p.launch_hint = int(self.__dict__.get('launch_hint', 0))
        p.power_cutoff, p.cutoff_freq = float(self.kw['power_cutoff']), float(self.kw['cutoff_freq'])
        return p

    def _cast(self, init, store_c, known_n_pts=None, alloc_only=False, known_reb_max=None):
        torch = _torch()
        kwargs_keep_log = self.keep_log
        L = _lib.lib()
        n, dev = self.n, self.device
        with torch.cuda.device(dev):
            handle = self.env.device_handle(dev.index)
            stream = _raw_stream()
            f64 = dict(dtype=torch.float64, device=dev)
            i32 = dict(dtype=torch.int32, device=dev)
            # every per-ray result out of ONE allocation (a dozen small ones cost the host more than the cast's launch)
            n8 = (n + 7) // 8 * 8
            buf = torch.empty(8 * n8 * 8, dtype=torch.uint8, device=dev)          # 5 doubles + 4 ints + 1 byte per ray, rounded up

            # The tensors are views into it; making a view costs the host ~2 us and there are two dozen of them (with the
            # rebound log): the kernel only needs ADDRESSES, so the single-pass cast is launched from the addresses and the
            # views are made while it runs (make_views; every other path makes them first).
            nbytes_of = {torch.float64: 8, torch.int32: 4, torch.uint8: 1}
            addr, todo = {}, []

            def carve(name, first_double, count, dtype, shape=None, src=None, src_ptr=None):
                addr[name] = (buf.data_ptr() if src is None else src_ptr) + first_double * 8
                todo.append((name, first_double, count, dtype, shape, buf if src is None else src))

            def make_views():
                while todo:
                    name, first_double, count, dtype, shape, src_t = todo.pop(0)
                    v = src_t[first_double * 8:first_double * 8 + count * nbytes_of[dtype]].view(dtype)
                    setattr(self, name, v if shape is None else v.view(*shape))

            carve('range', 0, 4 * n, torch.float64, (4, n))
            carve('slope_max', 4 * n8, n, torch.float64)
            carve('n_pts', 5 * n8, n, torch.int32)
            carve('stop_code', 5 * n8 + n8 // 2, n, torch.int32)
            carve('n_rebounds', 6 * n8, n, torch.int32)
            carve('status', 6 * n8 + n8 // 2, n, torch.int32)
            carve('monotone', 7 * n8, n, torch.uint8)
            self._scalars = None
            if self.target is not None:
                self.dist = torch.full((n,), float('nan'), **f64)
                self.step_target = torch.full((n,), -1, **i32)
                self.T_target = torch.full((n,), float('nan'), **f64)
            else:
                self.dist = self.step_target = self.T_target = None
            init_slope = init_mult = None
            if init is not None:
                init_slope = torch.as_tensor(np.ascontiguousarray(init[0], dtype=np.float64)).to(dev)
                init_mult = torch.as_tensor(np.ascontiguousarray(init[1], dtype=np.float64)).to(dev)
            self._init = (init_slope, init_mult)
            src = (C.c_double * 2)(float(self.source[0]), float(self.source[1]))

            def full_out():
                out = _lib.Rt2dRayOut()
                out.n_pts, out.stop_code = addr['n_pts'], addr['stop_code']
                out.n_rebounds, out.status = addr['n_rebounds'], addr['status']
                # the bounding box is all the heat-map pipeline needs of the post-pass; L_end / T_end / margin are
                # produced on demand (_lazy_scalars) so that the main cast does not carry them through every step
                out.range = addr['range']
                out.monotone = addr['monotone']
                out.slope_max = addr['slope_max']
                if self.target is not None:
                    out.dist_to_target, out.step_target = self.dist.data_ptr(), self.step_target.data_ptr()
                    out.T_target = self.T_target.data_ptr()
                return out

            def alloc_log(out, cap):
                # the rebound log: entry k of ray r at [k, r]; entries beyond a ray's rebound count are unspecified (never
                # read: no fills -- 800 MB of them for a 1M-ray fan); one allocation
                self.rebound_cap = cap
                m = cap * n
                m8 = (m + 7) // 8 * 8
                lb = torch.empty((4 * m8 + m8 // 2 + m8 // 8) * 8, dtype=torch.uint8, device=dev)

                lp = lb.data_ptr()
                for name, first, dtype in (('reb_velocity', 0, torch.float64), ('reb_angle', m8, torch.float64),
                                           ('reb_pos_x', 2 * m8, torch.float64), ('reb_pos_z', 3 * m8, torch.float64),
                                           ('reb_step', 4 * m8, torch.int32), ('reb_surface', 4 * m8 + m8 // 2, torch.uint8)):
                    carve(name, first, m, dtype, (cap, n), lb, lp)
                out.reb_step, out.reb_surface = addr['reb_step'], addr['reb_surface']
                out.reb_velocity, out.reb_angle = addr['reb_velocity'], addr['reb_angle']
                out.reb_pos_x, out.reb_pos_z = addr['reb_pos_x'], addr['reb_pos_z']

            def cast_write(out, cap, tile_off, X, Z, Cpt):
                p = self._params(cap)
                _lib.check(L.rt2d_cast_write(handle, self.angles_dev.data_ptr(), n, src, C.byref(p), _ptr(init_slope),
                                             _ptr(init_mult), C.byref(out), tile_off.data_ptr(), X.data_ptr(), Z.data_ptr(),
                                             _ptr(Cpt), stream))

            self.rebound_cap = 0
            self.tile_off = self.X = self.Z = self.Cpt = None
            self.total = 0
            self.padded = False
            self.passes = 'count'
            if n == 0 or alloc_only:
                make_views()
                return
            if not self.store:
                make_views()
                # scalars only: one pass with the post-pass folded in.  With a bounded rebound count the rebound log is kept
                # too (37 B per rebound): the fused heat-map kernel reads its rebound losses from it (heatmap_accumulate_fused)
                out = full_out()
                nrm = int(self.kw['n_rebounds_max'])
                if nrm > -1 and kwargs_keep_log:
                    alloc_log(out, nrm + 1)
                p = self._params(self.rebound_cap)
                _lib.check(L.rt2d_cast_count(handle, self.angles_dev.data_ptr(), n, src, C.byref(p), _ptr(init_slope),
                                             _ptr(init_mult), C.byref(out), stream))
                make_views()
                self._out = out
                return
            n_tiles = (n + TILE - 1) // TILE
            self.tile_off = torch.empty(n_tiles + 1, dtype=torch.int64, device=dev)
            total = C.c_int64(0)

            counted = False
            # ONE storing pass into a provisional layout with the same room for every tile (all of n_steps_max + 1 points
            # when that is small, else SPECULATIVE_ROOM points per ray), which is then KEPT as the fan's layout: 288 GB of
            # HBM make untouched room free, and the kernels that read trajectories never look past a ray's points.  Whether
            # every ray fitted (and no rebound log overflowed) is known on the device right away (rt2d_fan_summary, queued
            # behind the cast) but only READ when the host first needs a number of this fan (validate()): gain and heat
            # map can be queued behind the cast without a round trip.  A fan that did not fit is cast again by the
            # count -> layout -> store passes below.
            room = int(self.kw['n_steps_max']) + 1
            if n_tiles * room * TILE * 16 > SPECULATIVE_BYTES:
                room = min(room, SPECULATIVE_ROOM)
            nrm = int(self.kw['n_rebounds_max'])
            layout_bytes = n_tiles * room * TILE * (24 if store_c else 16)
            fits = layout_bytes <= SPECULATIVE_BYTES or (layout_bytes <= PADDED_BYTES and
                                                         layout_bytes * 1.6 <= _free_bytes(dev))
            if known_n_pts is not None:
                make_views()
                self.n_pts.copy_(known_n_pts)        # point counts from an earlier cast of these rays: no counting pass
                counted = True
            elif fits and not self._force_exact:
                out = full_out()
                alloc_log(out, nrm + 1 if nrm > -1 else REBOUND_LOG_CAP_DEFAULT)
                key = (dev.index, n_tiles, room)
                off0 = _UNIFORM_LAYOUTS.get(key)
                if off0 is None:
                    off0 = torch.arange(n_tiles + 1, dtype=torch.int64, device=dev) * (room * TILE)
                    if len(_UNIFORM_LAYOUTS) < 64:
                        _UNIFORM_LAYOUTS[key] = off0
                X0 = torch.empty(n_tiles * room * TILE, **f64)
                Z0 = torch.empty(n_tiles * room * TILE, **f64)
                C0 = torch.empty(n_tiles * room * TILE, **f64) if store_c else None
                cast_write(out, self.rebound_cap, off0, X0, Z0, C0)
                make_views()                           # (the cast is running)
                exact = torch.empty(n_tiles + 1, dtype=torch.int64, device=dev)
                if n_tiles <= 256:                   # (one workgroup does it all: 9 us for 999 rays, 107 us for 65 536)
                    # small fan: exact layout, summary and the mask of the error-free rays in ONE launch
                    self._summary_dev = torch.empty(SUMMARY_LEN, dtype=torch.float64, device=dev)
                    self._ok_mask = torch.empty(n, dtype=torch.uint8, device=dev)
                    _lib.check(L.rt2d_layout_summary(self.n_pts.data_ptr(), self.status.data_ptr(), self.range.data_ptr(), n,
                                                     exact.data_ptr(), self._summary_dev.data_ptr(), SUMMARY_LEN,
                                                     self._ok_mask.data_ptr(), stream))
                else:
                    _lib.check(L.rt2d_traj_layout(self.n_pts.data_ptr(), n, exact.data_ptr(), None, stream))
                    self._queue_summary(exact, stream)
                self.tile_off, self._tile_off_exact = off0, exact
                self.X, self.Z, self.Cpt = X0, Z0, C0
                self.total = n_tiles * room * TILE
                self._room = room
                self._out = out
                self.padded = True
                self.passes = 'padded' if layout_bytes > SPECULATIVE_BYTES else 'speculative'
                self._pending = True
                return

            # pass 1: count points and rebounds only (no post-pass, no stores) -- unless a single-pass attempt above
            # already did: it stops storing at its room but steps and counts to the end
            make_views()
            if not counted:
                cnt = _lib.Rt2dRayOut()
                cnt.n_pts, cnt.n_rebounds = self.n_pts.data_ptr(), self.n_rebounds.data_ptr()
                p = self._params(0)
                _lib.check(L.rt2d_cast_count(handle, self.angles_dev.data_ptr(), n, src, C.byref(p), _ptr(init_slope),
                                             _ptr(init_mult), C.byref(cnt), stream))
            _lib.check(L.rt2d_traj_layout(self.n_pts.data_ptr(), n, self.tile_off.data_ptr(), C.byref(total), stream))
            self.total = int(total.value)
            # pass 2: store trajectories, the rebound log and the post-pass scalars
            out = full_out()
            if known_n_pts is not None:
                # the rebound counts of these rays are known too (subset()): the log is sized by them, so it cannot
                # overflow -- nothing behind this pass would act on an overflow
                nrm_k = int(self.kw['n_rebounds_max'])
                if known_reb_max is not None:
                    alloc_log(out, max(1, int(known_reb_max)))
                else:
                    alloc_log(out, nrm_k + 1 if nrm_k > -1 else REBOUND_LOG_CAP_DEFAULT)
            else:
                alloc_log(out, max(1, int(self.n_rebounds.max().item())))
            self.X = torch.empty(self.total, **f64)
            self.Z = torch.empty(self.total, **f64)
            self.Cpt = torch.empty(self.total, **f64) if store_c else None
            cast_write(out, self.rebound_cap, self.tile_off, self.X, self.Z, self.Cpt)
            make_views()
            self._out = out
            self.passes = 'count+store'
            self._queue_summary(self.tile_off, stream)
            self._summary = self._summary_dev.tolist()

    def _queue_summary(self, tile_off_exact, stream):
        """rt2d_fan_summary of the finished cast, queued behind it; read by validate()."""
        torch = _torch()
        # (a ninth word behind the summary: "the one-pass heat map ran out of rows for its side rays", rt2d_classify_rays)
        self._summary_dev = torch.zeros(SUMMARY_LEN, dtype=torch.float64, device=self.device)
        _lib.check(_lib.lib().rt2d_fan_summary(self.status.data_ptr(), self.range.data_ptr(), None, self.n,
                                               tile_off_exact.data_ptr(), self._summary_dev.data_ptr(), stream))

    def validate(self):
        """The one device -> host read of a single-pass cast (64 bytes): [points of the exact layout, longest tile, a
        rebound log overflowed, rays without error, plot range x 4].  If a ray outgrew the room of the provisional layout
        or a rebound log overflowed, the fan is cast again with counted sizes (what was derived from it is dropped; gains that
        had been populated are populated again).
        Returns True when the fan had to be cast again."""
        if not self._pending:
            return False
        self._pending = False
        self._summary = self._summary_dev.tolist()
        self._onepass_overflow = bool(self._summary[8])
        longest, log_full = int(self._summary[1]), bool(self._summary[2])
        if longest <= self._room * TILE and not log_full:
            return False
        self._force_exact = True
        had_arrays, had_ends = list(self.G), [f for f in self.G_end if f not in self.G]
        self.G, self.G_end, self._post, self._host, self._scalars, self._ok_mask = {}, {}, None, {}, None, None
        self._cast(self._init_host, self.Cpt is not None)
        # gains that were queued behind the provisional cast belonged to it: the same frequencies again, on the final one
        # (a caller that holds on to fan.G[f] across validate() keeps finding it, now with every ray's full log behind it)
        if had_arrays:
            self.populate(*had_arrays)
        if had_ends:
            self.populate(*had_ends, keep_arrays=False)
        return True

    # -- scalars produced on demand ---------------------------------------------------------------
    def _lazy_scalars(self):
        """L_end, T_end (ray.py:272-275, last elements) and the decision margin: one more pass of the stepping kernel
        without stores.  The kernel is deterministic, so these belong to exactly the rays of the main cast."""
        if self._scalars is None:
            torch = _torch()
            with torch.cuda.device(self.device):
                f64 = dict(dtype=torch.float64, device=self.device)
                d = {k: torch.empty(self.n, **f64) for k in ('L_end', 'T_end', 'margin')}
                if self.n:
                    out = _lib.Rt2dRayOut()
                    out.L_end, out.T_end, out.margin = d['L_end'].data_ptr(), d['T_end'].data_ptr(), d['margin'].data_ptr()
                    src = (C.c_double * 2)(float(self.source[0]), float(self.source[1]))
                    p = self._params(0)
                    _lib.check(_lib.lib().rt2d_cast_count(self.env.device_handle(self.device.index), self.angles_dev.data_ptr(),
                                                         self.n, src, C.byref(p), _ptr(self._init[0]), _ptr(self._init[1]),
                                                         C.byref(out), _raw_stream()))
            self._scalars = d
        return self._scalars

    @property
    def L_end(self):
        return self._lazy_scalars()['L_end']

    @property
    def T_end(self):
        return self._lazy_scalars()['T_end']

    @property
    def margin(self):
        return self._lazy_scalars()['margin']

    # -- host copies ------------------------------------------------------------------------------
    def host(self, name):
        """NumPy copy of a per-ray tensor (cached)."""
        self.validate()
        h = self._host.get(name)
        if h is None:
            h = getattr(self, name).cpu().numpy()
            self._host[name] = h
        return h

    @property
    def angles(self):
        if self._angles_host is None:
            self._angles_host = self.angles_dev.cpu().numpy()
        return self._angles_host

    def ray_steps(self):
        """Total ray-steps of the fan = appended points, excluding the sources."""
        return int((self.n_pts.sum() - self.n).item())

    def _need_store(self):
        if not self.store:
            if not getattr(self, '_lazy_store', False):
                raise _lib.Rt2dError('this fan was cast with store=False: trajectories were not kept')
            # a subset of a scalars-only fan (subset()): its trajectories are cast now, in ONE storing pass -- the point
            # counts are already known, so the exact layout can be built before the pass
            known = self.n_pts
            reb_max = int(self.n_rebounds.max().item()) if self.n else 0
            self.store, self._lazy_store = True, False
            self._host = {}
            self._cast(self._init_host, False, known_n_pts=known, known_reb_max=reb_max)
            if self._summary is not None and bool(self._summary[2]):
                raise _lib.Rt2dError('internal: a rebound log sized by the known rebound counts overflowed')

    def subset(self, indices):
        """The rays `indices` of a fan cast with store=False as a fan of their own: per-ray scalars are copied, nothing is
        cast; the trajectories are cast (once, with the known point counts) if anything asks for them.  The stepping
        kernel is deterministic, so a ray cast again is the same ray."""
        torch = _torch()
        if self.store:
            raise _lib.Rt2dError('subset() is for fans cast with store=False')
        if hasattr(indices, 'data_ptr'):                      # a device tensor is used where it is
            idx = indices.to(device=self.device, dtype=torch.int64)
            if self._angles_host is not None or self._init_host is not None:
                indices = idx.cpu().numpy()
        else:
            idx = torch.as_tensor(np.asarray(indices, dtype=np.int64), device=self.device)
        sub = object.__new__(RayFan)
        sub.env, sub.source, sub.target, sub.device, sub.kw = self.env, self.source, self.target, self.device, dict(self.kw)
        sub.store, sub.keep_log, sub._lazy_store = False, False, True
        sub.angles_dev = self.angles_dev[idx].contiguous()
        sub._angles_host = None if self._angles_host is None else self._angles_host[np.asarray(indices, dtype=np.int64)]
        sub.n = int(idx.numel())
        sub._host, sub.G, sub.G_end, sub._post = {}, {}, {}, None
        sub._pending, sub._force_exact, sub._summary, sub._scalars, sub._uncast = False, False, None, None, False
        init = None
        if self._init_host is not None:
            sel = np.asarray(indices, dtype=np.int64)
            init = (np.asarray(self._init_host[0])[sel], np.asarray(self._init_host[1])[sel])
        sub._init_host = init
        sub._init = (None, None) if self._init[0] is None else (self._init[0][idx], self._init[1][idx])
        for name in ('n_pts', 'stop_code', 'n_rebounds', 'status', 'monotone', 'slope_max', 'dist', 'step_target', 'T_target'):
            t = getattr(self, name)
            setattr(sub, name, None if t is None else t[idx].contiguous())
        sub.range = self.range[:, idx].contiguous()
        sub.rebound_cap, sub.total, sub.padded, sub.passes = 0, 0, False, 'subset'
        sub.tile_off = sub.X = sub.Z = sub.Cpt = None
        return sub

    def _ensure_compact(self):
        """Turns a padded layout (see PADDED_BYTES) into the exact one: X, Z, C and the gains kept so far are copied."""
        self.validate()
        if not self.padded:
            return
        torch = _torch()
        L = _lib.lib()
        with torch.cuda.device(self.device):
            stream = _raw_stream()
            f64 = dict(dtype=torch.float64, device=self.device)
            exact = self._tile_off_exact
            total = int(self._summary[0]) if self._summary is not None else int(exact[-1].item())

            def move(a, b, c=None):
                na, nb = torch.empty(total, **f64), torch.empty(total, **f64)
                nc = torch.empty(total, **f64) if c is not None else None
                _lib.check(L.rt2d_traj_compact(self.tile_off.data_ptr(), exact.data_ptr(), self.n, a.data_ptr(), b.data_ptr(),
                                               _ptr(c), na.data_ptr(), nb.data_ptr(), _ptr(nc), stream))
                return na, nb, nc

            X, Z, Cp = move(self.X, self.Z, self.Cpt)
            freqs = list(self.G)
            newG = {}
            for k in range(0, len(freqs), 2):
                pair = freqs[k:k + 2]
                a = self.G[pair[0]]
                b = self.G[pair[-1]]
                na, nb, _ = move(a, b)
                newG[pair[0]] = na
                if len(pair) == 2:
                    newG[pair[1]] = nb
            self.X, self.Z, self.Cpt, self.G = X, Z, Cp, newG
            self.tile_off, self.total, self.padded = exact, total, False
            self._host.pop('tile_off', None)
            self.passes = self.passes + '+compact'

    def ray_slice(self, tensor, i):
        """Points of ray i out of a tile-layout tensor (a strided device view, no copy)."""
        self._need_store()
        self.validate()
        base = int(self.host('tile_off')[i // TILE]) + (i % TILE)
        return tensor[base::TILE][:int(self.host('n_pts')[i])]

    def padded_host(self, tensor_x=None, tensor_z=None):
        """(X, Z, valid) as (n, max_pts) NumPy arrays, NaN-padded: the whole fan's points on the host."""
        self._need_store()
        self._ensure_compact()
        X = self.X if tensor_x is None else tensor_x
        Z = self.Z if tensor_z is None else tensor_z
        n_pts, off = self.host('n_pts'), self.host('tile_off')
        m = int(n_pts.max()) if self.n else 0
        outx = np.full((self.n, m), np.nan)
        outz = np.full((self.n, m), np.nan)
        for t in range(len(off) - 1):
            steps = int(off[t + 1] - off[t]) // TILE
            lanes = min(TILE, self.n - t * TILE)
            bx = X[int(off[t]):int(off[t + 1])].view(steps, TILE).cpu().numpy()
            bz = Z[int(off[t]):int(off[t + 1])].view(steps, TILE).cpu().numpy()
            outx[t * TILE:t * TILE + lanes, :steps] = bx.T[:lanes]
            outz[t * TILE:t * TILE + lanes, :steps] = bz.T[:lanes]
        valid = np.arange(m)[None, :] < n_pts[:, None]
        outx[~valid] = np.nan
        outz[~valid] = np.nan
        return outx, outz, valid

    # -- post-pass arrays (ray.py:271-275) --------------------------------------------------------
    def post(self):
        if self._post is None:
            self._need_store()
            self._ensure_compact()               # five more arrays of `total` elements: not in the padded layout
            torch = _torch()
            L = _lib.lib()
            with torch.cuda.device(self.device):
                f64 = dict(dtype=torch.float64, device=self.device)
                d = {k: torch.empty(self.total, **f64) for k in ('dL', 'L', 'C', 'dT', 'T')}
                _lib.check(L.rt2d_post(self.env.device_handle(self.device.index), self.X.data_ptr(), self.Z.data_ptr(),
                                       self.tile_off.data_ptr(), self.n_pts.data_ptr(), self.n, d['dL'].data_ptr(),
                                       d['L'].data_ptr(), d['C'].data_ptr(), d['dT'].data_ptr(), d['T'].data_ptr(),
                                       _raw_stream()))
            self._post = d
        return self._post

    # -- gain (ray.py:339-369) --------------------------------------------------------------------
    def populate(self, *freqs, keep_arrays=True):
        """Cumulative gain [dB] for every ray at the given frequencies (new ones only)."""
        self._need_store()
        torch = _torch()
        # a frequency whose end gains are known but whose array was not kept (keep_arrays=False) is new again when the
        # arrays are asked for
        new = [f for f in dict.fromkeys(freqs) if f not in self.G_end or (keep_arrays and f not in self.G)]
        if not new:
            return
        if self.padded and keep_arrays and len(new) + len(self.G) > 1 and (len(new) + len(self.G)) * self.total * 8 > SPECULATIVE_BYTES:
            self._ensure_compact()               # a big padded layout only has room for one gain array beside X and Z
        L = _lib.lib()
        with torch.cuda.device(self.device):
            f64 = dict(dtype=torch.float64, device=self.device)
            fr = _freq_tensor(new, self.device)
            G = torch.empty((len(new), self.total), **f64) if keep_arrays else None
            G_end = torch.empty((len(new), self.n), **f64)
            # workspace for the rebound losses (their own kernel, see rt2d_gain) while it stays modest
            ws_elems = len(new) * self.rebound_cap * self.n
            loss_ws = torch.empty(ws_elems, **f64) if 0 < ws_elems * 8 <= LOSS_WS_BYTES else None
            _lib.check(L.rt2d_gain(self.env.device_handle(self.device.index), self.X.data_ptr(), self.Z.data_ptr(),
                                   self.tile_off.data_ptr(), self.n_pts.data_ptr(), C.byref(self._out),
                                   self.rebound_cap, self.n, fr.data_ptr(), len(new), self.total, _ptr(G),
                                   G_end.data_ptr(), _ptr(loss_ws), _raw_stream()))
        for k, f in enumerate(new):
            if G is not None:
                self.G[f] = G[k]
            self.G_end[f] = G_end[k]

    # -- distance to another target (ray.py:282-287) ----------------------------------------------
    def dist_to(self, target):
        self._need_store()
        torch = _torch()
        L = _lib.lib()
        with torch.cuda.device(self.device):
            out = torch.empty(self.n, dtype=torch.float64, device=self.device)
            tgt = (C.c_double * 2)(float(target[0]), float(target[1]))
            _lib.check(L.rt2d_dist(self.X.data_ptr(), self.Z.data_ptr(), self.tile_off.data_ptr(), self.n_pts.data_ptr(),
                                   self.n, tgt, out.data_ptr(), _raw_stream()))
        return out

    # -- heat map (simulation.py:120-131) ---------------------------------------------------------
    def heatmap_mask(self, res, shape, ray_mask=None):
        """ray_mask for rt2d_heatmap_accumulate with the hint bit (value 2) set for the rays whose x never decreases and
        whose bounding box lies inside the map: the tile kernel need not check those point by point."""
        torch = _torch()
        # (one launch: int(x) >= 0, int(z) <= 0, last column and row inside the map, x never decreases -- a dozen torch
        # kernels of 5 us each when written with tensor operations)
        out = torch.empty(self.n, dtype=torch.uint8, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().rt2d_heatmap_hints(self.monotone.data_ptr(), self.range.data_ptr(), _ptr(ray_mask), self.n,
                                                     (C.c_int32 * 2)(int(res[0]), int(res[1])),
                                                     (C.c_int32 * 2)(int(shape[0]), int(shape[1])), out.data_ptr(), _raw_stream()))
        return out                                          # 1 | 2: vouched for; 1 | 4: known not to qualify

    def fused_mask(self, res, shape, ray_mask=None):
        """uint8 mask of the rays rt2d_cast_heatmap may bin (include/rt2d.h): no error, x never decreases, bounding box
        inside the map, at most 8 rows per column by the slope bound."""
        torch = _torch()
        rng = self.range
        ok = ((self.status == 0) & (self.monotone != 0) & (rng[0] > -1) & (rng[3] < 1)
              & (torch.div(torch.trunc(rng[2]), int(res[0]), rounding_mode='floor') < int(shape[0]))
              & (torch.div(-torch.trunc(rng[1]), int(res[1]), rounding_mode='floor') < int(shape[1]))
              & (torch.floor(self.slope_max * (float(res[0]) / float(res[1]))) + 2 <= 8))
        if ray_mask is not None:
            ok = ok & (ray_mask != 0)
        return ok.to(torch.uint8)

    def heatmap_accumulate_fused(self, H, res, shape, freq=1, ray_mask=None, dropped=None):
        """Heat map of a fan cast WITHOUT trajectories (store=False): the rays are propagated again by the fused kernel
        (cast -> gain -> binning, nothing stored), except those it cannot bin exactly (fused_mask), which are cast with
        trajectories as a small side fan and binned by the general kernels.  Same map as heatmap_accumulate."""
        torch = _torch()
        L = _lib.lib()
        with torch.cuda.device(self.device):
            sel = (self.status == 0) if ray_mask is None else ((self.status == 0) & (ray_mask != 0))
            mask = self.fused_mask(res, shape, ray_mask)
            # (an earlier pass met rays its binner could not take, and flagged them: those go the stored way this time -- the
            # propagation is deterministic, so the others pass again as they did)
            flagged = self.__dict__.get('_irregular')
            if self.__dict__.get('_exact_binning'):
                mask = mask & (flagged == 0).to(torch.uint8) if flagged is not None else torch.zeros_like(mask)
            irr = torch.zeros(self.n, dtype=torch.uint8, device=self.device)
            res_c = (C.c_int32 * 2)(int(res[0]), int(res[1]))
            shape_c = (C.c_int32 * 2)(int(shape[0]), int(shape[1]))
            src = (C.c_double * 2)(float(self.source[0]), float(self.source[1]))
            p = self._params(0)
            viol = torch.zeros(1, dtype=torch.int64, device=self.device)
            handle = self.env.device_handle(self.device.index)
            stream = _raw_stream()
            loss, cap = None, 0
            if self.rebound_cap > 0 and self.rebound_cap * self.n * 8 <= LOSS_WS_BYTES:
                # the rebound losses at this frequency from the count pass's log, by their own kernel (rt2d_rebound_loss)
                cap = self.rebound_cap
                loss = torch.empty(cap * self.n, dtype=torch.float64, device=self.device)
                fr = _freq_tensor([freq], self.device)
                _lib.check(L.rt2d_rebound_loss(handle, C.byref(self._out), cap, self.n, fr.data_ptr(), 1, loss.data_ptr(), stream))
            # (the rays the fused kernel cannot take are known before it starts: they go through the stored path on a
            # second stream while it runs)
            rest = torch.nonzero(sel & (mask == 0)).flatten()
            before = torch.cuda.current_stream().record_event()
            rc = L.rt2d_cast_heatmap(handle, self.angles_dev.data_ptr(), self.n, src,
                                     C.byref(p), _ptr(self._init[0]), _ptr(self._init[1]), mask.data_ptr(), float(freq),
                                     res_c, shape_c, 1.0, H.data_ptr(), None, irr.data_ptr(), viol.data_ptr(), _ptr(loss), cap, stream)
            self._irregular = irr if flagged is None else (irr | flagged)
            if rc != 0:
                # the fused kernel cannot take this environment (its tables do not fit in LDS) or these parameters
                # (power_cutoff): every ray goes through the stored path, a chunk at a time
                self.fused_unavailable = L.rt2d_last_error().decode()
                rest = torch.nonzero(sel).flatten()
            self.last_fused_violations = self._viol_pending = viol
            self._bin_side_rays(rest, H, res, shape, freq, dropped, fast=rc != 0, after=before)

    def _bin_side_rays(self, rest, H, res, shape, freq, dropped, fast=False, after=None):
        """Adds the rays `rest` (device index tensor) to H through the stored path -- cast with trajectories as side
        fans of at most SIDE_FAN_RAYS rays, gain, general binner -- on a second stream, so that their few long waves
        run beside the kernel that bins everybody else; the calling stream waits for them at the end.  after: an event
        recorded on the calling stream BEFORE that kernel was queued (the side stream starts from there, not behind it)."""
        torch = _torch()
        self.last_slow_rays = int(rest.numel())
        if self.last_slow_rays == 0:
            return
        main = torch.cuda.current_stream()
        side_stream = _side_stream(self.device) if SIDE_STREAM else main
        if side_stream is main:
            pass
        elif after is not None:
            side_stream.wait_event(after)                # H is zeroed, the angles are there
        else:
            side_stream.wait_stream(main)
        # rays longer than the room of a provisional layout (default step sizes: 100 001 points): this fan holds their
        # point and rebound counts, so the side fan is cast ONCE, straight into its exact layout (subset()), instead of
        # provisional pass -> overflow -> count -> store
        long_rays = int(self.kw['n_steps_max']) + 1 > SPECULATIVE_ROOM and not self.__dict__.get('_uncast')
        with torch.cuda.stream(side_stream):
            for c0 in range(0, int(rest.numel()), SIDE_FAN_RAYS):
                part = rest[c0:c0 + SIDE_FAN_RAYS]
                if long_rays:
                    side = self.subset(part)
                    side.target, side.dist, side.step_target, side.T_target = None, None, None, None
                    if after is not None and side_stream is not main:
                        side.launch_hint = 1             # RT2D_LAUNCH_SMALL_LDS: beside the fused kernel, whose workgroups fill the LDS
                    side._need_store()
                else:
                    init = None if self._init[0] is None else (self._init[0][part].cpu().numpy(), self._init[1][part].cpu().numpy())
                    side = RayFan(self.env, self.source, self.angles_dev[part], target=None, store=True, device=self.device,
                                  init=init, **dict(self.kw))
                    side.validate()                          # (a ray that outgrew the provisional room: cast again, now)
                side.heatmap_accumulate(H, res, shape, freq=freq, dropped=dropped, fast=fast, stream=False)
                del side
        if side_stream is not main:
            main.wait_stream(side_stream)

    def _chunk_rays(self):
        """Rays per chunk of the chunked heat map: CHUNK_RAYS when set, else what a third of the free HBM (and at most
        PADDED_BYTES) holds of provisional trajectories (16 B per point of room) -- a fan that fits is ONE chunk: the
        memory that is there is used, a fan of any size still runs."""
        if CHUNK_RAYS > 0:
            return CHUNK_RAYS
        torch = _torch()
        room = min(int(self.kw['n_steps_max']) + 1, SPECULATIVE_ROOM)
        budget = min(PADDED_BYTES, _free_bytes(self.device) // 3)
        if not self.__dict__.get('_uncast') and self.n and self.__dict__.get('n_pts') is not None:
            # a fan that was counted before: its chunks go straight into exact layouts (a tile = its longest ray), plus a
            # gain array per chunk below STREAM_MIN_RAYS rays -- at the reference's default step sizes a ray has 100 001
            # points, 2.4 MB of trajectory and gain
            longest = max(1, int(self.n_pts.max().item()))
            chunk = max(TILE, int(budget // (longest * 24)) // TILE * TILE)
            pieces = (self.n + chunk - 1) // chunk
            return max(TILE, ((self.n + pieces - 1) // pieces + TILE - 1) // TILE * TILE)
        chunk = max(1 << 16, int(budget // (room * 16)) >> 16 << 16)
        pieces = (self.n + chunk - 1) // chunk                     # equal pieces rather than full chunks and a small rest
        return max(TILE, ((self.n + pieces - 1) // pieces + TILE - 1) // TILE * TILE)

    def heatmap_accumulate_chunked(self, H, res, shape, freq=1, ray_mask=None, dropped=None):
        """Heat map of a fan that keeps no trajectories (store=False), CHUNK_RAYS rays at a time: the chunk is cast WITH
        trajectories, binned by the stored pipeline, and dropped again -- the trajectories in memory are never more than
        one chunk's.  A lazy fan that has not been cast yet gets its per-ray results from those casts (single storing pass
        into a provisional layout each): no counting pass runs at all.  A fan that was counted before casts every chunk
        straight into its exact layout.  Returns False when a chunk outgrew its provisional layout: H is then incomplete,
        the fan has been cast (counted) the ordinary way, and the caller starts over with a zeroed map.  Same map as
        heatmap_accumulate of the same fan cast with store=True."""
        torch = _torch()
        L = _lib.lib()
        first = bool(self.__dict__.get('_uncast'))
        n = self.n
        with torch.cuda.device(self.device):
            chunk = self._chunk_rays()
            adopt = first and n <= chunk              # one chunk: its per-ray results ARE the fan's (no copies, no second summary)
            if first:
                self._uncast = False
                if not adopt:
                    self._cast(self._init_host, False, alloc_only=True)
            checks, slow, slows, viols = [], 0, [], []
            # chunks alternate between two streams, each starting when the previous chunk's CAST is done: the long last
            # waves of one chunk's kernels (a chunk is one round of waves: its kernels last as long as its longest ray)
            # run beside the next chunk's
            main = torch.cuda.current_stream()
            lanes = [main] + ([_side_stream(self.device, w) for w in range(CHUNK_STREAMS - 1)] if SIDE_STREAM and n > chunk else [])
            cast_done = None
            for k, c0 in enumerate(range(0, n, chunk)):
              lane_stream = lanes[k % len(lanes)]
              if cast_done is not None and lane_stream is not main:
                  lane_stream.wait_event(cast_done)
              with torch.cuda.stream(lane_stream):
                c1 = min(n, c0 + chunk)
                if first:
                    init = None if self._init_host is None else (np.asarray(self._init_host[0])[c0:c1], np.asarray(self._init_host[1])[c0:c1])
                    sub = RayFan(self.env, self.source, self.angles_dev[c0:c1], target=self.target, store=True, device=self.device,
                                 init=init, **dict(self.kw))
                    if adopt:
                        for name in ('n_pts', 'stop_code', 'n_rebounds', 'status', 'monotone', 'slope_max', 'dist', 'step_target',
                                     'T_target', 'range', '_init'):
                            setattr(self, name, getattr(sub, name))
                        self._scalars, self.rebound_cap, self.total, self.padded = None, 0, 0, False
                        self.tile_off = self.X = self.Z = self.Cpt = None
                        if not sub._pending:
                            sub.validate()
                        self._summary_dev = sub._summary_dev
                    else:
                        for name in ('n_pts', 'stop_code', 'n_rebounds', 'status', 'monotone', 'slope_max', 'dist', 'step_target', 'T_target'):
                            t = getattr(self, name)
                            if t is not None:
                                t[c0:c1].copy_(getattr(sub, name))
                        self.range[:, c0:c1].copy_(sub.range)
                    if sub._pending:
                        checks.append((sub._summary_dev, sub._room))
                else:
                    sub = self.subset(np.arange(c0, c1))
                    sub._need_store()
                if self.__dict__.get('_exact_binning'):
                    sub._exact_binning = True
                cast_done = lane_stream.record_event()
                mask = (sub.status == 0) if ray_mask is None else ((sub.status == 0) & (ray_mask[c0:c1] != 0))
                # (a chunk's summary -- did it fit its provisional layout, did a one-pass map have rows enough for its side
                # rays -- is checked below for all chunks at once: nothing here waits for the device)
                sub.heatmap_accumulate(H, res, shape, freq=freq, ray_mask=mask.to(torch.uint8), dropped=dropped,
                                       defer=first and sub._pending)
                sub._pending = False
                if sub.__dict__.get('_viol_pending') is not None:
                    viols.append(sub._viol_pending)
                w = getattr(sub, '_last_slow', 0)           # an int, or the work list of rt2d_heatmap_accumulate ([0] < 0: every ray)
                slows.append(w if isinstance(w, int) else torch.where(w[0] < 0, torch.full_like(w[0], sub.n), w[0]))
                del sub
            for lane_stream in lanes[1:]:
                main.wait_stream(lane_stream)
            for w in slows:
                slow = slow + w
            self.last_fused_violations = torch.zeros(1, dtype=torch.int64, device=self.device)
            for v in viols:
                self.last_fused_violations = self.last_fused_violations + v
            if viols:
                self._viol_pending = self.last_fused_violations
            self._last_slow = slow if isinstance(slow, int) else slow.reshape(1)
            if first:
                if not adopt:
                    stream = _raw_stream()
                    n_tiles = (n + TILE - 1) // TILE
                    exact = torch.empty(n_tiles + 1, dtype=torch.int64, device=self.device)
                    _lib.check(L.rt2d_traj_layout(self.n_pts.data_ptr(), n, exact.data_ptr(), None, stream))
                    self._queue_summary(exact, stream)
                self.passes = 'chunked'
                # -- the one read: every chunk's summary and the fan's own --
                got = torch.cat([c for c, _ in checks] + [self._summary_dev]).tolist()
                for k, (_, room) in enumerate(checks):
                    if int(got[SUMMARY_LEN * k + 1]) > room * TILE or bool(got[SUMMARY_LEN * k + 2]) or bool(got[SUMMARY_LEN * k + 8]):
                        self._cast(self._init_host, False)          # cast the ordinary way; the caller starts over
                        return False
                self._summary = got[-SUMMARY_LEN:]
        return True

    def heatmap_accumulate_stream(self, H, res, shape, freq=1, ray_mask=None, dropped=None, defer=False):
        """Gain and binning in ONE forward pass over the stored trajectories (rt2d_gain_heatmap_rest): no gain array of the
        layout's size exists.  The rays that pass cannot bin (fused_mask: turned back, left the map, too steep for its
        eight row slots) only get their gain from it, into one row each of a compact array, and the general kernel bins
        them from there.  Same map as populate(freq) + heatmap_accumulate."""
        self._need_store()
        torch = _torch()
        L = _lib.lib()
        with torch.cuda.device(self.device):
            if defer and self._pending and self.padded:
                # Queued behind a single-pass cast whose summary has not been read: the host knows neither the point
                # counts nor which rays go to the general kernel.  rt2d_classify_rays sorts the rays out on the device and
                # deals the side rays rows of a compact gain array sized by bounds (a row = the room of the provisional
                # layout, n / 8 rows); "ran out of rows" lands in the word behind the fan's summary, and the caller --
                # who reads the summary anyway (validate) -- then bins the fan again with known sizes (defer=False).
                stream = _raw_stream()
                rows, g_room = max(ONEPASS_MIN_ROWS, self.n // ONEPASS_ROWS_DIV), int(self._room)
                m = torch.empty(self.n, dtype=torch.uint8, device=self.device)
                slot = torch.empty(self.n, dtype=torch.int32, device=self.device)
                rest_list = torch.empty(self.n + 2, dtype=torch.int32, device=self.device)
                G_rest = torch.empty(rows * g_room, dtype=torch.float64, device=self.device)
                viol = torch.zeros(1, dtype=torch.int64, device=self.device)
                res_c = (C.c_int32 * 2)(int(res[0]), int(res[1]))
                shape_c = (C.c_int32 * 2)(int(shape[0]), int(shape[1]))
                _lib.check(L.rt2d_classify_rays(self.status.data_ptr(), self.monotone.data_ptr(), self.range.data_ptr(),
                                                self.slope_max.data_ptr(), _ptr(ray_mask), self.n, res_c, shape_c, rows,
                                                m.data_ptr(), slot.data_ptr(), rest_list.data_ptr(),
                                                self._summary_dev[8:].data_ptr(), stream))
                ws_elems = self.rebound_cap * self.n
                loss_ws = torch.empty(ws_elems, dtype=torch.float64, device=self.device) if 0 < ws_elems * 8 <= LOSS_WS_BYTES else None
                _lib.check(L.rt2d_gain_heatmap_rest(self.env.device_handle(self.device.index), self.X.data_ptr(), self.Z.data_ptr(),
                                                    self.tile_off.data_ptr(), self.n_pts.data_ptr(), C.byref(self._out), self.rebound_cap,
                                                    self.n, float(freq), m.data_ptr(), res_c, shape_c, H.data_ptr(),
                                                    rest_list.data_ptr(), rows, slot.data_ptr(), g_room, G_rest.data_ptr(),
                                                    _ptr(dropped), viol.data_ptr(), _ptr(loss_ws),
                                                    torch.empty(rows + 2, dtype=torch.int32, device=self.device).data_ptr(), stream))
                self._last_slow = rest_list            # ([0] = the number of side rays: read on demand)
                self.last_fused_violations = self._viol_pending = viol
                return
            self.validate()                              # the masks below need the final cast
            self._onepass_overflow = False
            stream = _raw_stream()
            sel = (self.status == 0) if ray_mask is None else ((self.status == 0) & (ray_mask != 0))
            m = self.fused_mask(res, shape, ray_mask)
            rest = sel & (m == 0)
            # the rays the one pass cannot bin: it computes their gain into one row each of a compact array and the general
            # kernel bins them from there (rt2d_gain_heatmap_rest)
            slot = torch.cumsum(rest, 0, dtype=torch.int32) - 1
            idx = torch.nonzero(rest).flatten().to(torch.int32)
            n_rest = int(idx.numel())
            m = m + 2 * rest.to(torch.uint8)
            g_room = int(self._summary[1]) // TILE if self._summary is not None else int(self.n_pts.max().item())
            G_rest = torch.empty(max(1, n_rest * g_room), dtype=torch.float64, device=self.device)
            rest_list = torch.cat([torch.tensor([n_rest, 0], dtype=torch.int32, device=self.device), idx])
            viol = torch.zeros(1, dtype=torch.int64, device=self.device)
            res_c = (C.c_int32 * 2)(int(res[0]), int(res[1]))
            shape_c = (C.c_int32 * 2)(int(shape[0]), int(shape[1]))
            ws_elems = self.rebound_cap * self.n
            loss_ws = torch.empty(ws_elems, dtype=torch.float64, device=self.device) if 0 < ws_elems * 8 <= LOSS_WS_BYTES else None
            _lib.check(L.rt2d_gain_heatmap_rest(self.env.device_handle(self.device.index), self.X.data_ptr(), self.Z.data_ptr(),
                                                self.tile_off.data_ptr(), self.n_pts.data_ptr(), C.byref(self._out), self.rebound_cap,
                                                self.n, float(freq), m.data_ptr(), res_c, shape_c, H.data_ptr(),
                                                rest_list.data_ptr(), n_rest, slot.data_ptr(), g_room, G_rest.data_ptr(),
                                                _ptr(dropped), viol.data_ptr(), _ptr(loss_ws),
                                                torch.empty(n_rest + 2, dtype=torch.int32, device=self.device).data_ptr(), stream))
            self.last_slow_rays = n_rest
            self.last_fused_violations = self._viol_pending = viol

    def heatmap_accumulate_multi(self, H, res, shape, freqs, ray_mask=None, dropped=None):
        """Adds this fan's rays into the maps of several frequencies at once (H: (F, nx*nz) device f64, contiguous): one
        pass over the trajectories for all of them (rt2d_heatmap_accumulate_multi)."""
        self._need_store()
        torch = _torch()
        L = _lib.lib()
        freqs = list(freqs)
        with torch.cuda.device(self.device):
            res_c = (C.c_int32 * 2)(int(res[0]), int(res[1]))
            shape_c = (C.c_int32 * 2)(int(shape[0]), int(shape[1]))
            work = torch.empty(self.n + 2, dtype=torch.int32, device=self.device)
            if self.n > HEATMAP_FAST_MIN_RAYS:
                ray_mask = self.heatmap_mask(res, shape, ray_mask)

            # scratch of the tile kernel's map-minor deposits (one allocation per call, reused by every group of maps: the
            # groups are binned one after the other on this stream)
            cells = int(shape[0]) * int(shape[1])
            minor = None
            if MULTI_MINOR and len(freqs) > 1:
                minor = torch.empty((len(freqs) + 3) // 4 * 4 * cells, dtype=torch.float64, device=self.device)

            def bin_maps(k0, k1):
                ptrs = torch.tensor([self.G[f].data_ptr() for f in freqs[k0:k1]], dtype=torch.int64, device=self.device)
                use_minor = minor if minor is not None and ((k1 - k0 + 3) // 4 * 4 * cells) <= minor.numel() else None
                _lib.check(L.rt2d_heatmap_accumulate_multi(self.X.data_ptr(), self.Z.data_ptr(), ptrs.data_ptr(), k1 - k0,
                                                           self.tile_off.data_ptr(), self.n_pts.data_ptr(), self.n, _ptr(ray_mask),
                                                           res_c, shape_c, H[k0:k1].data_ptr(), _ptr(dropped), work.data_ptr(),
                                                           _ptr(use_minor), _raw_stream()))

            # Binning is bound by the f64 atomics of the L2s, the gain by vector issue and HBM: with many new frequencies on
            # a big fan the gain of the next group of frequencies runs on a second stream beside the binning of this one.
            group = MULTI_GROUP
            missing = [f for f in freqs if f not in self.G]
            if group > 0 and len(missing) > group and self.n * len(freqs) >= MULTI_PIPELINE_MIN:
                main, side = torch.cuda.current_stream(), _side_stream(self.device, 0)
                side.wait_stream(main)                               # the trajectories are ready where main stands now
                bounds = list(range(0, len(freqs), group)) + [len(freqs)]
                self.populate(*freqs[bounds[0]:bounds[1]])
                for g in range(len(bounds) - 1):
                    ready = None
                    if g + 2 < len(bounds):
                        with torch.cuda.stream(side):
                            self.populate(*freqs[bounds[g + 1]:bounds[g + 2]])
                            ready = side.record_event()
                        for f in freqs[bounds[g + 1]:bounds[g + 2]]:
                            self.G[f].record_stream(main)
                    bin_maps(bounds[g], bounds[g + 1])
                    if ready is not None:
                        main.wait_event(ready)
            else:
                self.populate(*freqs)
                bin_maps(0, len(freqs))
            self._last_slow = work

    SMALL_TAIL_MAX_RAYS = int(os.environ.get('RT2D_SMALL_TAIL_MAX_RAYS', 6000))     # (0: never)
    SMALL_TAIL_MAX_PTS = 2048

    def small_tail_room(self, freq, policy=True):
        """Upper bound of the stored points per ray if rt2d_small_fan_heatmap may bin this fan (a small stored fan whose gain at
        freq does not exist yet and whose rays fit a workgroup), else 0.  Known on the host without waiting for the cast.
        policy: also require what makes it the better choice -- a fan below the tile kernel's threshold (HEATMAP_FAST_MIN_RAYS,
        the fans the general one-workgroup-per-ray binner would take)."""
        if not self.store or self.n == 0 or freq in self.G:
            return 0
        if policy and self.n > min(self.SMALL_TAIL_MAX_RAYS, HEATMAP_FAST_MIN_RAYS):
            return 0
        if self.padded:
            room = int(self._room)
        elif self._summary is not None:
            room = int(self._summary[1]) // TILE
        else:
            return 0
        if self.rebound_cap * 12 + 8 > 15 * 1024:
            return 0
        if room > self.SMALL_TAIL_MAX_PTS and not (self._pending and self.padded):
            return 0              # (a fresh single-pass cast may have more ROOM than that: the kernel reports a ray that is longer
                                  #  in the word behind the fan's summary, which the caller reads anyway -- validate())
        return room

    def heatmap_small(self, H, res, shape, freq=1, ray_mask=None, dropped=None, out=None, scratch=None, cutoff=.02, reduction_power=.5):
        """Gain at freq and binning of every error-free ray in ONE launch (rt2d_small_fan_heatmap: one workgroup per ray, no
        gain array), plus the normalised map into `out` when given.  Same map as populate(freq) + heatmap_accumulate."""
        torch = _torch()
        room = self.small_tail_room(freq, policy=False)
        if not room:
            raise _lib.Rt2dError('heatmap_small: not a stored fan whose rays fit a workgroup (see RayFan.small_tail_room)')
        with torch.cuda.device(self.device):
            res_c = (C.c_int32 * 2)(int(res[0]), int(res[1]))
            shape_c = (C.c_int32 * 2)(int(shape[0]), int(shape[1]))
            _lib.check(_lib.lib().rt2d_small_fan_heatmap(
                self.env.device_handle(self.device.index), self.X.data_ptr(), self.Z.data_ptr(), self.tile_off.data_ptr(),
                self.n_pts.data_ptr(), C.byref(self._out), self.rebound_cap, self.n, room, float(freq), _ptr(ray_mask), res_c, shape_c,
                H.data_ptr(), _ptr(dropped), float(cutoff), float(reduction_power), _ptr(out), _ptr(scratch),
                self._summary_dev[8:].data_ptr() if room > self.SMALL_TAIL_MAX_PTS else None, _raw_stream()))
        self._last_slow = self.n

    @property
    def last_slow_rays(self):
        """Rays of the last heatmap_accumulate* call that went through the general (one workgroup per ray) kernel."""
        v = getattr(self, '_last_slow', 0)
        if not isinstance(v, int):
            q = int(v[0].item())              # work[0] of rt2d_heatmap_accumulate: -1 = the tile kernel was not used
            v = self.n if q < 0 else q
            self._last_slow = v
        return v

    @last_slow_rays.setter
    def last_slow_rays(self, v):
        self._last_slow = int(v)

    def heatmap_accumulate(self, H, res, shape, freq=1, ray_mask=None, dropped=None, fast=True, hint=True, stream=None, defer=False):
        """Adds this fan's rays (or those with ray_mask != 0, a device uint8 tensor of n) into H (nx*nz, device f64).
        fast=False forces the general one-wave-per-ray kernel for every ray (used by the tests).
        stream=True: gain + binning in one pass (heatmap_accumulate_stream): the gain array (8 B per point) is never
        allocated.  The default for fans of STREAM_MIN_RAYS rays and more whose gain at freq is not there yet: on 1M rays
        the one pass takes 9.1 + 1.7 ms (its kernel, then the general kernel for the rays it cannot bin) against 5.5 +
        6.6 ms for rt2d_gain followed by the tile kernel, and needs no gain array (8 B per point of the layout); a fan
        of 131 072 rays is too few waves for its ray-sequential walk (5.6 against 4.9 ms for the whole cast + map).
        A fan cast with store=False goes through the fused kernel (heatmap_accumulate_fused)."""
        if not self.store:
            if self.__dict__.get('_uncast') or self.n >= CHUNKED_MIN_RAYS:
                if self.heatmap_accumulate_chunked(H, res, shape, freq=freq, ray_mask=ray_mask, dropped=dropped):
                    return
                H.zero_()                        # a chunk outgrew its provisional layout: again, with counted sizes
                if dropped is not None:
                    dropped.zero_()
                self.heatmap_accumulate_chunked(H, res, shape, freq=freq, ray_mask=ray_mask, dropped=dropped)
                return
            return self.heatmap_accumulate_fused(H, res, shape, freq=freq, ray_mask=ray_mask, dropped=dropped)
        if stream is None:
            stream = fast and freq not in self.G and self.n >= STREAM_MIN_RAYS
        if self.__dict__.get('_exact_binning'):
            stream = False                       # (a one-pass binning of this fan met a ray it could not take: gain array + binning kernels)
        if stream:
            return self.heatmap_accumulate_stream(H, res, shape, freq=freq, ray_mask=ray_mask, dropped=dropped, defer=defer)
        torch = _torch()
        L = _lib.lib()
        if freq not in self.G:
            self.populate(freq)
        with torch.cuda.device(self.device):
            res_c = (C.c_int32 * 2)(int(res[0]), int(res[1]))
            shape_c = (C.c_int32 * 2)(int(shape[0]), int(shape[1]))
            fast = fast and self.n > HEATMAP_FAST_MIN_RAYS               # smaller fans go through the general kernel anyway (no
            work = torch.empty(self.n + 2, dtype=torch.int32, device=self.device) if fast else None     # work list, no launch to mark it)
            if fast and hint:
                ray_mask = self.heatmap_mask(res, shape, ray_mask)
            # (what the host knows of the rays' lengths without asking the device: the general kernel sizes its hash by it)
            max_pts = 0
            if self._summary is not None:
                max_pts = int(self._summary[1]) // TILE
            elif self.tile_off is not None and not self._pending and self.total:
                max_pts = int(self.total) // max(1, (self.n + TILE - 1) // TILE * TILE) * 4
            _lib.check(L.rt2d_heatmap_accumulate_hinted(self.X.data_ptr(), self.Z.data_ptr(), self.G[freq].data_ptr(),
                                                        self.tile_off.data_ptr(), self.n_pts.data_ptr(), self.n,
                                                        _ptr(ray_mask), res_c, shape_c, H.data_ptr(), _ptr(dropped), _ptr(work),
                                                        max_pts, _raw_stream()))
            self._last_slow = work if fast else self.n                   # rays binned by the general kernel: read on demand


class _LinearPath:
    """calc_z of a ray: interp1d(X, Z, kind='linear', bounds_error=True) semantics (ray.py:282)."""

    def __init__(self, X, Z):
        order = np.argsort(X, kind='mergesort')
        self.x, self.y = X[order], Z[order]

    def __call__(self, x_new):
        x_new = np.asarray(x_new, dtype=np.float64)
        if np.any(x_new < self.x[0]) or np.any(x_new > self.x[-1]):
            raise ValueError('A value in x_new is outside the interpolation range.')
        idx = np.clip(np.searchsorted(self.x, x_new), 1, len(self.x) - 1)
        lo, hi = idx - 1, idx
        slope = (self.y[hi] - self.y[lo]) / (self.x[hi] - self.x[lo])
        return slope * (x_new - self.x[lo]) + self.y[lo]


class Ray2D:
    """One ray.  Same surface as the reference's Ray2D (ray.py:23-405); the numbers come from a RayFan."""

    def __init__(self, env, source, angle, **kwargs):
        self._propagated = False
        self.stop_reason = None
        self.dist_to_target = None
        self.env = env
        self.source = source
        self.target = kwargs.get('target', None)
        self.angle = angle
        self._fan, self._i = None, 0
        self._cache = {}
        self.freqs = list()
        self.G = dict()
        self.Tmult = dict()
        self.Tmult_target = np.empty((0, 2))
        self.range_min = np.zeros(2)
        self.range_max = np.zeros(2)
        self.steps = 1
        self.n_rebounds = 0

    @classmethod
    def _from_fan(cls, fan, i):
        ray = cls(fan.env, fan.source, float(fan.angles[i]), target=fan.target)
        ray._bind(fan, i)
        return ray

    def _bind(self, fan, i):
        self._fan, self._i = fan, int(i)
        self._propagated = True
        self.backprop, self.n_steps_max = fan.kw['backprop'], fan.kw['n_steps_max']
        self.n_rebounds_max, self.dx_max, self.dz_max = fan.kw['n_rebounds_max'], fan.kw['dx_max'], fan.kw['dz_max']
        self.status = int(fan.host('status')[i])
        self.steps = int(fan.host('n_pts')[i])
        self.stop_reason = STOP_REASONS[int(fan.host('stop_code')[i])]
        self.n_rebounds = int(fan.host('n_rebounds')[i])
        rng = fan.host('range')
        self.range_min = np.array([rng[0, i], rng[1, i]])
        self.range_max = np.array([rng[2, i], rng[3, i]])
        if fan.target is not None:
            self.dist_to_target = np.float64(fan.host('dist')[i])
            self.step_target = int(fan.host('step_target')[i])
            self.T_target = np.float64(fan.host('T_target')[i])
        else:
            self.step_target = None
            self.T_target = None

    @property
    def decision_margin(self):
        """Smallest distance [m] by which any comparison of the loop was decided (diagnostic, not in the reference)."""
        return float(self._fan.host('margin')[self._i])

    def __repr__(self):
        return f'Ray2D ({self.source}, {self.angle})'

    def propagate(self, **kwargs):
        """Propagate this ray on the GPU (reference: ray.py:70-289).  Raises ValueError where the reference
        does (slope table left its range, SURVEY trap 6)."""
        self.verbose = kwargs.get('verbose', False)
        if self._propagated:
            if self.verbose:
                print('Ray2D: ERROR: Ray already propagated')
            return
        kw = {k: kwargs[k] for k in RayFan.PROPAGATE_DEFAULTS if k in kwargs}
        fan = RayFan(self.env, self.source, np.array([self.angle], dtype=np.float64), target=self.target, **kw)
        self._bind(fan, 0)
        if self.status == RAY_ERR_INTERP:
            raise ValueError('A value in x_new is outside the interpolation range.')

    # lazily materialised arrays --------------------------------------------------------------------
    def _array(self, name, tensor):
        """tensor: a tile-layout tensor of the fan, or a callable giving it (evaluated once the fan has its trajectories:
        a subset of a scalars-only fan casts them on first use)."""
        a = self._cache.get(name)
        if a is None:
            self._fan._need_store()
            if callable(tensor):
                tensor = tensor()
            a = self._fan.ray_slice(tensor, self._i).cpu().numpy()
            self._cache[name] = a
        return a

    @property
    def XZ(self):
        if not self._propagated:
            return np.expand_dims(np.asarray(self.source, dtype=np.float64).copy(), axis=0)
        xz = self._cache.get('XZ')
        if xz is None:
            xz = np.stack((self._array('X', lambda: self._fan.X), self._array('Z', lambda: self._fan.Z)), axis=1)
            self._cache['XZ'] = xz
        return xz

    def _post(self, key):
        return self._array(key, lambda: self._fan.post()[key])

    @property
    def dL(self):
        return self._post('dL')[1:]

    @property
    def L(self):
        return self._post('L')

    @property
    def C(self):
        return self._post('C')

    @property
    def dT(self):
        return self._post('dT')[1:]

    @property
    def T(self):
        if not self._propagated:
            return np.array([0., ])
        return self._post('T')

    @property
    def calc_z(self):
        return _LinearPath(self.XZ[:, 0], self.XZ[:, 1])

    @property
    def rebounds(self):
        """The rebound log as the reference's list of dicts (ray.py:185,208)."""
        f, i = self._fan, self._i
        out = []
        for k in range(min(self.n_rebounds, f.rebound_cap)):
            out.append({'step': int(f.host('reb_step')[k, i]),
                        'surface': 'sediment' if f.host('reb_surface')[k, i] == 0 else 'water-surface',
                        'velocity': float(f.host('reb_velocity')[k, i]), 'angle': float(f.host('reb_angle')[k, i]),
                        'position': np.array([f.host('reb_pos_x')[k, i], f.host('reb_pos_z')[k, i]])})
        return out

    # gain --------------------------------------------------------------------------------------------
    def populate_NEW(self, *freqs):
        new = [f for f in freqs if f not in self.freqs]
        if new:
            self._fan.populate(*new)
        for f in new:
            if f in self.freqs:
                continue
            self.G[f] = self._fan.ray_slice(self._fan.G[f], self._i).cpu().numpy()
            self.freqs.append(f)
        self.freqs.sort()

    def populate(self, *freqs):
        """As populate_NEW, plus Tmult = 10^(G/10) (reference: ray.py:291-337; its failure without a target,
        SURVEY trap 5, is not reproduced)."""
        self.populate_NEW(*freqs)
        for f in freqs:
            if f not in self.Tmult:
                self.Tmult[f] = np.power(10, self.G[f] / 10)
                if self.target is not None and len(self.G[f]) > self.step_target:
                    self.Tmult_target = np.insert(self.Tmult_target, 0, np.array([f, self.Tmult[f][self.step_target]]), axis=0)
        self.Tmult_target = np.sort(self.Tmult_target, axis=0)

    def closest_step_to_point(self, target):
        d = np.sqrt(np.power(target[0] - self.XZ[:, 0], 2) + np.power(target[1] - self.XZ[:, 1], 2))
        return np.argmin(d)

    def point(self, step):
        return self.XZ[step]

    def gain(self, step):
        return np.array([self.G[freq][step] for freq in self.freqs])

    def gen_filter(self, target):
        """Travel time and transmittance at a target near the path (reference: ray.py:382-405)."""
        step_target = self.closest_step_to_point(target)
        p = self.point(step_target)
        d_target = np.linalg.norm(target - p)
        z_avg = ((target + p) / 2)[1]
        G_add = -1 * self.env.penv.calc_dl_dG(np.array(self.freqs, dtype=np.float64), z_avg) / 1000 * d_target
        G_target = self.gain(step_target) + G_add
        Tmult_target = np.power(10, G_target / 10)
        T_add = d_target / self.env.penv.calc_c(z_avg)
        return self.T[step_target] + T_add, Tmult_target

    def plot(self, fig, **kwargs):
        if self._propagated:
            import matplotlib.pyplot as plt
            plt.plot(self.XZ[:, 0], self.XZ[:, 1], figure=fig, **kwargs)


class _RayList:
    """List-like view over the rays of a pack; Ray2D objects are created on access."""

    def __init__(self):
        self._segments = []      # (fan, index array) or (None, [Ray2D, ...])
        self._counts = []
        self._made = {}

    def __len__(self):
        return int(sum(self._counts))

    def _add_segment(self, fan, idx):
        self._segments.append((fan, idx))
        self._counts.append(len(idx))

    def __getitem__(self, k):
        if isinstance(k, slice):
            return [self[i] for i in range(*k.indices(len(self)))]
        n = len(self)
        if k < 0:
            k += n
        if not 0 <= k < n:
            raise IndexError(k)
        ray = self._made.get(k)
        if ray is None:
            off = k
            for (fan, idx), cnt in zip(self._segments, self._counts):
                if off < cnt:
                    ray = idx[off] if fan is None else Ray2D._from_fan(fan, int(idx[off]))
                    break
                off -= cnt
            self._made[k] = ray
        return ray

    def __iter__(self):
        for k in range(len(self)):
            yield self[k]


class RayPack2D:
    """Container of rays, one per angle (reference: ray.py:430-489).  Whole fans are added without creating
    per-ray objects (``add_fan``); ``rays``, ``angles``, ``stop_reasons`` and ``dist`` are built lazily."""

    def __init__(self):
        self._rays = _RayList()
        self.freqs = list()
        self._n_rays, self._n_angles, self.n_freqs = 0, 0, 0
        self._range_min_plot = np.zeros(2)
        self._range_max_plot = np.zeros(2)
        self._size_plot = np.zeros(2)
        self._dist_sorted = None
        self._angle_set = set()          # angles of rays added one by one
        self._angle_blocks = []          # sorted angle arrays of the fans added whole
        self._dist_keys = []
        self._maps = None
        self._n_failed = 0       # rays whose status was an error and that were therefore not added
        self._pending = []       # fans added but not yet folded in (add_fan): (fan, indices)

    # The reference's attributes.  Whole fans are folded in on first use (_resolve): casting a fan must not wait for the
    # device, and a heat map needs none of this on the host (Simulation2D.heatmap).
    def _resolve(self):
        while self._pending:
            fan, indices = self._pending.pop(0)
            self._add_fan_now(fan, indices)

    def _sole_pending_fan(self):
        """The fan, if this pack is exactly one whole fan that has not been folded in yet (else None)."""
        if len(self._pending) == 1 and self._n_rays == 0 and not len(self._rays):
            fan, indices = self._pending[0]
            if indices is None or (getattr(fan, '_whole_in_pack', False) and len(indices) == fan.n):
                return fan
        return None

    rays = property(lambda self: (self._resolve(), self._rays)[1])
    n_rays = property(lambda self: (self._resolve(), self._n_rays)[1])
    n_angles = property(lambda self: (self._resolve(), self._n_angles)[1])
    n_failed = property(lambda self: (self._resolve(), self._n_failed)[1])
    range_min_plot = property(lambda self: (self._resolve(), self._range_min_plot)[1])
    range_max_plot = property(lambda self: (self._resolve(), self._range_max_plot)[1])
    size_plot = property(lambda self: (self._resolve(), self._size_plot)[1])
    dist_sorted = property(lambda self: (self._resolve(), self._dist_sorted)[1])

    def __repr__(self):
        return f'Raypack containing {self.n_rays} rays cast with {self.n_angles} angles'

    def _fold_range(self, rmin, rmax):
        # ray.py:466-472: starts from zeros, so the origin is always inside the plot range
        self._range_min_plot = np.minimum(self._range_min_plot, rmin)
        self._range_max_plot = np.maximum(self._range_max_plot, rmax)
        self._size_plot = self._range_max_plot - self._range_min_plot

    def _known_angles(self, a):
        """Boolean mask: which of the angles `a` the pack already holds."""
        a = np.atleast_1d(np.asarray(a, dtype=np.float64))
        known = np.zeros(len(a), dtype=bool)
        if self._angle_set:
            if len(a) <= 64:
                known |= np.fromiter((x in self._angle_set for x in a.tolist()), dtype=bool, count=len(a))
            else:
                known |= np.isin(a, np.fromiter(self._angle_set, dtype=np.float64, count=len(self._angle_set)))
        for block in self._angle_blocks:
            pos = np.minimum(np.searchsorted(block, a), len(block) - 1)
            known |= block[pos] == a
        return known

    def add_fan(self, fan, indices=None):
        """Add rays of a fan (all, or the given indices), skipping errored rays and angles already present.  The fan is
        folded in when the pack is first looked at."""
        if indices is not None and not hasattr(indices, 'data_ptr'):         # (a device tensor is fetched when the pack is resolved)
            indices = np.asarray(indices, dtype=np.int64)
        self._pending.append((fan, indices))
        self._maps = None

    def _add_fan_now(self, fan, indices=None):
        if indices is not None and hasattr(indices, 'data_ptr'):
            indices = indices.cpu().numpy()
        ang = np.asarray(fan.angles, dtype=np.float64)
        status = fan.host('status')
        idx_all = np.arange(fan.n) if indices is None else np.asarray(indices, dtype=np.int64)
        # whole-array form of the reference's per-ray loop (ray.py:451-465): errored rays are left out, an angle is kept
        # the first time it is seen (a million-ray fan must not spend a second in a Python loop here)
        ok = status[idx_all] != RAY_ERR_INTERP
        self._n_failed += int((~ok).sum())
        cand = idx_all[ok]
        a = ang[cand]
        step = np.diff(a)
        rising, falling = bool((step > 0).all()), bool((step < 0).all())
        if not (rising or falling):                            # a fan is normally a strictly monotonic sweep: all distinct
            _, first = np.unique(a, return_index=True)         # index of the first occurrence of every distinct angle
            if len(first) != len(a):
                first.sort()
                cand, a = cand[first], a[first]
        if self._angle_set or self._angle_blocks:
            fresh = ~self._known_angles(a)
            if not fresh.all():
                cand, a = cand[fresh], a[fresh]
        if 0 < len(a) <= 64:
            self._angle_set.update(a.tolist())                  # a handful of rays (refine adds them one by one): the set
        elif len(a):
            self._angle_blocks.append(a.copy() if rising else a[::-1].copy() if falling else np.sort(a))
        keep = cand
        if not len(keep):
            return
        keep = np.asarray(keep, dtype=np.int64)
        self._rays._add_segment(fan, keep)
        self._n_rays += len(keep)
        self._n_angles += len(keep)
        summ = getattr(fan, '_summary', None)
        if summ is not None and len(keep) == fan.n and int(summ[3]) == fan.n:
            lo_hi = np.array([-summ[4], -summ[5], summ[6], summ[7]])     # every ray kept and error-free: the cast's own fold
        elif 'range' in fan._host or len(keep) < 4096:
            rng = fan.host('range')
            lo_hi = np.array([rng[0, keep].min(), rng[1, keep].min(), rng[2, keep].max(), rng[3, keep].max()])
        else:                                                    # big fan: fold on the device, 4 numbers come back
            torch = _torch()
            rng = fan.range if len(keep) == fan.n else fan.range[:, torch.as_tensor(keep, device=fan.device)]
            lo_hi = torch.cat((rng[:2].amin(dim=1), rng[2:].amax(dim=1))).cpu().numpy()
        self._fold_range(lo_hi[:2], lo_hi[2:])
        if fan.target is not None:
            self._dist_keys.append(fan.host('dist')[keep])
        else:
            self._dist_keys.append(np.array([np.nan]))           # placeholder for dist_to_target = None (one key, as in a dict)
        self._update_dist_sorted()
        self._maps = None

    def add(self, *rays):
        self._resolve()
        for ray in rays:
            if ray.angle in self._angle_set or self._known_angles(ray.angle)[0]:
                continue
            self._angle_set.add(ray.angle)
            if ray._fan is not None:
                self._rays._add_segment(ray._fan, np.array([ray._i], dtype=np.int64))
                self._rays._made[len(self._rays) - 1] = ray
            else:
                self._rays._add_segment(None, [ray])
            self._n_rays += 1
            self._n_angles += 1
            self._fold_range(ray.range_min, ray.range_max)
            d = ray.dist_to_target
            self._dist_keys.append(np.array([np.nan if d is None else d], dtype=np.float64))
        self._update_dist_sorted()
        self._maps = None

    def _update_dist_sorted(self):
        if self._dist_keys:
            self._dist_sorted = np.sort(np.unique(np.concatenate(self._dist_keys)))   # unique keeps a single NaN, like dict keys of one None

    def _build_maps(self):
        if self._maps is None:
            angles, stops, dist = {}, {}, {}
            for ray in self.rays:
                angles[ray.angle] = ray
                stops.setdefault(ray.stop_reason, []).append(ray)
                dist.setdefault(ray.dist_to_target, []).append(ray)
            self._maps = (angles, stops, dist)
        return self._maps

    @property
    def angles(self):
        return self._build_maps()[0]

    @property
    def stop_reasons(self):
        return self._build_maps()[1]

    @property
    def dist(self):
        return self._build_maps()[2]

    def segments(self):
        """[(fan, index array)] of the fan-backed rays, for batched device work."""
        self._resolve()
        return [(fan, idx) for fan, idx in self._rays._segments if fan is not None]

    def populate(self, *freqs):
        fans = [f for f, _ in self._pending] if self._pending and not len(self._rays) else [f for f, _ in self.segments()]
        for fan in fans:
            fan.populate(*freqs)
        for f in freqs:
            if f not in self.freqs:
                self.freqs.append(f)
        self.n_freqs = len(self.freqs)
