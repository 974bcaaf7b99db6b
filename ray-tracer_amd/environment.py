"""2D environment: bounds, sea floor / surface tables and the device handle.

Host-side mirror of the reference's ``Environment2D`` (environment.py:20-61): same constructor
(``floor, ceiling, range_min, range_max, res_x=, res_z=``) and attributes (``floor, ceil, dx_floor,
dx_ceil, floor_avg, ceil_avg, range_min, range_max, size, penv, plot``).

Tabulation contract: the reference calls ``floor(x)`` / ``ceiling(x)`` as Python callables at every
step (ray.py:172-173), which a GPU cannot do.  Here a callable is sampled once on a uniform grid
(``res_x / floor_oversample``) into a piecewise-linear :class:`PPTable`; ``env.floor`` IS that
table (still a callable, usable from any host code).  Pass a :class:`PPTable` or an ``(x, z)``
pair of arrays to control the table exactly (e.g. a bathymetry cut).
"""

import ctypes as C

import numpy as np

from .preferences import RES_X_DEFAULT, RES_Z_DEFAULT, FLOOR_OVERSAMPLE_DEFAULT
from .physics_environment import PhysicsEnvironment2D
from .pptable import (PPTable, linear_interp_table, quadratic_interp_table, tabulate_callable,
                      OOB_CLAMP, OOB_ERROR)
from . import _lib


def _as_surface_table(obj, x_min, x_max, n_samples):
    if isinstance(obj, PPTable):
        if obj.order != 2:
            raise ValueError('floor / ceiling tables must be piecewise linear (order 2)')
        # the stepping kernel clamps both surfaces outside their breaks
        return obj if obj.oob == OOB_CLAMP else PPTable(obj.breaks, obj.coef, oob=OOB_CLAMP)
    if isinstance(obj, (tuple, list)) and len(obj) == 2 and np.ndim(obj[0]) == 1:
        return linear_interp_table(obj[0], obj[1], oob=OOB_CLAMP)
    if callable(obj):
        # a reference-style object exposing its own (xs, zs) samples keeps them exactly
        xs, zs = getattr(obj, 'xs', None), getattr(obj, 'zs', None)
        if xs is not None and zs is not None:
            return linear_interp_table(xs, zs, oob=OOB_CLAMP)
        return tabulate_callable(obj, x_min, x_max, n_samples, oob=OOB_CLAMP)
    raise TypeError('floor / ceiling must be a callable, a PPTable or an (x, z) pair of arrays')


class Environment2D:

    def __init__(self, floor, ceiling, range_min, range_max, **kwargs):
        """
        :param floor: sea floor z(x): callable, PPTable or (x, z) arrays
        :param ceiling: surface z(x): same
        :param range_min / range_max: simulation rectangle corners (x, z)
        :param kwargs/res_x, res_z: sampling steps of the physics grid (m)
        :param kwargs/floor_oversample: floor/ceiling table samples per res_x (build-specific)
        """
        self.res_x = kwargs.get('res_x', RES_X_DEFAULT)
        self.res_z = kwargs.get('res_z', RES_Z_DEFAULT)
        self.range_min = np.asarray(range_min)
        self.range_max = np.asarray(range_max)
        self.size = self.range_max - self.range_min

        self.penv = PhysicsEnvironment2D(self.range_min, self.range_max, res_x=self.res_x, res_z=self.res_z)

        over = int(kwargs.get('floor_oversample', FLOOR_OVERSAMPLE_DEFAULT))
        n_samples = over * (self.penv.x.shape[0] - 1) + 1
        self.floor = _as_surface_table(floor, float(self.range_min[0]), float(self.range_max[0]), n_samples)
        self.ceil = _as_surface_table(ceiling, float(self.range_min[0]), float(self.range_max[0]), n_samples)

        # slopes sampled on the physics x grid with the nominal res_x, out of range = error
        # (environment.py:46-53)
        self._floor_sampled = self.floor(self.penv.x)
        self._ceil_sampled = self.ceil(self.penv.x)
        self._dx_floor_sampled = np.gradient(self._floor_sampled) / self.res_x
        self._dx_ceil_sampled = np.gradient(self._ceil_sampled) / self.res_x
        self.dx_floor = quadratic_interp_table(self.penv.x, self._dx_floor_sampled, oob=OOB_ERROR)
        self.dx_ceil = quadratic_interp_table(self.penv.x, self._dx_ceil_sampled, oob=OOB_ERROR)
        self.floor_avg = np.mean(self._floor_sampled)
        self.ceil_avg = np.mean(self._ceil_sampled)

        self._handles = {}     # device index -> rt2d_env_t

    def __repr__(self):
        return self.__class__.__name__ + ' object'

    def plot(self, fig, **kwargs):
        import matplotlib.pyplot as plt
        plt.plot(self.penv.x, self._ceil_sampled, figure=fig, **kwargs)
        plt.plot(self.penv.x, self._floor_sampled, figure=fig, **kwargs)

    # -- device side ------------------------------------------------------------------------------
    def tables(self):
        """The 13 tables of Rt2dEnvDesc, by field name."""
        p = self.penv
        t = dict(c=p.calc_c, dz_c=p.calc_dz_c, floor=self.floor, ceil=self.ceil, dx_floor=self.dx_floor,
                 dx_ceil=self.dx_ceil, impedance=p.calc_Z)
        for k in range(6):
            t[f'absorb{k}'] = p.absorption_tables[k]
        return t

    def env_desc(self):
        """(Rt2dEnvDesc, keep-alive list) over host NumPy memory."""
        desc = _lib.Rt2dEnvDesc()
        desc.range_min[:] = [float(v) for v in self.range_min]
        desc.range_max[:] = [float(v) for v in self.range_max]
        keep = []

        def fill(dst, tab):
            dst.order, dst.n_pieces, dst.oob = tab.order, tab.n_pieces, tab.oob
            dst.breaks = tab.breaks.ctypes.data
            dst.coef = tab.coef.ctypes.data
            keep.append(tab)

        t = self.tables()
        for name in ('c', 'dz_c', 'floor', 'ceil', 'dx_floor', 'dx_ceil', 'impedance'):
            fill(getattr(desc, name), t[name])
        for k in range(6):
            fill(desc.absorb[k], t[f'absorb{k}'])
        return desc, keep

    def device_handle(self, device_index):
        """rt2d_env_t for a GPU (created on first use, tables copied once)."""
        h = self._handles.get(device_index)
        if h is None:
            import torch
            L = _lib.lib()
            desc, keep = self.env_desc()
            out = C.c_void_p()
            with torch.cuda.device(device_index):
                _lib.check(L.rt2d_env_create(C.byref(desc), C.byref(out)))
            h = out.value
            self._handles[device_index] = h
        return h

    def __del__(self):
        try:
            L = _lib._lib
            if L is not None:
                for h in self._handles.values():
                    L.rt2d_env_destroy(h)
        except Exception:
            pass
