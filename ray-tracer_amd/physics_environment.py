"""Frequency-invariant water-column fields on a depth grid, as device-ready tables.

Host-side mirror of the reference's ``PhysicsEnvironment2D`` (physics_environment.py:32-86): same
constructor, same attributes (``x, z, S, T, P, pH, c, dz_c, rho, Z`` and the callables
``calc_c, calc_dz_c, calc_rho, calc_Z, calc_dl_dG``).  The callables are :class:`PPTable`
objects, i.e. exactly the piecewise-quadratic tables the HIP kernels stage into LDS.
"""

import numpy as np

from .preferences import RES_X_DEFAULT, RES_Z_DEFAULT
from . import physics
from .pptable import PPTable, quadratic_spline_pieces, quadratic_interp_table, OOB_EXTRAPOLATE

X_DEFAULT = 0.
Z_DEFAULT = 0.


class PhysicsEnvironment2D:

    def __init__(self, range_min, range_max, **kwargs):
        """
        :param range_min: lower-left corner (x, z) of the simulated rectangle
        :param range_max: upper-right corner (x, z)
        :param kwargs/res_x: horizontal sampling step (m)
        :param kwargs/res_z: vertical sampling step (m)
        """
        self.res_x = kwargs.get('res_x', RES_X_DEFAULT)
        self.res_z = kwargs.get('res_z', RES_Z_DEFAULT)
        self.range_min = np.asarray(range_min)
        self.range_max = np.asarray(range_max)
        self.size = self.range_max - self.range_min

        # sample counts as in physics_environment.py:53-56 (floor division, one extra point)
        n_x = int(self.size[0] // self.res_x) + 1 if self.res_x != 0 else 1
        n_z = int(self.size[1] // self.res_z) + 1 if self.res_z != 0 else 1
        self.x = np.linspace(self.range_min[0], self.range_max[0], n_x) if self.res_x != 0 else np.array([X_DEFAULT])
        self.z = np.linspace(self.range_min[1], self.range_max[1], n_z) if self.res_z != 0 else np.array([Z_DEFAULT])
        if self.z.shape[0] < 3:
            raise ValueError('the depth grid needs at least 3 samples for quadratic tables')
        self._generate()

    def _generate(self):
        z = self.z
        self.S = physics.calc_S(z)
        self.T = physics.calc_T(z)
        self.P = physics.calc_P(z)
        self.pH = physics.calc_pH(z)

        # sound speed and its vertical derivative; the derivative divides by the NOMINAL res_z,
        # not the true linspace spacing, as physics_environment.py:72 does
        self.c = physics.sound_velocity_medwin(z, self.S, self.T)
        self.calc_c = quadratic_interp_table(z, self.c, oob=OOB_EXTRAPOLATE)
        self.dz_c = np.gradient(self.c) / self.res_z
        self.calc_dz_c = quadratic_interp_table(z, self.dz_c, oob=OOB_EXTRAPOLATE)

        self.rho = physics.calc_rho(z, self.S, self.T, self.P)
        self.calc_rho = quadratic_interp_table(z, self.rho, oob=OOB_EXTRAPOLATE)

        # absorption: six depth factors interpolated in z, frequency applied at evaluation time
        self._dl_dG_coefs = physics.calc_dl_dG_coefs(z, self.T, self.S, self.pH)          # (6, N)
        breaks, coef = quadratic_spline_pieces(z, self._dl_dG_coefs.T)                      # (N-2, 3, 6)
        self.absorption_tables = [PPTable(breaks, coef[:, :, k], oob=OOB_EXTRAPOLATE) for k in range(6)]

        self.Z = physics.calc_Z(self.rho, self.c)
        self.calc_Z = quadratic_interp_table(z, self.Z, oob=OOB_EXTRAPOLATE)

    def calc_dl_dG_coefs_interp(self, z):
        return np.stack([t(z) for t in self.absorption_tables])

    def calc_dl_dG(self, f, z):
        """Absorption [dB/km] at frequency f [Hz] and altitude z (reference: physics_environment.py:82)."""
        return physics.absorption_from_coefs(np.asarray(f, dtype=np.float64), self.calc_dl_dG_coefs_interp(z))
