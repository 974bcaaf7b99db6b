"""Ocean-acoustics closed forms and tabulated water-column profiles (host side, NumPy).

These run once per environment over O(500) depth samples and only feed the device tables, so
they stay on the host ("not on the path but feeding it").  Each function
states which reference routine it stands in for; the models themselves are the published ones
(Medwin sound speed, UNESCO IES-80 density, Ainslie-McColm absorption, coherent rough-surface
and layered-sediment reflection).
"""

import numpy as np

from .pptable import (PPTable, linear_interp_table, quadratic_interp_table, OOB_EXTRAPOLATE)
from .preferences import WAVE_HEIGHT_RMS_DEFAULT, DZ_SEDIMENT_DEFAULT

# --- water-column profiles ------------------------------------------------------------------
# (altitude z [m] <= 0, value) sample points; same data as the reference's
# physics/profile_temperature.py:10-28 and physics/profile_ph.py:10-25.
_TEMPERATURE_SAMPLES = (
    (0, 22), (-100, 21), (-200, 20), (-300, 16), (-400, 10), (-500, 4), (-600, 3.5), (-700, 3),
    (-800, 3), (-900, 3), (-1000, 3), (-1100, 3), (-1200, 3), (-1300, 3), (-1400, 3), (-1500, 3),
    (-10000, 3),
)
_PH_SAMPLES = (
    (0, 7.98), (-100, 7.88), (-200, 7.84), (-250, 7.87), (-300, 7.85), (-400, 7.79), (-450, 7.7),
    (-500, 7.7), (-600, 7.71), (-800, 7.73), (-1000, 7.72), (-2000, 7.73), (-3000, 7.74),
    (-10000, 7.74),
)
SALINITY_DEFAULT = 35

_t = np.array(_TEMPERATURE_SAMPLES, dtype=np.float64)[::-1]
_p = np.array(_PH_SAMPLES, dtype=np.float64)[::-1]
#: temperature [deg C] as a quadratic interpolant with extrapolation (reference: profile_temperature.py:30-36)
#: pH as a linear interpolant with extrapolation (reference: profile_ph.py:27-33)
# These two only produce the O(500) samples the depth tables are built from.  dc/dz is a difference of
# neighbouring c samples, so one ulp of T moves it by 1e-12 relative and the ray dynamics amplify that further;
# with scipy present the samples are therefore taken from the same interp1d objects the reference uses.
try:
    from scipy.interpolate import interp1d as _interp1d
    calc_T = _interp1d(_t[:, 0], _t[:, 1], kind='quadratic', bounds_error=False, fill_value='extrapolate')
    calc_pH = _interp1d(_p[:, 0], _p[:, 1], kind='linear', bounds_error=False, fill_value='extrapolate')
except ImportError:                         # pragma: no cover
    calc_T = quadratic_interp_table(_t[:, 0], _t[:, 1], oob=OOB_EXTRAPOLATE)
    calc_pH = linear_interp_table(_p[:, 0], _p[:, 1], oob=OOB_EXTRAPOLATE)


def calc_S(z):
    """Practical salinity, constant 35 (reference: profile_salinity.py:9-12)."""
    return np.ones_like(z) * SALINITY_DEFAULT


def calc_P(z):
    """Applied pressure, 0.1 per metre of depth (reference: profile_pressure.py:9-13)."""
    return np.abs(z) * 0.1


# --- sediment impedances (reference: profile_impedance.py:8-19) ----------------------------------
Z_REFERENCE_VALUES = {
    'clayey silt': 2390000, 'silt': 2550000, 'sand silt clay': 2660000, 'sandy silt': 2680000,
    'silty clay': 3160000, 'rock': 17000000,
}
Zp1_DEFAULT = float(np.mean([Z_REFERENCE_VALUES[k] for k in
                             ('clayey silt', 'silt', 'sand silt clay', 'sandy silt', 'silty clay')]))
Zp2_DEFAULT = float(Z_REFERENCE_VALUES['rock'])


# --- models ----------------------------------------------------------------------------------
def sound_velocity_medwin(z, S, T):
    """Medwin sound speed [m/s] (reference: model_velocity.py:42-58; same term order)."""
    return 1449.2 + \
        4.6 * T - 0.055 * np.power(T, 2) + 0.00029 * np.power(T, 3) + \
        (1.34 - 0.01 * T) * (S - 35) - \
        0.016 * z


def calc_rho(z, S, T, P):
    """IES-80 high-pressure equation of state of seawater [kg/m^3] (reference: model_rho.py:8-49).

    The UNESCO polynomial is kept term by term (np.power forms) so table values agree with the
    reference's to the last bits."""
    T2, T3, T4, T5 = np.power(T, 2), np.power(T, 3), np.power(T, 4), np.power(T, 5)
    S15, S2 = np.power(S, 1.5), np.power(S, 2)
    rho_w = 999.842594 + 6.793952e-2 * T - 9.09529e-3 * T2 + 1.001685e-4 * T3 - 1.120083e-6 * T4 + 6.536336e-9 * T5
    a = 8.24493e-1 - 4.0899e-3 * T + 7.6438e-5 * T2 - 8.2467e-7 * T3 + 5.3875e-9 * T4
    b = -5.72466e-3 + 1.0227e-4 * T - 1.6546e-6 * T2
    c = 4.8314e-4
    rho_1atm = rho_w + a * S + b * S15 + c * S2

    k_w = 19652.21 + 148.4206 * T - 2.327105 * T2 + 1.360477e-2 * T3 - 5.155288e-5 * T4
    a_w = 3.239908 + 1.43713e-3 * T + 1.16092e-4 * T2 - 5.79905e-7 * T3
    b_w = 8.50935e-5 - 6.12293e-6 * T + 5.2787e-8 * T2
    k_1atm = k_w + \
        (54.6746 - 0.603459 * T + 1.09987e-2 * T2 - 6.1670e-5 * T3) * S + \
        (7.944e-2 + 1.6483e-2 * T - 5.3009e-4 * T2) * S15
    a_p = a_w + \
        (2.2838e-3 - 1.0981e-5 * T - 1.6078e-6 * T2) * S + \
        1.91075e-3 * S15
    b_p = b_w + \
        (-9.9348e-7 + 2.0816e-8 * T + 9.1697e-10 * T2) * S
    k = k_1atm + a_p * P + b_p * np.power(P, 2)
    return rho_1atm / (1 - P / k)


def calc_Z(rho, c):
    """Specific acoustic impedance (reference: model_impedance.py:5-6)."""
    return rho * c


ABSORPTION_A1 = 0.106
ABSORPTION_A3 = 0.00049


def calc_dl_dG_coefs(z, T, S, pH):
    """Depth-dependent factors of the Ainslie-McColm absorption, shape (6, N):
    f1, f2 [kHz], A2, E1, E2, E3 (reference: model_absorption.py:8-16)."""
    f1 = 0.78 * np.sqrt(S / 35) * np.exp(T / 26)
    f2 = 42 * np.exp(T / 17)
    a2 = 0.52 * (1 + T / 43) * (S / 35)
    e1 = np.exp((pH - 8) / 0.56)
    e2 = np.exp(z / 6000)
    e3 = np.exp(z / 17000 - T / 27)
    return np.stack((f1, f2, a2, e1, e2, e3))


def absorption_from_coefs(f_hz, coefs):
    """Seawater absorption [dB/km] at frequency ``f_hz`` from interpolated depth factors
    (reference: model_absorption.py:19-46; the input frequency is NOT modified in place here)."""
    f1, f2, a2, e1, e2, e3 = coefs
    f = f_hz / 1000
    ff = np.power(f, 2)
    alpha_1 = ABSORPTION_A1 * (f1 * ff) / (np.power(f1, 2) + ff)
    alpha_2 = a2 * (f2 * ff) / (np.power(f2, 2) + ff)
    alpha_3 = ABSORPTION_A3 * ff
    return alpha_1 * e1 + alpha_2 * e2 + alpha_3 * e3


def calc_refcoef_surface(wavelength, angle, wave_height_rms=WAVE_HEIGHT_RMS_DEFAULT):
    """Coherent rough-surface reflection coefficient (reference: model_reflection.py:11-18)."""
    return np.exp(-2 * np.power(2 * np.pi / wavelength * wave_height_rms * np.sin(angle), 2))


def calc_refcoef_sediment(wavelength, Zp0, dz_sediment=DZ_SEDIMENT_DEFAULT, Zp1=Zp1_DEFAULT, Zp2=Zp2_DEFAULT):
    """Layered-sediment reflection coefficient, shear ignored (reference: model_reflection.py:21-46
    with its defaults Zs2 = theta_s2 = 0)."""
    k1 = 2 * np.pi / wavelength
    r01 = (Zp1 - Zp0) / (Zp1 + Zp0)
    r12 = (Zp2 - Zp1) / (Zp2 + Zp1)
    r12_exp = r12 * np.exp(-2j * k1 * dz_sediment)
    return np.abs((r01 + r12_exp) / (1 + r01 * r12_exp))
