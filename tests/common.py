"""Shared helpers of the parity tests: golden fixtures -> product environment / oracle environment."""

import os

import numpy as np

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')

# tolerances (north_star: bit-exact termination / rebound counts, 1e-9 relative on x/z)
TRAJ_RTOL = 1e-9        # met by >= 99 % of the rays of every fan (and by every ray with few rebounds)
TRAJ_RTOL_WORST = 4e-8  # ceiling for the most sensitive rays: 20 rebounds on a curved floor amplify a 1-ulp
                        # difference up to ~1e7 times (tests/test_oracle.py::test_sensitivity shows the reference's
                        # own algorithm doing so); measured maxima: 6.4e-9 (c2) and 8.9e-9 (c3) against the reference
                        # (tests/golden/PARITY.md), 2.65e-8 (c2) device tan/sin against the oracle's libm;
                        # see DESIGN.md "Sensitivity"
SCALAR_RTOL = 1e-9
KNIFE_EDGE_M = 1e-7     # a ray whose smallest decision margin is below this [m] is knife-edge


def load(name):
    return dict(np.load(os.path.join(GOLD, name + '.npz')))


def prop_kwargs(g):
    return dict(dx_max=float(g['fan_dx_max']), dz_max=float(g['fan_dz_max']), n_steps_max=int(g['fan_n_steps_max']),
                n_rebounds_max=int(g['fan_n_rebounds_max']), backprop=bool(g['fan_backprop']))


def target_of(g):
    return g['fan_target'] if int(g['fan_has_target']) else None


def product_env(rt, g):
    return rt.Environment2D((g['env_floor_x'], g['env_floor_z']), (g['env_ceil_x'], g['env_ceil_z']),
                            g['env_range_min'], g['env_range_max'], res_x=int(g['env_res_x']), res_z=int(g['env_res_z']))


def oracle_env(orc, g):
    fl = orc.LinearFloor(g['env_floor_x'], g['env_floor_z'])
    ce = orc.LinearFloor(g['env_ceil_x'], g['env_ceil_z'])
    return orc.OracleEnv(fl, ce, g['env_range_min'], g['env_range_max'], int(g['env_res_x']), int(g['env_res_z']))


def oracle_fan(orc, env, g, angles=None, init=None, **over):
    kw = prop_kwargs(g)
    kw.update(over)
    p = orc.make_params(target=target_of(g), **kw)
    angles = g['fan_angles'] if angles is None else angles
    rays = []
    for i, a in enumerate(angles):
        rays.append(orc.cast_ray(env, g['fan_source'], a, p, init=None if init is None else (init[0][i], init[1][i])))
    return rays


def host_init(c0, angles):
    """Start slope and curvature multiplier computed with NumPy exactly as ray.py:138-146 does."""
    a = -1 * np.asarray(angles) + (np.pi / 2)
    return 1 / np.tan(a), -1 * np.power(c0 / np.sin(a), 2)


def rel_err(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return np.abs(a - b) / np.maximum(1.0, np.abs(b))


def oracle_env_from_product(orc, env):
    """Oracle reading the PRODUCT's tables: HIP and C then evaluate identical numbers."""
    return orc.OracleEnv.from_tables(env.tables(), env.range_min, env.range_max)


def oracle_rays_from_fan(orc, rt, fan):
    """Oracle rays carrying the GPU's own points and rebound log (to test later stages in isolation)."""
    rays = []
    for i in range(fan.n):
        ray = rt.Ray2D._from_fan(fan, i)
        nl = min(ray.n_rebounds, fan.rebound_cap)
        r = orc.ray_from_points(ray.XZ, fan.host('reb_step')[:nl, i], fan.host('reb_surface')[:nl, i],
                                fan.host('reb_velocity')[:nl, i], fan.host('reb_angle')[:nl, i],
                                fan.host('reb_pos_z')[:nl, i], status=int(fan.host('status')[i]))
        r.range_min = np.array(fan.host('range')[:2, i])
        r.range_max = np.array(fan.host('range')[2:, i])
        rays.append(r)
    return rays
