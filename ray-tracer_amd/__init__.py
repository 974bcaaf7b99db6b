"""rt2d on MI355X: the 2D acoustic ray path of ahernot/ray-tracer behind its own Python API.

The directory is called ``ray-tracer_amd`` (not importable by that name); load it as the package
``ray_tracer_amd`` with ``rt2d_loader.load()`` from the repository root, or put a symlink on sys.path.
Compute happens only in ``librt2d_hip.so`` (C ABI: include/rt2d.h, sources: csrc/).
"""

from .preferences import *  # noqa: F401,F403
from .pptable import PPTable, linear_interp_table, quadratic_interp_table, tabulate_callable  # noqa: F401
from .physics_environment import PhysicsEnvironment2D  # noqa: F401
from .environment import Environment2D  # noqa: F401
from .rays import RayFan, Ray2D, RayPack2D  # noqa: F401
from .simulation import Simulation2D, EigenraySim2D  # noqa: F401
from .bathymetry import Map  # noqa: F401
from .presentation import heatmap_ticks, floor_mask, heatmap_image, plot_heatmap  # noqa: F401
from . import _lib  # noqa: F401

build = _lib.build
