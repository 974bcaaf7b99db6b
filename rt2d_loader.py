"""Import helper: the package directory is ``ray-tracer_amd`` (a hyphen is not a Python identifier), so it is
loaded under the module name ``ray_tracer_amd``."""

import importlib.util
import os
import sys

_ROOT = os.path.dirname(os.path.abspath(__file__))
PKG_DIR = os.path.join(_ROOT, 'ray-tracer_amd')


def load():
    mod = sys.modules.get('ray_tracer_amd')
    if mod is not None:
        return mod
    spec = importlib.util.spec_from_file_location('ray_tracer_amd', os.path.join(PKG_DIR, '__init__.py'),
                                                  submodule_search_locations=[PKG_DIR])
    mod = importlib.util.module_from_spec(spec)
    sys.modules['ray_tracer_amd'] = mod
    spec.loader.exec_module(mod)
    return mod
