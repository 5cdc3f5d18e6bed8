"""Import helper: the package directory `jefferson-2.0_amd` is not a valid Python
identifier, so it is loaded by path under the module name `jefferson_amd`.

    from jf_load import jf
"""
import importlib.util
import os
import sys

_ROOT = os.path.dirname(os.path.abspath(__file__))
_PKG = os.path.join(_ROOT, "jefferson-2.0_amd")


def _load():
    name = "jefferson_amd"
    if name in sys.modules:
        return sys.modules[name]
    spec = importlib.util.spec_from_file_location(name, os.path.join(_PKG, "__init__.py"),
                                                  submodule_search_locations=[_PKG])
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


jf = _load()
