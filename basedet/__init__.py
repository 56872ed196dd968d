"""Import alias: `basedet` -> `basedet_amd`.

A caller written against the reference (`from basedet.configs import RetinaNetConfig`, `from basedet.layers import Matcher`,
`from basedet.structures import Boxes`, `basedet.utils.registers`, a playground `config.py`) imports this package and lands on the
MI355X implementation of the same names.  No reference code lives here: every submodule is the basedet_amd module of that name."""
import importlib
import sys

import basedet_amd

_SUBMODULES = ("configs", "layers", "structures", "models", "solver", "utils", "data", "evaluators", "engine")

for _name in _SUBMODULES:
    _mod = importlib.import_module("basedet_amd." + _name)
    sys.modules[__name__ + "." + _name] = _mod
    if "." not in _name:
        globals()[_name] = _mod

__version__ = getattr(basedet_amd, "__version__", "0.2.0")
