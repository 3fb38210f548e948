"""Import shim: the package directory `lightdiffusion-next_amd/` is not a valid Python identifier, so
`import ldx_amd` loads it through importlib and aliases it."""
import importlib
import os
import sys

_root = os.path.dirname(os.path.abspath(__file__))
if _root not in sys.path:
    sys.path.insert(0, _root)
_pkg = importlib.import_module("lightdiffusion-next_amd")
sys.modules[__name__] = _pkg
