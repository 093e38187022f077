"""Import shim: `import dsp_map_amd` -> the package directory `dsp-map_amd/`
(a hyphen is not a valid identifier, importlib does not care)."""
import importlib
import os
import sys

_root = os.path.dirname(os.path.abspath(__file__))
if _root not in sys.path:
    sys.path.insert(0, _root)
_pkg = importlib.import_module("dsp-map_amd")
sys.modules[__name__] = _pkg
sys.modules.setdefault("dsp_map_amd", _pkg)
