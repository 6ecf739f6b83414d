"""Import alias: `import drone2d_amd` returns the package in gym-drone2d-activeperception_amd/ (whose
directory name cannot be written in an import statement)."""
import importlib
import os
import sys

_REAL = 'gym-drone2d-activeperception_amd'
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
_pkg = importlib.import_module(_REAL)
sys.modules[__name__] = _pkg
for _k, _v in list(sys.modules.items()):          # submodules loaded by the package's own __init__
    if _k.startswith(_REAL + '.'):
        sys.modules.setdefault(__name__ + _k[len(_REAL):], _v)
