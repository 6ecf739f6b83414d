"""Import alias: `import drone2d_amd` returns the package in gym-drone2d-activeperception_amd/ (whose
directory name cannot be written in an import statement)."""
import importlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.modules[__name__] = importlib.import_module('gym-drone2d-activeperception_amd')
