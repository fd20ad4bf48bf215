"""Import shim: `import pifuhd_amd` loads the package that lives in `rgb-d-pifuhd_amd/`.

The package directory name is fixed by the repo layout contract and is not a valid Python
identifier, so this module replaces itself in `sys.modules` with that package.
"""
import importlib.util
import os
import sys

_pkg_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'rgb-d-pifuhd_amd')
_spec = importlib.util.spec_from_file_location(
    'pifuhd_amd', os.path.join(_pkg_dir, '__init__.py'), submodule_search_locations=[_pkg_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules['pifuhd_amd'] = _mod
_spec.loader.exec_module(_mod)
