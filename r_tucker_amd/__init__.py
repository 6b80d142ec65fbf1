"""``import r_tucker_amd`` -> the package in ``r-tucker_amd/`` (the directory name the project layout
prescribes is not a Python identifier).  The real package is loaded under this name with a proper module spec
(origin and submodule search path in ``r-tucker_amd/``) and takes this module's place in ``sys.modules``, so
``importlib.reload``, ``inspect`` and relative imports behave as for any package."""
import importlib.util as _u
import os as _os
import sys as _sys

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "r-tucker_amd")
_spec = _u.spec_from_file_location(__name__, _os.path.join(_real, "__init__.py"), submodule_search_locations=[_real])
_mod = _u.module_from_spec(_spec)
_sys.modules[__name__] = _mod
_spec.loader.exec_module(_mod)
