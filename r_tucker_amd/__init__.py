"""Import shim: the product package lives in the directory ``r-tucker_amd/`` (the
name the project layout prescribes), which is not a valid Python identifier.
``import r_tucker_amd`` resolves here and continues there."""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "r-tucker_amd")
__path__.insert(0, _real)
with open(_os.path.join(_real, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"))
del _f
