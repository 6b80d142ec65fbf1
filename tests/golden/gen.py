"""Synthetic parameters / queries of the fixtures.  The generators live in the package
(``r-tucker_amd/synthetic.py``: ``bench.py`` and ``smoke()`` use the same inputs); the tests, the tools and
the golden-vector generators import them under this historical name.  Loaded by file path, so this
works before the package is importable (numpy only)."""
import importlib.util as _u
import os as _os

_p = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__)))),
                   "r-tucker_amd", "synthetic.py")
_spec = _u.spec_from_file_location("_rtk_synthetic", _p)
_m = _u.module_from_spec(_spec)
_spec.loader.exec_module(_m)
digest, make_params, make_planted_params, make_queries = _m.digest, _m.make_params, _m.make_planted_params, _m.make_queries
