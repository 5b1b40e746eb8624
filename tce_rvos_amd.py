"""Import shim: the package directory is `tce-rvos_amd/` (not a valid identifier); this makes it
importable as `tce_rvos_amd` by loading its __init__.py under that name."""
import importlib.util as _ilu
import os as _os
import sys as _sys

_d = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "tce-rvos_amd")
_spec = _ilu.spec_from_file_location("tce_rvos_amd", _os.path.join(_d, "__init__.py"),
                                     submodule_search_locations=[_d])
_mod = _ilu.module_from_spec(_spec)
_sys.modules["tce_rvos_amd"] = _mod
_spec.loader.exec_module(_mod)
