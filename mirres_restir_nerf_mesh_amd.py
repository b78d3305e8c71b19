"""Import alias: the package directory is named `mirres-restir_nerf_mesh_amd` (not a Python identifier);
`import mirres_restir_nerf_mesh_amd` loads it as a regular package."""
import importlib.util as _u
import os as _os
import sys as _sys

_d = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "mirres-restir_nerf_mesh_amd")
_spec = _u.spec_from_file_location(__name__, _os.path.join(_d, "__init__.py"), submodule_search_locations=[_d])
_mod = _u.module_from_spec(_spec)
_sys.modules[__name__] = _mod
_spec.loader.exec_module(_mod)
