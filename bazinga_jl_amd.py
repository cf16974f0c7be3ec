"""Importable alias of the package directory ``bazinga.jl_amd/`` (a dot is not legal in a
Python package name).  ``import bazinga_jl_amd`` gives the package itself."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "bazinga.jl_amd")
_spec = importlib.util.spec_from_file_location("bazinga_jl_amd", os.path.join(_dir, "__init__.py"),
                                               submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["bazinga_jl_amd"] = _mod
_spec.loader.exec_module(_mod)
