"""Import shim: the product package lives in the directory ``iad-r1_amd/`` (not a valid Python
identifier), and is importable as ``iadr1_amd``.  Importing this module replaces itself in
``sys.modules`` with the real package so ``import iadr1_amd.hip`` etc. work."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "iad-r1_amd")
_spec = importlib.util.spec_from_file_location("iadr1_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["iadr1_amd"] = _mod
_spec.loader.exec_module(_mod)
