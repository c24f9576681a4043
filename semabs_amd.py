"""Import alias: ``import semabs_amd`` loads the package kept in ``semantic-abstraction_amd/``.

The package directory carries the project's name (which has a hyphen and so is not a legal
Python identifier); this shim registers it in ``sys.modules`` under the importable name
``semabs_amd`` with the hyphenated directory as its submodule search path, so that
``import semabs_amd.clip`` / ``from semabs_amd.net import SemAbs3D`` work from the repo root.
"""
import importlib.util
import os
import sys

_PKG_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "semantic-abstraction_amd")
_spec = importlib.util.spec_from_file_location(
    "semabs_amd", os.path.join(_PKG_DIR, "__init__.py"), submodule_search_locations=[_PKG_DIR]
)
_mod = importlib.util.module_from_spec(_spec)
sys.modules["semabs_amd"] = _mod
_spec.loader.exec_module(_mod)
