"""Import shim: the product package lives in the directory ``occlusions-4d_amd/``
(name fixed by the build contract, not importable because of the hyphen).
``import occlusions4d_amd`` loads that directory as a regular package under the
importable name ``occlusions4d_amd``."""
import importlib.util
import os
import sys

_root = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'occlusions-4d_amd')
_spec = importlib.util.spec_from_file_location(
    __name__, os.path.join(_root, '__init__.py'), submodule_search_locations=[_root])
_pkg = importlib.util.module_from_spec(_spec)
sys.modules[__name__] = _pkg
_spec.loader.exec_module(_pkg)
