"""Import shim: the package directory is named ``fast-plaid_amd`` (not a valid Python
identifier), so ``import fast_plaid_amd`` loads it from here under the underscore name."""
import importlib.util
import os
import sys

_d = os.path.join(os.path.dirname(os.path.abspath(__file__)), "fast-plaid_amd")
_spec = importlib.util.spec_from_file_location(
    "fast_plaid_amd", os.path.join(_d, "__init__.py"), submodule_search_locations=[_d])
_m = importlib.util.module_from_spec(_spec)
sys.modules["fast_plaid_amd"] = _m
_spec.loader.exec_module(_m)
