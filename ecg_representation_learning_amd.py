"""
Import shim: the package directory is named `ecg-representation-learning_amd/` (hyphen, as the
project layout prescribes), which Python cannot import by name.  This module loads that directory
as the package `ecg_representation_learning_amd` and replaces itself in `sys.modules`.
"""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'ecg-representation-learning_amd')
_spec = importlib.util.spec_from_file_location(
    __name__, os.path.join(_dir, '__init__.py'), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules[__name__] = _mod
_spec.loader.exec_module(_mod)
