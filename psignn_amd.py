"""Importable alias of the ``psi-gnn_amd`` package directory (a hyphen is not a valid identifier)."""
import importlib
import os
import sys

_here = os.path.dirname(os.path.abspath(__file__))
if _here not in sys.path:
    sys.path.insert(0, _here)
_pkg = importlib.import_module("psi-gnn_amd")
sys.modules[__name__] = _pkg
