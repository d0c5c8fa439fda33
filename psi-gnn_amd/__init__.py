"""MI355X-native PSI-GNN fixed-point inference.

Import name: the directory is ``psi-gnn_amd``; ``import psignn_amd`` (alias module at the repo root) or
``importlib.import_module("psi-gnn_amd")`` both load this package.
"""
import sys as _sys

from . import data  # noqa: F401


def install_reference_aliases():
    """Register ``utilities`` / ``utilities.solver`` / ``model_psignn`` / ``model_dsgps`` / ``model_dss`` in ``sys.modules``
    (the module names the reference's ``tests/`` scripts import: test_multiple.py:31-33, spec_geo_2.py:32-34).

    Reference checkpoints pickle ``config["solver"]`` by its qualified name ``utilities.solver.broyden``
    (training_class.py:60-66) and reference scripts do ``from utilities import solver`` /
    ``import model_psignn``; with the aliases installed both resolve to this package.
    """
    from . import utilities, model_psignn, dsgps, dss
    from .utilities import solver
    _sys.modules.setdefault("utilities", utilities)
    _sys.modules.setdefault("utilities.solver", solver)
    _sys.modules.setdefault("model_psignn", model_psignn)
    _sys.modules.setdefault("model_dsgps", dsgps)
    _sys.modules.setdefault("model_dss", dss)


def load_reference_checkpoint(path, map_location="cpu"):
    """``torch.load(weights_only=True)`` of a reference ``*.pt`` checkpoint (no code from the file runs)."""
    import torch
    install_reference_aliases()
    from .utilities import solver
    # the pickle names the function by the reference's module path; map that path to our implementation
    fns = [getattr(solver, n) for n in ("broyden", "anderson", "forward_iteration", "newton")]
    # the reference pickles the solver as utilities.solver.<name>; checkpoints written by training_class.TrainModel here carry
    # this package's own module path
    torch.serialization.add_safe_globals(fns + [(f, f"utilities.solver.{f.__name__}") for f in fns])
    return torch.load(path, map_location=map_location, weights_only=True)
