"""Mixed (Dirichlet + Neumann) model family: same classes as ``model_psignn`` with the Neumann branch
(reference: ``mixed/psignn/model.py``, ``mixed/psignn/test/model_test.py``)."""
from . import model_psignn as _m


class ModelPSIGNN(_m.ModelPSIGNN):
    MIXED = True


class ModelPSIGNNIterative(_m.ModelPSIGNNIterative):
    MIXED = True


class ModelDEQDSS(_m.ModelDEQDSS):
    MIXED = True
