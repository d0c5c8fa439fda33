"""Dirichlet model family (reference: ``dirichlet/psignn/model.py``, ``tests/model_psignn.py``)."""
from .model_psignn import ModelDEQDSS, ModelPSIGNN, ModelPSIGNNIterative  # noqa: F401
