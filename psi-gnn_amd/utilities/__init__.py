from . import solver  # noqa: F401
