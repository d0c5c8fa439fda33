import importlib
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pkg(name=""):
    return importlib.import_module("psi-gnn_amd" + ("." + name if name else ""))


def load_weights(kind):
    w = np.load(os.path.join(GOLDEN, f"weights_{kind}.npz"))
    return {k: torch.from_numpy(w[k]) for k in w.files}


def load_case(name):
    """(golden dict, MeshData built from the stored input tensors)."""
    g = dict(np.load(os.path.join(GOLDEN, name + ".npz")))
    MeshData = pkg("data").MeshData
    fields = {k[3:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("in_")}
    return g, MeshData(**fields)


CASES = {
    "original_dirichlet_s0": "dirichlet",
    "original_dirichlet_s1": "dirichlet",
    "hex13_dirichlet_s0": "dirichlet",
    "hex13_mixed_s1": "mixed",
    "hex26_dirichlet_s0": "dirichlet",
}


def rel_l2(a, b):
    a = torch.as_tensor(a, dtype=torch.float64).reshape(-1)
    b = torch.as_tensor(b, dtype=torch.float64).reshape(-1)
    return float((a.cpu() - b.cpu()).norm() / b.cpu().norm())


@pytest.fixture(scope="session")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def load_circlelarge():
    """The reference's own 5 329-node disc (tests/special_geo/mesh_files/circlelarge, decoded by
    oracle/make_golden_circlelarge.py) with the problem its notebooks solved: (MeshData, recorded rows dict)."""
    import json
    z = np.load(os.path.join(GOLDEN, "mesh_circlelarge.npz"))
    hm = pkg("data.hexmesh")
    mesh = hm.make_from_triangulation(z["pos"], z["tri"].astype(np.int64), z["dirichlet"], radius=float(z["radius"]),
                                      coeffs=(z["param_f"], z["param_g"]))
    band = json.load(open(os.path.join(GOLDEN, "circlelarge_band.json")))
    return mesh, band


@pytest.fixture
def knobs(monkeypatch):
    """``knobs(PSIGNN_X="...")``: set PSIGNN_* run-time knobs of libpsignn_hip.so for this test (alternative kernel forms: MFMA
    stage 1, Hilbert tiling, ...) and make the library re-read them; everything is restored, and re-read, afterwards."""
    nat = pkg("_native")

    def apply(**env):
        for k, v in env.items():
            if v is None:
                monkeypatch.delenv(k, raising=False)
            else:
                monkeypatch.setenv(k, str(v))
        nat.lib().psignn_reload_knobs()
    yield apply
    monkeypatch.undo()
    nat.lib().psignn_reload_knobs()
