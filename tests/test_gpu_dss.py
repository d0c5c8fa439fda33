"""GPU parity of the DSS baseline (dirichlet/dss/model.py:25-147) with the reference's trained checkpoint
(k = 30 updates, one weight set per update, alpha = 1e-3): HIP tile kernel vs the CPU oracle and the golden vectors.
Tolerance: the latent state is a sum of 30 corrections scaled by alpha = 1e-3 whose inputs (message sums ~1e2) cancel to
~1e-2: the fp32 ORACLE itself sits 1.6e-5 (rel-L2) from the fp64 evaluation, so the gate is 1e-4 vs the fp64 golden
(measured ~1e-5) and 5e-5 vs the fp32 oracle; a single update <= 2e-6."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, load_case, pkg, rel_l2
from oracle import psignn_oracle as orc

pytestmark = pytest.mark.gpu


def _net(dev):
    w = np.load(os.path.join(GOLDEN, "weights_dss.npz"))
    sd = {n: torch.from_numpy(w[n]) for n in w.files if n not in ("k", "alpha")}
    net = pkg("dss").DeepStatisticalSolver(dict(latent_dim=10, k=int(w["k"]), alpha=float(w["alpha"]), gamma=0.9, path_logs=None))
    net.load_state_dict(sd)
    return sd, net.to(dev).eval()


@pytest.mark.parametrize("name", ["original_dirichlet_s0", "hex13_dirichlet_s0"])
def test_dss_inference_parity(name, dev):
    sd, net = _net(dev)
    _, mesh = load_case(name)
    g = np.load(os.path.join(GOLDEN, f"dss_{name}.npz"))
    dss = pkg("dss")
    b = dss.to_dss_batch(mesh).to(dev)
    assert rel_l2(net.latent(b, 1), g["h_1"]) < 2e-6
    hk = net.latent(b)
    assert rel_l2(hk, g["h_k64"]) < 1e-4 and rel_l2(hk, g["h_k"]) < 5e-5, (rel_l2(hk, g["h_k64"]), rel_l2(hk, g["h_k"]))
    u = net.inference(b)
    assert u.shape == (mesh.num_nodes, 1) and u.is_cuda
    assert rel_l2(u, g["u_k64"]) < 1e-4
    assert torch.equal(hk, net.latent(b))   # reproducible
    res = float(net.residual_loss(u, b.edge_index, b.a_ij, b.b_prime))
    assert abs(res - float(g["res_trace"][-1])) < 2e-2 * float(g["res_trace"][-1])
    ob = orc.dss_batch(mesh)
    assert abs(float(orc.dss_residual_loss(u.cpu(), ob.edge_index, ob.a_ij, ob.b_prime)) - res) < 1e-3 * res
    with pytest.raises(pkg("_native").NativeError):
        net.latent(b, 31)    # no weights for a 31st update


def test_dss_forward_diagnostics(dev):
    """DeepStatisticalSolver.forward (model.py:59-95 / tests/model_dss.py:58-104): every decoded iterate and the loss traces;
    the tests/ spelling of the edge fields (edge_attr / edge_attr_norm) is accepted."""
    sd, net = _net(dev)
    _, mesh = load_case("original_dirichlet_s0")
    g = np.load(os.path.join(GOLDEN, "dss_original_dirichlet_s0.npz"))
    dss = pkg("dss")
    b = dss.to_dss_batch(mesh).to(dev)
    U, ld = net(b)
    k = net.config["k"]
    assert list(U) == [str(i) for i in range(k + 1)]
    assert set(ld) == {"train_loss", "residual_loss", "mse_loss", "mse_dirichlet_loss"}
    assert rel_l2(U[str(k)], g["u_k64"]) < 1e-4 and torch.equal(U[str(k)], net.inference(b))
    res = np.array([float(ld["residual_loss"][str(i)]) for i in range(1, k + 1)])
    assert np.allclose(res, g["res_trace"], rtol=2e-2)
    assert abs(float(ld["mse_loss"][str(k)]) - float(g["mse"])) < 1e-3 * float(g["mse"])
    alt = pkg("data").MeshData(x=b.x, sol=b.sol, edge_index=b.edge_index, edge_attr=b.a_ij, edge_attr_norm=b.a_ij_norm,
                               b_prime=b.b_prime, b_prime_norm=b.b_prime_norm, pos=b.pos, tags=b.tags)
    assert torch.equal(net.inference(alt), U[str(k)])
