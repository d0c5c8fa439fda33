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


def test_dss_training_step_gradients(dev):
    """loss.backward() on train_loss (BPTT over the 30 updates, one weight set each) vs autograd on the oracle evaluated in
    float64: loss and all 480 gradient tensors <= 5e-4 of the largest tensor's norm.  (The fp32 ORACLE is itself 3.6e-3 away
    from the fp64 one on phi_to_list.17 -- a ReLU mask of a cancelling pre-activation flips -- see the note at the top.)"""
    sd, net = _net(dev)
    _, mesh = load_case("hex13_dirichlet_s0")
    dss = pkg("dss")
    net.train()
    U, ld = net(dss.to_dss_batch(mesh).to(dev))
    assert ld["train_loss"].requires_grad
    ld["train_loss"].backward()
    k, alpha, gamma = net.config["k"], net.config["alpha"], net.config["gamma"]
    ob = orc.dss_batch(mesh)
    for f in list(vars(ob)):
        t = getattr(ob, f)
        if torch.is_tensor(t) and t.is_floating_point():
            setattr(ob, f, t.double())
    wl, wg, wres = orc.dss_training_step({n: t.double() for n, t in sd.items()}, ob, k, alpha, gamma)
    print("train_loss", float(ld["train_loss"].detach()), float(wl))
    assert abs(float(ld["train_loss"].detach()) - float(wl)) < 1e-3 * abs(float(wl))
    got = {n: p.grad for n, p in net.named_parameters()}
    scale = max(float(t.norm()) for t in wg.values())
    errs = {}
    for n, w in wg.items():
        gn = got[n]
        if gn is None:
            assert float(w.abs().max()) == 0.0, n    # e.g. decoder_list.0 is also used for U_0 only through H_0 = 0
            continue
        errs[n] = float((gn.cpu().double() - w).norm()) / max(float(w.norm()), 1e-3 * scale)
    print("worst", max(errs, key=errs.get), max(errs.values()), len(errs))
    assert max(errs.values()) < 5e-4, {n: e for n, e in errs.items() if e >= 5e-4}


def test_dss_trainer(dev, tmp_path):
    TrainModelDSS = pkg("training_class").TrainModelDSS
    torch.manual_seed(0)
    cfg = dict(latent_dim=10, k=5, alpha=1e-3, gamma=0.9, path_logs=str(tmp_path))
    dss = pkg("dss")
    net = dss.DeepStatisticalSolver(cfg).to(dev)
    meshes = [dss.to_dss_batch(load_case(n)[1]).to(dev) for n in ("hex13_dirichlet_s0", "original_dirichlet_s0")]
    tcfg = dict(loader_train=meshes, loader_val=meshes[:1], model=net, config_model=cfg, lr=0.01, path_ckpt=str(tmp_path),
                min_loss_save=1e9, max_epochs=5, gradient_clip=0.01)
    tr = TrainModelDSS(tcfg)
    tr.train_model()
    assert set(tr.hist_train) == {"loss", "residual_loss", "mse_loss"} and len(tr.hist_train["loss"]) == 5
    assert all(np.isfinite(tr.hist_train["loss"])) and tr.hist_train["loss"][-1] < tr.hist_train["loss"][0]
    ck = torch.load(tmp_path / "running_model.pt", weights_only=True)
    assert set(ck) == {"epoch", "hyperparameters", "state_dict", "hist_train", "hist_val", "opt", "training_time"}


def test_dss_union_batch(dev):
    """A PyG-style union batch (data.collate) in the DSS schema: the plan tiles graph by graph (the graph ids travel with the
    batch), and the union's latent state is the concatenation of the per-graph ones; DSS-schema graphs collate as well."""
    sd, net = _net(dev)
    data, dss = pkg("data"), pkg("dss")
    meshes = [load_case(n)[1] for n in ("hex13_dirichlet_s0", "original_dirichlet_s0", "hex26_dirichlet_s0")]
    union = dss.to_dss_batch(data.collate(meshes)).to(dev)
    hk = net.latent(union)
    parts = torch.cat([net.latent(dss.to_dss_batch(m).to(dev)) for m in meshes], dim=0)
    assert rel_l2(hk, parts) < 1e-6
    union2 = data.collate([dss.to_dss_batch(m) for m in meshes]).to(dev)    # collating graphs already in the DSS schema
    assert torch.equal(union2.b_prime_norm, union.b_prime_norm) and torch.equal(union2.a_ij_norm, union.a_ij_norm)
    assert rel_l2(net.latent(union2), parts) < 1e-6
