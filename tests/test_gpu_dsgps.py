"""GPU parity of the DS-GPS baseline (dirichlet/dsgps/model.py:28-176) with the reference's trained
checkpoint (30 steps, gamma 0.9): HIP tile kernels vs the CPU oracle and the committed golden vectors.
Tolerance: k = 30 recurrent fp32 steps, each re-associated like f (<= 2e-6 per step) -> <= 1e-5 rel-L2 on H_k / U_k."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, load_case, pkg, rel_l2
from oracle import psignn_oracle as orc

pytestmark = pytest.mark.gpu


def _net(dev, k=None, mixed=False):
    w = np.load(os.path.join(GOLDEN, "weights_dsgps_mixed.npz" if mixed else "weights_dsgps.npz"))
    sd = {n: torch.from_numpy(w[n]) for n in w.files if n != "k"}
    cfg = dict(latent_dim=10, k=int(w["k"]) if k is None else k, alpha=1e-3, gamma=0.9, path_logs=None)
    if mixed:
        cfg.update(lamb=0.0, bc="mixed")
    net = pkg("dsgps").ModelDSGPS(cfg)
    net.load_state_dict(sd)
    return sd, net.to(dev).eval()


@pytest.mark.parametrize("name", ["original_dirichlet_s0", "hex13_dirichlet_s0", "hex26_dirichlet_s0"])
def test_dsgps_inference_parity(name, dev):
    sd, net = _net(dev)
    _, mesh = load_case(name)
    g = np.load(os.path.join(GOLDEN, f"dsgps_{name}.npz"))
    md = mesh.to(dev)
    u = net.inference(md)
    assert u.shape == (mesh.num_nodes, 1) and u.is_cuda
    assert rel_l2(u, g["u_k"]) < 1e-5 and rel_l2(u, g["u_k64"]) < 1e-5
    eng = pkg("engine")
    h0 = net.autoencoder.encoder(md.x)
    w = net.packed(dev)
    plan = eng.plan_for(md)
    assert plan.tiled
    # a single update against the oracle and the golden vector, the whole recurrence against fp32 and fp64 goldens
    assert rel_l2(eng.dsgps_forward(plan, w, h0, md.prb_data, 1), g["h_1"]) < 2e-6
    hk = eng.dsgps_forward(plan, w, h0, md.prb_data, net.config["k"])
    assert rel_l2(hk, g["h_k"]) < 1e-5 and rel_l2(hk, g["h_k64"]) < 1e-5
    assert torch.equal(eng.dsgps_forward(plan, w, h0, md.prb_data, 0), h0)
    assert torch.equal(hk, eng.dsgps_forward(plan, w, h0, md.prb_data, net.config["k"]))   # reproducible
    # Dirichlet rows are the encoder rows of every iterate (model.py:152)
    idx = torch.where(md.tags.reshape(-1) == 1)[0]
    assert torch.equal(hk[idx], h0[idx])


def test_dsgps_forward_diagnostics(dev):
    """ModelDSGPS.forward: every decoded iterate and the per-step loss dictionaries (model.py:48-128)."""
    sd, net = _net(dev)
    _, mesh = load_case("original_dirichlet_s0")
    g = np.load(os.path.join(GOLDEN, "dsgps_original_dirichlet_s0.npz"))
    U, ld = net(mesh.to(dev))
    k = net.config["k"]
    assert list(U) == [str(i) for i in range(k + 1)]
    assert set(ld) == {"train_loss", "residual_loss", "encoder_loss", "autoencoder_loss", "mse_dirichlet", "mse_loss"}
    assert rel_l2(U[str(k)], g["u_k"]) < 1e-5
    res = np.array([float(ld["residual_loss"][str(i)]) for i in range(k + 1)])
    mse = np.array([float(ld["mse_loss"][str(i)]) for i in range(k + 1)])
    assert np.allclose(res, g["res_trace"], rtol=2e-3) and np.allclose(mse, g["mse_trace"], rtol=1e-4)
    assert abs(float(ld["mse_dirichlet"][str(k)]) - float(g["mse_dirichlet"])) < 1e-3 * float(g["mse_dirichlet"])
    assert torch.isfinite(ld["train_loss"])


def test_dsgps_more_steps_and_other_tiles(dev):
    """k beyond the trained horizon (the reference evaluates k = 150 / 200 with the same weights) and other tile sizes."""
    sd, net = _net(dev, k=100)
    _, mesh = load_case("hex13_dirichlet_s0")
    md = mesh.to(dev)
    u, h = orc.dsgps_inference(sd, mesh, 100)
    assert rel_l2(net.inference(md), u) < 5e-5
    eng = pkg("engine")
    h0 = net.autoencoder.encoder(md.x)
    base = eng.dsgps_forward(eng.plan_for(md), net.packed(dev), h0, md.prb_data, 30)
    for tt in (32, 100):
        assert torch.equal(eng.dsgps_forward(eng.MeshPlan(md, tile_target=tt), net.packed(dev), h0, md.prb_data, 30), base)
    with pytest.raises(pkg("_native").NativeError):
        eng.dsgps_forward(eng.MeshPlan(md, tile_target=-1), net.packed(dev), h0, md.prb_data, 1)


def test_dsgps_mixed_family(dev):
    """mixed/dsgps/model.py:50-95 with its checkpoint (k = 50): Neumann rows replaced by update_neumann, Dirichlet rows
    from H_0.  The oracle's restatement is pinned only by behaviour (residual 1.2 -> 1.7e-3 on the synthetic mixed mesh:
    the trained recurrence works); bit-level parity of the reference is unpinned (torch_geometric absent)."""
    w = np.load(os.path.join(GOLDEN, "weights_dsgps_mixed.npz"))
    sd = {n: torch.from_numpy(w[n]) for n in w.files if n != "k"}
    k = int(w["k"])
    net = pkg("dsgps").ModelDSGPS(dict(latent_dim=10, k=k, alpha=1e-3, gamma=0.9, lamb=0.0, path_logs=None, bc="mixed"))
    assert set(net.state_dict()) == set(sd)
    net.load_state_dict(sd)
    net = net.to(dev).eval()
    _, mesh = load_case("hex13_mixed_s1")
    g = np.load(os.path.join(GOLDEN, "dsgps_hex13_mixed_s1.npz"))
    md = mesh.to(dev)
    eng = pkg("engine")
    h0 = net.autoencoder.encoder(md.x)
    plan, wf = eng.plan_for(md), net.packed(dev)
    assert rel_l2(eng.dsgps_forward(plan, wf, h0, md.prb_data, 1, md.unit_normal_vector), g["h_1"]) < 2e-6
    hk = eng.dsgps_forward(plan, wf, h0, md.prb_data, k, md.unit_normal_vector)
    assert rel_l2(hk, g["h_k"]) < 1e-5 and rel_l2(hk, g["h_k64"]) < 1e-5
    u = net.inference(md)
    assert rel_l2(u, g["u_k64"]) < 1e-5
    idx_d = torch.where(md.tags[:, 1] == 1)[0]
    assert torch.equal(hk[idx_d], h0[idx_d])
    U, ld = net(md)
    assert rel_l2(U[str(k)], g["u_k"]) < 1e-5
    res = np.array([float(ld["residual_loss"][str(i)]) for i in range(k + 1)])
    assert np.allclose(res, g["res_trace"], rtol=5e-3)
    assert float(g["res_trace"][-1]) < 1e-2 * float(g["res_trace"][1])


@pytest.mark.parametrize("name", ["original_dirichlet_s0", "hex13_dirichlet_s0", "hex26_dirichlet_s0", "hex13_mixed_s1"])
def test_dsgps_step_backward_parity(name, dev):
    """Backward of one recurrent update (what loss.backward() runs per unrolled step, dirichlet/dsgps/model.py:72-89) vs
    autograd on the oracle's ``dsgps_step``: all 14 gradients <= 2e-5 of the largest tensor's norm, d/dh <= 2e-5."""
    mixed = name.endswith("mixed_s1")
    sd, net = _net(dev, mixed=mixed)
    g, mesh = load_case(name)
    gd = np.load(os.path.join(GOLDEN, f"dsgps_{name}.npz"))
    eng = pkg("engine")
    md = mesh.to(dev)
    nrm = md.unit_normal_vector if mixed else None
    h = torch.from_numpy(gd["h_1"])
    h0 = orc.encoder(sd, mesh.x).detach()
    wv = torch.randn(h.shape, generator=torch.Generator().manual_seed(4))
    names = [k for k in sd if not k.startswith(("autoencoder", "laynorm"))]
    p = {k: (t.clone().requires_grad_(True) if k in names else t) for k, t in sd.items()}
    hh, hz = h.clone().requires_grad_(True), h0.clone().requires_grad_(True)
    gr = torch.autograd.grad(orc.dsgps_step(p, hh, hz, mesh), [p[k] for k in names] + [hh, hz], wv)
    want = dict(zip(names, gr[:-2]))
    wf, wg = eng.pack_dsgps_train(sd, dev)
    got, dh = eng.dsgps_step_backward(eng.plan_for(md), wf, wg, h.to(dev), md.prb_data, wv.to(dev), nrm)
    assert set(got) == set(want)
    scale = max(float(t.norm()) for t in want.values())
    errs = {k: float((got[k].cpu() - want[k]).norm()) / max(float(want[k].norm()), 1e-4 * scale) for k in want}
    assert max(errs.values()) < 2e-5, errs
    assert rel_l2(dh, gr[-2]) < 2e-5
    mask = (mesh.tags[:, 1:2] if mixed else mesh.tags.reshape(-1, 1)) == 1
    assert torch.equal(gr[-1], torch.where(mask, wv, torch.zeros_like(wv)))   # the Dirichlet rows' share goes to H_0
    got2, _ = eng.dsgps_step_backward(eng.plan_for(md), wf, wg, h.to(dev), md.prb_data, wv.to(dev), nrm)
    assert all(torch.equal(got[k], got2[k]) for k in got)


@pytest.mark.parametrize("name", ["hex13_dirichlet_s0", "hex13_mixed_s1"])
def test_dsgps_training_step_gradients(name, dev):
    """loss.backward() on train_loss through the HIP model (BPTT over k = 30 / 50 updates) vs autograd on the oracle's restated
    forward: loss <= 1e-4, all gradients <= 5e-4 of the largest tensor's norm (30 chained fp32 VJPs; measured 7e-5)."""
    sd, net = _net(dev, mixed=name.endswith("mixed_s1"))
    _, mesh = load_case(name)
    net.train()
    U, ld = net(mesh.to(dev))
    assert ld["train_loss"].requires_grad
    ld["train_loss"].backward()
    k, gamma = net.config["k"], net.config["gamma"]
    wl, wg, wres, wmse = orc.dsgps_training_step(sd, mesh, k, gamma)
    print("train_loss", float(ld["train_loss"].detach()), float(wl))
    assert abs(float(ld["train_loss"].detach()) - float(wl)) < 1e-4 * abs(float(wl))
    assert abs(float(ld["residual_loss"][str(k)].detach()) - wres) < 1e-3 * wres
    got = {n: p.grad for n, p in net.named_parameters()}
    scale = max(float(t.norm()) for t in wg.values())
    errs = {}
    for n, w in wg.items():
        if n.startswith("laynorm"):        # declared by the reference, unused in its forward: no gradient
            assert got[n] is None or float(got[n].abs().max()) == 0.0
            continue
        errs[n] = float((got[n].cpu() - w).norm()) / max(float(w.norm()), 1e-4 * scale)
    print("worst", max(errs, key=errs.get), max(errs.values()))
    assert max(errs.values()) < 5e-4, {n: e for n, e in errs.items() if e >= 5e-4}


def test_dsgps_trainer(dev, tmp_path):
    """TrainModelDSGPS (dirichlet/dsgps/training_class.py surface): a few epochs from random initialisation lower the loss;
    checkpoint keys as in the reference; resume."""
    TrainModelDSGPS = pkg("training_class").TrainModelDSGPS
    torch.manual_seed(0)
    cfg = dict(latent_dim=10, k=10, alpha=1e-3, gamma=0.9, path_logs=str(tmp_path))
    net = pkg("dsgps").ModelDSGPS(cfg).to(dev)
    meshes = [load_case(n)[1].to(dev) for n in ("hex13_dirichlet_s0", "original_dirichlet_s0")]
    tcfg = dict(loader_train=meshes, loader_val=meshes[:1], model=net, config_model=cfg, lr=0.01, path_ckpt=str(tmp_path),
                min_loss_save=1e9, max_epochs=6, gradient_clip=0.01)
    tr = TrainModelDSGPS(tcfg)
    tr.train_model()
    assert len(tr.hist_train["loss"]) == 6 and set(tr.hist_train) == {"loss", "residual_loss", "encoder_loss", "autoencoder_loss", "mse_loss"}
    assert all(np.isfinite(tr.hist_train["loss"])) and tr.hist_train["loss"][-1] < tr.hist_train["loss"][0]
    ck = torch.load(tmp_path / "running_model.pt", weights_only=True)
    assert set(ck) == {"epoch", "hyperparameters", "state_dict", "hist_train", "hist_val", "opt_deq", "training_time"}
    tr2 = TrainModelDSGPS(dict(tcfg, model=pkg("dsgps").ModelDSGPS(cfg).to(dev)))
    tr2.load_model(str(tmp_path / "running_model.pt"))
    assert tr2.hist_train == tr.hist_train
