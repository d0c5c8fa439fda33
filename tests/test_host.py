"""CPU: host-side logic and the C ABI surface (no compute calls — there is no GPU here)."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from conftest import ROOT, load_weights, pkg


def test_library_exports_every_declared_symbol():
    nat = pkg("_native")
    hdr = open(os.path.join(ROOT, "include", "psignn_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(psignn_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 20
    lib = ctypes.CDLL(nat.LIB_PATH)
    missing = [s for s in sorted(declared) if not hasattr(lib, s)]
    assert not missing, missing
    # the ctypes table binds exactly the declared functions
    assert set(nat.SIGNATURES) == declared
    # ... with as many arguments as the header declares (a drifted signature would corrupt the call silently)
    for name, params in re.findall(r"\b(psignn_[a-z0-9_]+)\s*\(([^()]*)\)\s*;", hdr):
        params = params.strip()
        n = 0 if params in ("", "void") else params.count(",") + 1
        assert len(nat.SIGNATURES[name][1]) == n, (name, n, len(nat.SIGNATURES[name][1]))


def test_pure_host_entry_points():
    nat = pkg("_native")
    lib = nat.lib()
    assert lib.psignn_version() >= 100
    assert lib.psignn_weights_size(0, 1) == 64 + 2 * 350 + (10 * 32 + 120) + 244 + 950
    assert lib.psignn_weights_size(1, 1) == 64 + 2 * 350 + (10 * 33 + 120) + 244 + 350 + (10 * 25 + 120) + 112 + 950 + 620
    assert lib.psignn_weights_size(0, 0) == -1
    assert lib.psignn_plan_num_nodes(None) == -1


def test_weight_packing_layout():
    eng = pkg("engine")
    sd = load_weights("dirichlet")
    flat = eng.pack_weights(sd)
    assert flat.numel() == 1448 + 950
    P = "deqdss.f."
    assert torch.equal(flat[0:10], sd[P + "laynorm.weight"])
    assert torch.equal(flat[20:52], sd[P + "alpha.0.weight"].reshape(-1))
    assert torch.equal(flat[64:64 + 230], sd[P + "phi_to_list.0.mlp.mlp.0.weight"].reshape(-1))
    assert torch.equal(flat[64 + 350:64 + 350 + 230], sd[P + "phi_from_list.0.mlp.mlp.0.weight"].reshape(-1))
    assert torch.equal(flat[64 + 700:64 + 700 + 320], sd[P + "update_list.0.mlp.0.weight"].reshape(-1))
    sdm = load_weights("mixed")
    flatm = eng.pack_weights(sdm)
    assert flatm.numel() == 2290 + 950 + 620
    assert torch.equal(flatm[-1570 - 112 - 370:-1570 - 112 - 120], sdm[P + "update_neumann.mlp.0.weight"].reshape(-1))
    # transposed section: [k][o] copy of W1j of Phi_to starts the tile kernel's block
    assert torch.equal(flat[1448:1548].reshape(10, 10), sd[P + "phi_to_list.0.mlp.mlp.0.weight"][:, 10:20].t())
    U1, W2 = sd[P + "update_list.0.mlp.0.weight"].double(), sd[P + "phi_to_list.0.mlp.mlp.2.weight"].double()
    assert torch.allclose(flat[64 + 700 + 440:64 + 700 + 540].double(), (U1[:, 10:20] @ W2).reshape(-1), rtol=1e-6, atol=1e-7)
    assert eng.is_mixed_state_dict(sdm) and not eng.is_mixed_state_dict(sd)


@pytest.mark.parametrize("kind,mod", [("dirichlet", "model_psignn"), ("mixed", "mixed")])
def test_reference_state_dict_loads(kind, mod):
    m = pkg(mod)
    net = m.ModelPSIGNN(dict(latent_dim=10, n_layers=1, fw_tol=1e-5, fw_thres=500))
    sd = load_weights(kind)
    assert set(net.state_dict()) == set(sd)
    net.load_state_dict(sd, strict=True)
    n_params = sum(p.numel() for p in net.parameters())
    assert n_params == (1444 if kind == "dirichlet" else 2175)  # model_config.csv:88 / :96


def test_no_cpu_fallback():
    """CPU tensors are rejected loudly; nothing silently routes around the HIP library."""
    nat = pkg("_native")
    solver = pkg("utilities.solver")
    with pytest.raises(nat.NativeError):
        solver.broyden(lambda x: 0.5 * x, torch.zeros(4, 10), threshold=5, eps=1e-3)
    data = pkg("data")
    mesh = data.make_hex_problem(3, seed=0)
    net = pkg("model_psignn").ModelPSIGNN(dict(latent_dim=10, n_layers=1))
    with pytest.raises(nat.NativeError):
        net(mesh)
    import glob
    files = glob.glob(os.path.join(ROOT, "psi-gnn_amd", "**", "*.py"), recursive=True) + [os.path.join(ROOT, "psignn_amd.py")]
    assert len(files) > 15
    for f in files:     # nothing in the product imports or mentions the CPU oracle
        assert "oracle" not in open(f).read(), f


def test_hex_mesh_schema_and_sizes():
    data = pkg("data")
    for n, N in ((13, 547), (26, 2107)):
        m = data.make_hex_problem(n, seed=0)
        assert m.num_nodes == N == 3 * n * n + 3 * n + 1
        assert int((m.tags == 1).sum()) == 6 * n
        r, c = m.edge_index
        assert bool(((r[1:] > r[:-1]) | ((r[1:] == r[:-1]) & (c[1:] > c[:-1]))).all())  # row-major, unique
        assert int((r == c).sum()) == N  # full diagonal
        bd = (m.tags[:, 0] == 1).nonzero()[:, 0]
        assert bool(torch.isin(r, bd).logical_and(r != c).sum() == 0)  # Dirichlet rows are identity rows
        A = torch.sparse_coo_tensor(m.edge_index, m.a_ij[:, 0].double(), (N, N))
        assert float((torch.sparse.mm(A, m.sol.double()) - m.y.double()).abs().max()) < 1e-4
    assert data.make_hex_problem(13, seed=0).edge_index.shape[1] == 3361
    mm = data.make_hex_problem(13, seed=1, mixed=True)
    assert mm.tags.shape == (547, 3) and bool((mm.tags.sum(1) == 1).all())
    assert mm.tags.sum(0).tolist() == [469.0, 42.0, 36.0]
    assert mm.prb_data.shape == (547, 3) and mm.unit_normal_vector.shape == (547, 2)
    assert data.hex_n_for_nodes(1_000_000) == 577 and data.hex_n_for_nodes(100_000) == 183


def test_collate_disjoint_union():
    data = pkg("data")
    a, b = data.make_hex_problem(3, seed=0), data.make_hex_problem(4, seed=1)
    u = data.collate([a, b])
    assert u.num_nodes == a.num_nodes + b.num_nodes
    assert u.edge_index.shape[1] == a.num_edges + b.num_edges
    assert int(u.edge_index[:, a.num_edges:].min()) == a.num_nodes
    assert u.batch.tolist() == [0] * a.num_nodes + [1] * b.num_nodes
    assert u.ptr.tolist() == [0, a.num_nodes, u.num_nodes]


def test_plan_reference_numpy():
    """The numpy statement of the plan's canonical CSR/CSC order used by the GPU bit-exact test."""
    from plan_ref import plan_reference
    ei = np.array([[0, 1, 2, 2, 1, 0, 2], [1, 0, 2, 0, 2, 0, 1]])
    ref = plan_reference(ei, 3)
    assert ref["csr_ptr"].tolist() == [0, 1, 3, 5]
    assert ref["csr_nbr"].tolist() == [1, 0, 2, 0, 1]
    assert ref["csr_eid"].tolist() == [0, 1, 4, 3, 6]
    assert ref["csc_ptr"].tolist() == [0, 2, 4, 5]
    assert ref["csc_nbr"].tolist() == [1, 2, 0, 2, 1]


def test_reader_reproduces_the_reference_schema():
    """utilities/reader.py on raw arrays (A, b, sol, prb_data, tags, coordinates, distance) == the tensors our
    generator emits for the same problem (same normalisation constants and initial guess, reader.py:73-110)."""
    import scipy.sparse as sp
    data, reader = pkg("data"), pkg("utilities.reader")
    hm = pkg("data.hexmesh")
    for mixed in (False, True):
        m = data.make_hex_problem(5, seed=3, mixed=mixed)
        N = m.num_nodes
        r, c = m.edge_index.numpy()
        A = sp.csr_matrix((m.a_ij[:, 0].double().numpy(), (r, c)), shape=(N, N))
        pos = m.pos.double().numpy()
        d = pos[r] - pos[c]
        dist = np.concatenate([d, np.sqrt((d ** 2).sum(1, keepdims=True))], axis=1)
        if mixed:
            prb_raw = m.prb_data.double().numpy() * hm.MIX_PRB_STD + hm.MIX_PRB_MEAN
            nrm_raw = m.unit_normal_vector.double().numpy() * hm.MIX_NRM_STD + hm.MIX_NRM_MEAN
        else:
            prb_raw = m.prb_data.double().numpy() * hm.DIR_PRB_STD + hm.DIR_PRB_MEAN
            nrm_raw = None
        # shuffle the distance rows the way an older scipy's column-major find() would have produced them
        got = reader.from_arrays(A, m.y.numpy(), m.sol.numpy(), prb_raw, m.tags.numpy(), pos, dist, nrm_raw)
        assert torch.equal(got.edge_index, m.edge_index)
        for k in ("x", "y", "sol", "tags", "a_ij"):
            assert torch.allclose(getattr(got, k), getattr(m, k), atol=1e-6), k
        assert torch.allclose(got.edge_attr, m.edge_attr, atol=2e-5)
        assert torch.allclose(got.prb_data, m.prb_data, atol=1e-5)
        if mixed:
            assert torch.allclose(got.unit_normal_vector, m.unit_normal_vector, atol=1e-5)
    assert reader.split_indices(10, "train") == [0, 1, 2, 3, 4, 5] and reader.split_indices(10, "val") == [6, 7]
    assert reader.split_indices(10, "test") == [8, 9]


def test_raw_dir_loader_never_unpickles_by_default(tmp_path):
    """The reference's raw dataset files are pickled object arrays (parity of the reader on such files is UNPINNED: the
    reference ships none).  load_raw_dir must not unpickle unless the caller opts in."""
    import pytest
    import scipy.sparse as sp
    data, reader = pkg("data"), pkg("utilities.reader")
    hm = pkg("data.hexmesh")
    raw = tmp_path / "data"
    raw.mkdir()
    cols = {n: [] for n in reader.RAW_FILES}
    for seed in range(5):
        m = data.make_hex_problem(3, seed=seed)
        N = m.num_nodes
        r, c = m.edge_index.numpy()
        pos = m.pos.double().numpy()
        d = pos[r] - pos[c]
        vals = [sp.csr_matrix((m.a_ij[:, 0].double().numpy(), (r, c)), shape=(N, N)), m.y.numpy(), m.sol.numpy(),
                m.prb_data.double().numpy() * hm.DIR_PRB_STD + hm.DIR_PRB_MEAN, m.tags.numpy(), pos,
                np.concatenate([d, np.sqrt((d ** 2).sum(1, keepdims=True))], axis=1)]
        for n, v in zip(reader.RAW_FILES, vals):
            cols[n].append(v)
    for n, v in cols.items():
        a = np.empty(len(v), dtype=object)
        a[:] = v
        np.save(raw / n, a, allow_pickle=True)      # the reference's np.save of a Python list (generate_data.py:104-128)
    with pytest.raises(ValueError, match="allow_pickle"):
        reader.load_raw_dir(str(tmp_path), mode="all")
    got = reader.load_raw_dir(str(tmp_path), mode="all", allow_pickle=True)   # our own files, written above
    assert len(got) == 5 and torch.equal(got[2].edge_index, data.make_hex_problem(3, seed=2).edge_index)
    assert len(reader.load_raw_dir(str(tmp_path), mode="test", allow_pickle=True)) == 1


def test_dsgps_state_dict_and_packing():
    """ModelDSGPS has the reference's module tree (dirichlet/dsgps/model.py:28-45) and packs to the kernel layout."""
    w = np.load(os.path.join(os.path.dirname(__file__), "golden", "weights_dsgps.npz"))
    sd = {k: torch.from_numpy(w[k]) for k in w.files if k != "k"}
    net = pkg("dsgps").ModelDSGPS(dict(latent_dim=10, k=int(w["k"]), alpha=1e-3, gamma=0.9, path_logs=None))
    assert set(net.state_dict()) == set(sd)
    net.load_state_dict(sd)
    flat = pkg("engine").pack_dsgps(net.state_dict())
    assert flat.numel() == 1690
    Wz = sd["z_k.mlp.0.weight"]
    assert torch.equal(flat[700:1020].reshape(32, 10), Wz.t())
    At = sd["phi_to.mlp.mlp.0.weight"][:, 20:23]
    assert torch.equal(flat[400:430].reshape(3, 10), At.t() * torch.tensor([-1.0, -1.0, 1.0])[:, None])


def test_loader_stand_ins():
    """DataListLoader / DataLoader / DataParallel keep the calling conventions of the PyG pieces the reference's
    scripts use (main.py:70-78,106; test_func.py:134-140)."""
    data, loader = pkg("data"), pkg("loader")
    ds = [data.make_hex_problem(3, seed=s) for s in range(5)]
    ll = loader.DataListLoader(ds, batch_size=2, shuffle=False)
    chunks = list(ll)
    assert len(ll) == 3 and [len(c) for c in chunks] == [2, 2, 1] and chunks[0][1] is ds[1]
    dl = list(loader.DataLoader(ds, batch_size=2))
    assert dl[0].num_graphs == 2 and dl[0].num_nodes == 2 * ds[0].num_nodes and dl[2].num_graphs == 1
    sh = loader.DataListLoader(ds, batch_size=5, shuffle=True, generator=torch.Generator().manual_seed(0))
    assert sorted(id(g) for g in next(iter(sh))) == sorted(id(g) for g in ds)

    class Probe(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.w = torch.nn.Parameter(torch.zeros(1))

        def forward(self, batch):
            return batch.num_nodes, getattr(batch, "num_graphs", 1)

    dp = loader.DataParallel(Probe())
    assert dp.module is not None
    assert dp([ds[0]]) == (ds[0].num_nodes, 1)
    assert dp(ds[:3]) == (3 * ds[0].num_nodes, 3)


def test_baseline_training_layouts_round_trip():
    """The weight buffers handed to psignn_dsgps_step_backward / psignn_dss_step_backward use the layout their gradients come
    back in: unpacking a packed weight buffer as if it were a gradient returns the state dict's tensors (both DS-GPS
    families, DSS with its (10, 21) first Phi layers padded to 23 columns)."""
    eng, nat = pkg("engine"), pkg("_native")
    gold = os.path.join(ROOT, "tests", "golden")
    for fn, mixed in (("weights_dsgps.npz", False), ("weights_dsgps_mixed.npz", True)):
        w = np.load(os.path.join(gold, fn))
        sd = {n: torch.from_numpy(w[n]) for n in w.files if n != "k"}
        wf, wg = eng.pack_dsgps_train(sd, "cpu")
        base = int(nat.lib().psignn_param_grad_size(int(mixed), 1))
        assert wf.numel() == nat.lib().psignn_weights_size(int(mixed), 1)
        assert base + wg.numel() == nat.lib().psignn_dsgps_grad_size(int(mixed))
        back = eng.unpack_dsgps_grads(torch.cat([wf[:base], wg]), mixed)
        want = {n: t for n, t in sd.items() if not n.startswith(("autoencoder", "laynorm"))}
        assert set(back) == set(want)
        assert all(torch.equal(back[n], want[n]) for n in want)
    w = np.load(os.path.join(gold, "weights_dss.npz"))
    sd = {n: torch.from_numpy(w[n]) for n in w.files if n not in ("k", "alpha")}
    for t in (0, 7, int(w["k"]) - 1):
        wf = eng.pack_dss_train(sd, t, "cpu")
        assert wf.numel() == nat.lib().psignn_dss_grad_size()
        back = eng.unpack_dss_grads(wf, t)
        assert set(back) == {n for n in sd if n.startswith((f"phi_to_list.{t}.", f"phi_from_list.{t}.", f"psi_list.{t}."))}
        assert all(torch.equal(back[n], sd[n]) for n in back)


def test_dss_schema_collates():
    """Graphs in the DSS schema (reader.py:61-92) keep their fields through data.collate, and a union batch converted with
    to_dss_batch keeps its graph ids (the plan tiles graph by graph)."""
    data, dss = pkg("data"), pkg("dss")
    meshes = [data.make_hex_problem(n, seed=n) for n in (5, 7)]
    a = data.collate([dss.to_dss_batch(m) for m in meshes])
    b = dss.to_dss_batch(data.collate(meshes))
    for k in ("a_ij_norm", "b_prime", "b_prime_norm", "edge_index", "batch"):
        assert torch.equal(getattr(a, k), getattr(b, k)), k
    assert a.num_graphs == 2 and int(b.batch.max()) == 1


def test_checkpoints_written_here_load_with_weights_only(tmp_path):
    """TrainModel pickles config_model["solver"] by this package's module path, the reference by ``utilities.solver.broyden``:
    ``load_reference_checkpoint`` (weights_only=True) accepts both."""
    top, solver = pkg(""), pkg("utilities.solver")
    cfg = dict(latent_dim=10, n_layers=1, solver=solver.broyden, fw_tol=1e-5, fw_thres=10, bw_tol=1e-6, bw_thres=10, path_logs=None)
    net = pkg("model_psignn").ModelDEQDSS(cfg)
    tr = pkg("training_class").TrainModel(dict(loader_train=[], loader_val=[], model=net, config_model=cfg, lr_deq=1e-3, lr_ae=1e-3,
                                              sched_step_deq=0.5, sched_step_ae=0.5, path_ckpt=str(tmp_path), min_loss_save=1e9,
                                              max_epochs=0, gradient_clip=0.1, sup_weight=0.0, jac_weight=1.0))
    tr.save_model(tr.checkpoint(0), dirName=str(tmp_path), model_name="m")
    ck = top.load_reference_checkpoint(str(tmp_path / "m.pt"))
    assert ck["hyperparameters"]["solver"] is solver.broyden
    assert set(ck["state_dict"]) == set(net.state_dict())
