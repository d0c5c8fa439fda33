"""Training from random initialisation with the reference's launch configuration (dirichlet/psignn/launch_local.sh:
batch 50, lr_deq 0.01, lr_ae 0.05, gradient_clip 0.1, solver broyden, jac_weight 1.0, fw 1e-5 / 400, bw 1e-8 / 400,
seed 1234) through the reference's main.py shape (DataListLoader + DataParallel + TrainModel), on synthetic hexagon
meshes in the reader's schema (the reference's dataset needs FEniCS + gmsh to generate).

    python3 scripts/train_demo.py [graphs=300] [epochs=15] [out_dir=gpurun_out/train_demo] [model=psignn|dsgps|dss]

model = dsgps / dss: the baselines with their launch configurations (dirichlet/dsgps/launch.sh, dirichlet/dss/launch.sh:
k 30, alpha 1e-3, gamma 0.9, lr 0.01, gradient_clip 0.01, batch 100 -> 50 here) through TrainModelDSGPS / TrainModelDSS.

Prints one JSON line: per-epoch training / validation losses, seconds per epoch, Broyden step statistics.
"""
import importlib
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = lambda n="": importlib.import_module("psi-gnn_amd" + ("." + n if n else ""))


def main():
    G = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    epochs = int(sys.argv[2]) if len(sys.argv) > 2 else 15
    out_dir = sys.argv[3] if len(sys.argv) > 3 else os.path.join(ROOT, "gpurun_out", "train_demo")
    which = sys.argv[4] if len(sys.argv) > 4 else "psignn"
    sched = float(sys.argv[5]) if len(sys.argv) > 5 else 0.5   # 0.8 in the recorded run of the reference's checkpoint
    os.makedirs(out_dir, exist_ok=True)
    torch.manual_seed(1234)
    np.random.seed(1234)
    data, loader, solver = pkg("data"), pkg("loader"), pkg("utilities.solver")
    rng = np.random.default_rng(1234)
    graphs = [data.make_hex_problem(int(rng.integers(9, 15)), seed=s, phase=0.37 * s) for s in range(G)]
    n_tr, n_va = int(0.6 * G), int(0.2 * G)            # reader.py split: 60 / 20 / 20
    train, val = graphs[:n_tr], graphs[n_tr:n_tr + n_va]
    dev = torch.device("cuda:0")
    if which in ("dsgps", "dss"):
        return baseline(which, train, val, epochs, out_dir, dev, n_tr, n_va)
    cfg = {"latent_dim": 10, "hidden_dim": 10, "n_layers": 1, "fw_tol": 1e-5, "fw_thres": 400, "bw_tol": 1e-8,
           "bw_thres": 400, "solver": solver.broyden, "path_logs": out_dir}
    net = pkg("model_psignn").ModelDEQDSS(cfg)
    model = loader.DataParallel(net).to(dev)
    TrainModel = pkg("training_class").TrainModel
    tr = TrainModel({"model": model, "config_model": cfg,
                     "loader_train": loader.DataListLoader(train, batch_size=50, shuffle=True),
                     "loader_val": loader.DataListLoader(val, batch_size=50, shuffle=False),
                     "sup_weight": 0.0, "jac_weight": 1.0, "gradient_clip": 0.1, "lr_deq": 0.01, "sched_step_deq": sched,
                     "lr_ae": 0.05, "sched_step_ae": sched, "max_epochs": epochs, "min_loss_save": 1e5, "path_ckpt": out_dir})
    t0 = time.perf_counter()
    tr.train_model()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0

    def steps(name):
        path = os.path.join(out_dir, name)
        if not os.path.exists(path):
            return None
        rows = [l.split() for l in open(path).read().splitlines() if l.strip()]
        n = [int(r[1]) for r in rows if len(r) == 2]
        return {"solves": len(n), "mean_nstep": float(np.mean(n)), "max_nstep": int(np.max(n))} if n else None
    nodes = sum(int(g.x.shape[0]) for g in train)
    # held-out test split (the last 20 %), protocol of test_func.py:68-120: the model just trained (best checkpoint by
    # validation residual) next to the reference's checkpoint on the same graphs
    ev = pkg("evaluation")
    test = graphs[n_tr + n_va:]
    report = {}
    best = pkg().load_reference_checkpoint(os.path.join(out_dir, "best_model.pt"))["state_dict"]   # weights_only=True
    w = np.load(os.path.join(ROOT, "tests", "golden", "weights_dirichlet.npz"))
    for tag, sd in (("trained_here", best), ("reference_checkpoint", {k: torch.from_numpy(w[k]) for k in w.files})):
        m = pkg("model_psignn").ModelPSIGNN(dict(latent_dim=10, n_layers=1, fw_tol=1e-5, fw_thres=500))
        m.load_state_dict(sd)
        rep = ev.test_dataset(m.to(dev).eval(), test, dev, batch_size=50)
        report[tag] = {"mean": rep["mean"], "std": rep["std"], "graphs": rep["graphs"]}
    print(json.dumps({"test_set": report,
        "workload": f"{n_tr} training / {n_va} validation hexagon meshes (n = 9..14, {nodes} training nodes), batch 50, "
                    f"{epochs} epochs from xavier initialisation, reference launch_local.sh hyper-parameters (jac_weight 1.0)",
        "seconds_total": dt, "seconds_per_epoch": dt / epochs,
        "train": {k: [round(float(x), 6) for x in v] for k, v in tr.hist_train.items()},
        "val": {k: [round(float(x), 6) for x in v] for k, v in tr.hist_val.items()},
        "forward_solves": steps("forward_iteration.csv"), "backward_solves": steps("backward_iteration.csv")}))


def baseline(which, train, val, epochs, out_dir, dev, n_tr, n_va):
    loader, tc = pkg("loader"), pkg("training_class")
    cfg = {"latent_dim": 10, "k": 30, "alpha": 1e-3, "gamma": 0.9, "path_logs": out_dir}
    if which == "dss":
        dss = pkg("dss")
        train, val = [dss.to_dss_batch(g) for g in train], [dss.to_dss_batch(g) for g in val]
        net, Trainer = dss.DeepStatisticalSolver(cfg), tc.TrainModelDSS
    else:
        net, Trainer = pkg("dsgps").ModelDSGPS(cfg), tc.TrainModelDSGPS
    model = loader.DataParallel(net).to(dev)
    tr = Trainer({"model": model, "config_model": cfg, "loader_train": loader.DataListLoader(train, batch_size=50, shuffle=True),
                  "loader_val": loader.DataListLoader(val, batch_size=50, shuffle=False), "lr": 0.01, "gradient_clip": 0.01,
                  "max_epochs": epochs, "min_loss_save": 1e5, "path_ckpt": out_dir})
    t0 = time.perf_counter()
    tr.train_model()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(json.dumps({
        "workload": f"{which}: {n_tr} training / {n_va} validation hexagon meshes (n = 9..14), batch 50, {epochs} epochs from "
                    f"xavier initialisation, k = 30 unrolled updates, reference launch.sh hyper-parameters",
        "seconds_total": dt, "seconds_per_epoch": dt / epochs,
        "train": {k: [round(float(x), 6) for x in v] for k, v in tr.hist_train.items()},
        "val": {k: [round(float(x), 6) for x in v] for k, v in tr.hist_val.items()}}))


if __name__ == "__main__":
    main()
