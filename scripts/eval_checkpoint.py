"""Held-out test split of scripts/train_demo.py (the last 20 % of its graphs, protocol of test_func.py:68-120): a checkpoint
written by TrainModel next to the reference's checkpoint on the same graphs.

    python3 scripts/eval_checkpoint.py <best_model.pt> [graphs=1000]
"""
import importlib
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = lambda n="": importlib.import_module("psi-gnn_amd" + ("." + n if n else ""))


def main():
    path = sys.argv[1]
    G = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    data, ev = pkg("data"), pkg("evaluation")
    rng = np.random.default_rng(1234)
    graphs = [data.make_hex_problem(int(rng.integers(9, 15)), seed=s, phase=0.37 * s) for s in range(G)]
    test = graphs[int(0.6 * G) + int(0.2 * G):]
    dev = torch.device("cuda:0")
    ck = pkg().load_reference_checkpoint(path)
    w = np.load(os.path.join(ROOT, "tests", "golden", "weights_dirichlet.npz"))
    report = {"epoch_of_checkpoint": int(ck["epoch"]), "validation_residual_of_checkpoint": float(ck["hist_val"]["residual_loss"][ck["epoch"]])}
    for tag, sd in (("trained_here", ck["state_dict"]), ("reference_checkpoint", {k: torch.from_numpy(w[k]) for k in w.files})):
        m = pkg("model_psignn").ModelPSIGNN(dict(latent_dim=10, n_layers=1, fw_tol=1e-5, fw_thres=500))
        m.load_state_dict(sd)
        rep = ev.test_dataset(m.to(dev).eval(), test, dev, batch_size=50)
        report[tag] = {"mean": rep["mean"], "std": rep["std"], "graphs": rep["graphs"], "nsteps_per_batch": [int(x) for x in rep["nsteps_per_batch"]]}
    print(json.dumps(report))


if __name__ == "__main__":
    main()
