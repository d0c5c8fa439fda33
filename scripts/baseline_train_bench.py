"""Training-step timing of the two baselines (back-propagation through their unrolled updates) on a union batch of synthetic
hexagon meshes, with the CPU oracle's restated step (plain autograd) timed beside it.

    python3 scripts/baseline_train_bench.py [graphs_per_batch=50] [hex_n=13] [steps=5] [cpu=1]

DS-GPS: k = 30, gamma 0.9 (dirichlet/dsgps/launch.sh); DSS: k = 30, alpha 1e-3, gamma 0.9 (dirichlet/dss/launch.sh).
Weights: the reference checkpoints (tests/golden/weights_dsgps.npz, weights_dss.npz).  Prints one JSON line.
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


def timed(fn, steps):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 50
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 13
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
    cpu = int(sys.argv[4]) if len(sys.argv) > 4 else 1
    data, dss = pkg("data"), pkg("dss")
    dev = torch.device("cuda:0")
    meshes = [data.make_hex_problem(n, seed=s, phase=0.37 * s) for s in range(B)]
    batch = data.collate(meshes)
    out = {"workload": f"union batch of {B} hexagon meshes (n={n}): {batch.num_nodes} nodes, k = 30 unrolled updates"}

    w = np.load(os.path.join(ROOT, "tests", "golden", "weights_dsgps.npz"))
    sd = {k: torch.from_numpy(w[k]) for k in w.files if k != "k"}
    net = pkg("dsgps").ModelDSGPS(dict(latent_dim=10, k=int(w["k"]), alpha=1e-3, gamma=0.9, path_logs=None))
    net.load_state_dict(sd)
    net = net.to(dev).train()
    bd = batch.to(dev)

    def step_dsgps():
        net.zero_grad()
        net(bd)[1]["train_loss"].backward()
    out["dsgps_gpu_s_per_step"] = timed(step_dsgps, steps)

    w2 = np.load(os.path.join(ROOT, "tests", "golden", "weights_dss.npz"))
    sd2 = {k: torch.from_numpy(w2[k]) for k in w2.files if k not in ("k", "alpha")}
    net2 = dss.DeepStatisticalSolver(dict(latent_dim=10, k=int(w2["k"]), alpha=float(w2["alpha"]), gamma=0.9, path_logs=None))
    net2.load_state_dict(sd2)
    net2 = net2.to(dev).train()
    b2 = dss.to_dss_batch(batch).to(dev)

    def step_dss():
        net2.zero_grad()
        net2(b2)[1]["train_loss"].backward()
    out["dss_gpu_s_per_step"] = timed(step_dss, steps)
    if cpu:
        from oracle import psignn_oracle as orc
        torch.set_num_threads(min(32, os.cpu_count() or 1))
        t0 = time.perf_counter()
        orc.dsgps_training_step(sd, batch, int(w["k"]), 0.9)
        out["dsgps_cpu_s_per_step"] = time.perf_counter() - t0
        t0 = time.perf_counter()
        orc.dss_training_step(sd2, orc.dss_batch(batch), int(w2["k"]), float(w2["alpha"]), 0.9)
        out["dss_cpu_s_per_step"] = time.perf_counter() - t0
        out["cpu_threads"] = torch.get_num_threads()
        out["cpu_kind"] = "port: oracle training steps (plain autograd through the restated updates)"
    print(json.dumps(out))


if __name__ == "__main__":
    main()
