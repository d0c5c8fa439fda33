"""DSS step timing: k = 30 updates (one weight set each) on one synthetic hexagon mesh, HIP tile kernel vs
the CPU oracle.  Algorithmic bytes per step: N*(40 + 40 + 12) + E'*(8 + 4) = 92 N + 12 E' (scalar edge feature).

    python3 scripts/dss_bench.py [nodes=1000000] [reps=5] [cpu_steps=2]"""
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
    nodes = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    cpu_steps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    data, nat, dss = pkg("data"), pkg("_native"), pkg("dss")
    w = np.load(os.path.join(ROOT, "tests", "golden", "weights_dss.npz"))
    sd = {n: torch.from_numpy(w[n]) for n in w.files if n not in ("k", "alpha")}
    k, alpha = int(w["k"]), float(w["alpha"])
    mesh = data.make_hex_problem(data.hex_n_for_nodes(nodes), seed=0, compute_sol=False)
    mesh.sol = torch.zeros_like(mesh.x)
    dev = torch.device("cuda:0")
    net = dss.DeepStatisticalSolver(dict(latent_dim=10, k=k, alpha=alpha, gamma=0.9, path_logs=None))
    net.load_state_dict(sd)
    net = net.to(dev).eval()
    b = dss.to_dss_batch(mesh).to(dev)
    net.latent(b, 2)
    torch.cuda.synchronize()
    nat.prof_enable(True)
    t0 = time.perf_counter()
    for _ in range(reps):
        net.latent(b)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    kern = nat.prof_collect()
    nat.prof_enable(False)
    calls, ms = kern["k_dss_tile"]
    us = ms / calls * 1e3
    N, Ep = mesh.num_nodes, int(b.edge_index.shape[1])
    bf = 92 * N + 12 * Ep
    out = {"workload": f"DSS, {N}-node hexagon mesh, E'={Ep}, k={k} updates, trained checkpoint dss_results",
           "gpu_s_per_inference": dt, "steps_per_s": k / dt, "edges_per_s": Ep * k / dt, "k_dss_tile_avg_us": us,
           "alg_bytes_per_step": bf,
           "roofline": {"bound": "hbm", "achieved": bf / us / 1e3, "peak": 8000.0, "unit": "GB/s", "frac": bf / us / 1e3 / 8000.0}}
    if cpu_steps:
        from oracle import psignn_oracle as orc
        torch.set_num_threads(min(32, os.cpu_count() or 1))
        ob = orc.dss_batch(mesh)
        h = torch.zeros(N, 10)
        t0 = time.perf_counter()
        for t in range(cpu_steps):
            h = orc.dss_step(sd, t, h, ob, alpha)
        ct = (time.perf_counter() - t0) / cpu_steps
        gp = net.latent(b, cpu_steps).cpu()
        out["cpu_baseline"] = {"value": 1.0 / ct, "unit": "steps/s", "cores": torch.get_num_threads(), "kind": "port",
                               "sample": f"{cpu_steps} oracle dss_step calls on the same mesh",
                               "rel_l2_gpu_vs_cpu": float((gp - h).norm() / h.norm())}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
