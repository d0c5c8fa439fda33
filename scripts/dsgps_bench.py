"""DS-GPS step timing: k recurrent updates on one synthetic hexagon mesh, HIP tile kernel vs the CPU
oracle.  Algorithmic bytes per step = B_f (89 N + 20 E'): the same traffic as one PSI-GNN f evaluation.

    python3 scripts/dsgps_bench.py [nodes=1000000] [k=30] [reps=5] [cpu_steps=2]"""
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
    k = int(sys.argv[2]) if len(sys.argv) > 2 else 30
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
    cpu_steps = int(sys.argv[4]) if len(sys.argv) > 4 else 2
    data, eng, nat = pkg("data"), pkg("engine"), pkg("_native")
    w = np.load(os.path.join(ROOT, "tests", "golden", "weights_dsgps.npz"))
    sd = {n: torch.from_numpy(w[n]) for n in w.files if n != "k"}
    n = data.hex_n_for_nodes(nodes)
    mesh = data.make_hex_problem(n, seed=0, compute_sol=False)
    dev = torch.device("cuda:0")
    net = pkg("dsgps").ModelDSGPS(dict(latent_dim=10, k=k, alpha=1e-3, gamma=0.9, path_logs=None))
    net.load_state_dict(sd)
    net = net.to(dev).eval()
    md = mesh.to(dev)
    plan = eng.plan_for(md)
    h0 = net.autoencoder.encoder(md.x)
    wf = net.packed(dev)
    eng.dsgps_forward(plan, wf, h0, md.prb_data, 2)
    torch.cuda.synchronize()
    nat.prof_enable(True)
    t0 = time.perf_counter()
    for _ in range(reps):
        hk = eng.dsgps_forward(plan, wf, h0, md.prb_data, k)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    kern = nat.prof_collect()
    nat.prof_enable(False)
    calls, ms = kern["k_dsgps_tile"]
    us = ms / calls * 1e3
    bf = 89 * plan.N + 20 * plan.Ep
    out = {"workload": f"DS-GPS, {plan.N}-node hexagon mesh, E'={plan.Ep}, k={k} steps, trained checkpoint 30_ite_gamma_0_9",
           "gpu_s_per_inference": dt, "steps_per_s": k / dt, "edges_per_s": plan.Ep * k / dt,
           "k_dsgps_tile_avg_us": us, "alg_bytes_per_step": bf,
           "roofline": {"bound": "hbm", "achieved": bf / us / 1e3, "peak": 8000.0, "unit": "GB/s", "frac": bf / us / 1e3 / 8000.0}}
    if cpu_steps:
        from oracle import psignn_oracle as orc
        torch.set_num_threads(min(32, os.cpu_count() or 1))
        h0c = orc.encoder(sd, mesh.x)
        h = h0c
        t0 = time.perf_counter()
        for _ in range(cpu_steps):
            h = orc.dsgps_step(sd, h, h0c, mesh)
        ct = (time.perf_counter() - t0) / cpu_steps
        gp = eng.dsgps_forward(plan, wf, h0, md.prb_data, cpu_steps).cpu()
        out["cpu_baseline"] = {"value": 1.0 / ct, "unit": "steps/s", "cores": torch.get_num_threads(), "kind": "port",
                               "sample": f"{cpu_steps} oracle dsgps_step calls on the same mesh",
                               "rel_l2_gpu_vs_cpu": float((gp - h).norm() / h.norm())}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
