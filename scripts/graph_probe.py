"""HIP-graph probe for the launch-bound regime: the Broyden loop on small meshes with the launches of every 8-iteration chunk
captured into a HIP graph (PSIGNN_GRAPH=1, csrc/solver.hip) versus plain stream launches.  Repeated solves on ONE solver
object, so the graphs are instantiated once and replayed (the best case for graphs; the reference's protocols solve a new
mesh every time).

    PSIGNN_GRAPH=0 python3 scripts/graph_probe.py ; PSIGNN_GRAPH=1 python3 scripts/graph_probe.py
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
    data, eng, solver = pkg("data"), pkg("engine"), pkg("utilities.solver")
    dev = torch.device("cuda:0")
    w = np.load(os.path.join(ROOT, "tests", "golden", "weights_dirichlet.npz"))
    sd = {k: torch.from_numpy(w[k]) for k in w.files}
    W = eng.PackedWeights(sd, dev)
    P = "autoencoder.encoder.mlp.mlp."
    out = {"PSIGNN_GRAPH": os.environ.get("PSIGNN_GRAPH", "0"), "iterations_per_solve": 64, "us_per_iteration": {}}
    side = torch.cuda.Stream()   # stream capture is not allowed on the legacy default stream
    torch.cuda.set_stream(side)
    for n in (13, 26, 58, 100):
        mesh = data.make_hex_problem(n, seed=1, compute_sol=False).to(dev)
        h0 = eng.mlp2(mesh.x, *[sd[P + k].to(dev) for k in ("0.weight", "0.bias", "2.weight", "2.bias")])
        fmap = eng.FixedPointMap(eng.plan_for(mesh), W, h0, mesh.prb_data, None)
        sv = eng.DeviceBroyden(plan=fmap.plan, threshold=64, keep_trace=False)
        run = lambda: solver.broyden(fmap, h0, threshold=64, eps=0.0, keep_trace=False, solver_obj=sv)
        first = time.perf_counter()
        r = run()
        torch.cuda.synchronize()
        first = time.perf_counter() - first
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        reps = 20
        for _ in range(reps):
            r = run()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        out["us_per_iteration"][str(mesh.num_nodes)] = {"replayed": round(1e6 * dt / 64, 1), "first_solve": round(1e6 * first / 64, 1),
                                                        "n_iter": int(r["n_iter"]), "lowest": float(r["lowest"])}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
