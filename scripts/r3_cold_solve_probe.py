"""Wall time of the user-facing call utilities.solver.broyden(fmap, h0, threshold=K, eps=0) on the 1M-node mesh -- a fresh DeviceBroyden
(2 K vectors of state) per call -- against the same solve on a solver object that is kept (solver_obj=...), and what the allocation costs.

    python3 scripts/r3_cold_solve_probe.py [nodes=1000000] [K=20]
"""
import importlib, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
nodes = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
K = int(sys.argv[2]) if len(sys.argv) > 2 else 20
pkg = importlib.import_module("psi-gnn_amd"); eng = importlib.import_module("psi-gnn_amd.engine"); solver = importlib.import_module("psi-gnn_amd.utilities.solver")
dev = torch.device("cuda:0")
mesh = pkg.data.make_hex_problem(pkg.data.hex_n_for_nodes(nodes), seed=0, compute_sol=False).to(dev)
w = np.load(os.path.join(ROOT, "tests", "golden", "weights_dirichlet.npz")); sd = {k: torch.from_numpy(w[k]) for k in w.files}
P = "autoencoder.encoder.mlp.mlp."
h0 = eng.mlp2(mesh.x, sd[P + "0.weight"].to(dev), sd[P + "0.bias"].to(dev), sd[P + "2.weight"].to(dev), sd[P + "2.bias"].to(dev))
fm = eng.FixedPointMap(eng.MeshPlan(mesh), eng.PackedWeights(sd, dev), h0, mesh.prb_data, None)


def wall(fn, n=5):
    ts = []
    for _ in range(n):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    return ts


solver.broyden(fm, fm.h0, threshold=K, eps=0.0, keep_trace=False)
fresh = wall(lambda: solver.broyden(fm, fm.h0, threshold=K, eps=0.0, keep_trace=False))
sv = eng.DeviceBroyden(fm.plan, K, keep_trace=False)
kept = wall(lambda: solver.broyden(fm, fm.h0, threshold=K, eps=0.0, keep_trace=False, solver_obj=sv))
alloc = wall(lambda: eng.DeviceBroyden(fm.plan, K, keep_trace=False).close())
print(f"N={fm.plan.N} K={K}: fresh solver per call {['%.2f' % t for t in fresh]} ms; kept solver {['%.2f' % t for t in kept]} ms; "
      f"create + destroy alone {['%.2f' % t for t in alloc]} ms ({sv.nbytes / 1e9:.2f} GB of state)")
