"""BASELINE configs[4] sizing check: 1M-node mesh, threshold 1500 (U + V = 120 GB; the reference's solver.py zero-fills the same
amount up front and OOMs at 21 205 nodes on its 4 GB card).  Prints time, memory and the residual trace summary."""
import importlib, json, os, sys, time, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
pkg = importlib.import_module("psi-gnn_amd"); eng = importlib.import_module("psi-gnn_amd.engine")
dev = torch.device("cuda:0")
thr = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
mesh = pkg.data.make_hex_problem(577, seed=0, compute_sol=False).to(dev)
w = np.load(os.path.join(ROOT, "tests", "golden", "weights_dirichlet.npz")); sd = {k: torch.from_numpy(w[k]) for k in w.files}
P = "autoencoder.encoder.mlp.mlp."
h0 = eng.mlp2(mesh.x, *[sd[P + k].to(dev) for k in ("0.weight", "0.bias", "2.weight", "2.bias")])
fm = eng.FixedPointMap(eng.MeshPlan(mesh), eng.PackedWeights(sd, dev), h0, mesh.prb_data)
sv = eng.DeviceBroyden(plan=fm.plan, threshold=thr, keep_trace=False)
torch.cuda.synchronize(); t0 = time.perf_counter()
out = sv.solve(fm, 1e-5, poll_every=50)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
r = np.array(out["rel_trace"][:out["n_iter"]])
print(json.dumps({"nodes": fm.plan.N, "threshold": thr, "solver_state_GB": (sv.nbytes() if callable(sv.nbytes) else sv.nbytes) / 1e9, "seconds": dt, "n_iter": out["n_iter"],
                  "iters_per_sec": out["n_iter"] / dt, "lowest": out["lowest"], "nstep": out["nstep"],
                  "rel_at": {str(i): float(r[i - 1]) for i in (1, 10, 100, 500, 1000, 1500) if i <= len(r)},
                  "torch_mem_GB": torch.cuda.memory_allocated() / 1e9}))
