import importlib, os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
pkg = importlib.import_module("psi-gnn_amd"); eng = importlib.import_module("psi-gnn_amd.engine")
solver = importlib.import_module("psi-gnn_amd.utilities.solver")
dev = torch.device("cuda:0")
mesh = pkg.data.make_hex_problem(pkg.data.hex_n_for_nodes(100000), seed=0, compute_sol=False).to(dev)
w = np.load(os.path.join(ROOT, "tests", "golden", "weights_dirichlet.npz")); sd = {k: torch.from_numpy(w[k]) for k in w.files}
P = "autoencoder.encoder.mlp.mlp."
h0 = eng.mlp2(mesh.x, *[sd[P + k].to(dev) for k in ("0.weight", "0.bias", "2.weight", "2.bias")])
fm = eng.FixedPointMap(eng.MeshPlan(mesh), eng.PackedWeights(sd, dev), h0, mesh.prb_data)
out = solver.broyden(fm, h0, threshold=300, eps=1e-6, keep_trace=False)
r = out["rel_trace"]
print("lowest", out["lowest"], "nstep", out["nstep"], "n_iter", out["n_iter"], "rel[9,19,49,99,199]", [float(r[i]) for i in (9, 19, 49, 99, 199)])
x = out["result"]
print("check residual of result:", float((fm(x) - x).norm() / fm(x).norm()))
