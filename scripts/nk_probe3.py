"""Native newton_krylov vs on-device Broyden on a large mesh (GPU).  python scripts/nk_probe3.py nodes=100000 warm=300 m=400 eta=0.05 outer=20"""
import importlib, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
kw = dict(a.split("=") for a in sys.argv[1:])
nodes = int(kw.get("nodes", 100000)); m = int(kw.get("m", 400)); eta = float(kw.get("eta", 0.05))
outer = int(kw.get("outer", 20)); warm = int(kw.get("warm", 300)); eps = float(kw.get("eps", 1e-5))
pkg = importlib.import_module("psi-gnn_amd"); eng = importlib.import_module("psi-gnn_amd.engine")
solver = importlib.import_module("psi-gnn_amd.utilities.solver")
dev = torch.device("cuda:0")
mesh = pkg.data.make_hex_problem(pkg.data.hex_n_for_nodes(nodes), seed=0, compute_sol=False).to(dev)
w = np.load(os.path.join(ROOT, "tests", "golden", "weights_dirichlet.npz")); sd = {k: torch.from_numpy(w[k]) for k in w.files}
P = "autoencoder.encoder.mlp.mlp."
h0 = eng.mlp2(mesh.x, sd[P + "0.weight"].to(dev), sd[P + "0.bias"].to(dev), sd[P + "2.weight"].to(dev), sd[P + "2.bias"].to(dev))
fm = eng.FixedPointMap(eng.MeshPlan(mesh), eng.PackedWeights(sd, dev), h0, mesh.prb_data)
torch.cuda.synchronize(); t0 = time.time()
ob = solver.broyden(fm, h0, threshold=int(kw.get("bthr", 1500)), eps=eps, keep_trace=False)
torch.cuda.synchronize(); t1 = time.time()
print(f"Broyden: N={fm.plan.N} n_iter={ob['n_iter']} lowest={ob['lowest']:.3e} at step {ob['nstep']} time={t1 - t0:.2f}s", flush=True)
torch.cuda.synchronize(); t0 = time.time()
out = solver.newton_krylov(fm, h0, threshold=outer, eps=eps, inner_m=m, inner_tol=eta, warm_start=warm)
torch.cuda.synchronize(); t1 = time.time()
print(f"NK (warm {warm}, m {m}, eta {eta}): lowest={out['lowest']:.3e} outer={out['n_outer']} n_feval={out['n_feval']} time={t1 - t0:.2f}s")
print("  rel trace", ["%.2e" % r for r in out["rel_trace"]])
print("  krylov", out["n_krylov"], "steps", out["step_lengths"])
