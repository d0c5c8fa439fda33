"""Convergence / cost probe: Newton-Krylov (analytic JVP) vs on-device Broyden on a large synthetic mesh."""
import importlib, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
nodes = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
eps = float(sys.argv[2]) if len(sys.argv) > 2 else 1e-5
pkg = importlib.import_module("psi-gnn_amd"); eng = importlib.import_module("psi-gnn_amd.engine")
solver = importlib.import_module("psi-gnn_amd.utilities.solver")
dev = torch.device("cuda:0")
mesh = pkg.data.make_hex_problem(pkg.data.hex_n_for_nodes(nodes), seed=0, compute_sol=False).to(dev)
w = np.load(os.path.join(ROOT, "tests", "golden", "weights_dirichlet.npz")); sd = {k: torch.from_numpy(w[k]) for k in w.files}
P = "autoencoder.encoder.mlp.mlp."
h0 = eng.mlp2(mesh.x, sd[P + "0.weight"].to(dev), sd[P + "0.bias"].to(dev), sd[P + "2.weight"].to(dev), sd[P + "2.bias"].to(dev))
fm = eng.FixedPointMap(eng.MeshPlan(mesh), eng.PackedWeights(sd, dev), h0, mesh.prb_data)
for m in (30, 60):
    torch.cuda.synchronize(); t0 = time.time()
    out = solver.newton_krylov(fm, h0, threshold=60, eps=eps, inner_m=m, inner_tol=1e-2)
    torch.cuda.synchronize(); t1 = time.time()
    print(f"NK m={m}: N={fm.plan.N} outer={out['n_outer']} fevals={out['n_feval']} lowest={out['lowest']:.2e} time={t1 - t0:.3f}s rel_trace={['%.1e' % r for r in out['rel_trace'][:12]]}")
thr = 1500 if nodes <= 200000 else 400
torch.cuda.synchronize(); t0 = time.time()
ob = solver.broyden(fm, h0, threshold=thr, eps=eps, keep_trace=False)
torch.cuda.synchronize(); t1 = time.time()
print(f"Broyden: n_iter={ob['n_iter']} lowest={ob['lowest']:.2e} time={t1 - t0:.3f}s")
if out["lowest"] < 10 * eps and ob["lowest"] < 10 * eps:
    print("rel diff NK vs Broyden:", float((out["result"] - ob["result"]).norm() / ob["result"].norm()))
