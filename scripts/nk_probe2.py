"""Algorithm probe (GPU, torch vector ops around the HIP f / JVP kernels): which globalised Newton-Krylov variant gets
below Broyden's stagnation level on large meshes.  Not product code: the result decides what csrc/krylov.hip implements.

    python scripts/nk_probe2.py nodes=100000 m=300 eta=0.05 outer=40 warm=0
"""
import importlib, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
kw = dict(a.split("=") for a in sys.argv[1:])
nodes = int(kw.get("nodes", 100000)); mmax = int(kw.get("m", 300)); eta = float(kw.get("eta", 0.05))
n_outer = int(kw.get("outer", 40)); warm = int(kw.get("warm", 0)); eps = float(kw.get("eps", 1e-6))
pkg = importlib.import_module("psi-gnn_amd"); eng = importlib.import_module("psi-gnn_amd.engine")
solver = importlib.import_module("psi-gnn_amd.utilities.solver")
dev = torch.device("cuda:0")
mesh = pkg.data.make_hex_problem(pkg.data.hex_n_for_nodes(nodes), seed=0, compute_sol=False).to(dev)
w = np.load(os.path.join(ROOT, "tests", "golden", "weights_dirichlet.npz")); sd = {k: torch.from_numpy(w[k]) for k in w.files}
P = "autoencoder.encoder.mlp.mlp."
h0 = eng.mlp2(mesh.x, sd[P + "0.weight"].to(dev), sd[P + "0.bias"].to(dev), sd[P + "2.weight"].to(dev), sd[P + "2.bias"].to(dev))
fm = eng.FixedPointMap(eng.MeshPlan(mesh), eng.PackedWeights(sd, dev), h0, mesh.prb_data)
F, J = fm.fp, fm.jvp_p


def rel_of(x):
    fx = F(x)
    g = fx - x
    return float(g.norm() / (fx.norm() + 1e-9)), g


x = fm.to_plan(h0)
nfe = 0
if warm > 0:   # Broyden warm start
    ob = solver.broyden(fm, h0, threshold=warm, eps=1e-9, keep_trace=False)
    x = fm.to_plan(ob["result"])
    nfe += ob["n_iter"]
    print(f"warm start: Broyden {ob['n_iter']} iterations, lowest {ob['lowest']:.3e} at step {ob['nstep']}")
r, g = rel_of(x)
nfe += 1
print(f"N={fm.plan.N} start rel {r:.3e}")
torch.cuda.synchronize(); t0 = time.time()
V = torch.empty(mmax + 1, x.numel(), device=dev)
for outer in range(n_outer):
    b = -g.reshape(-1)
    beta = b.norm()
    V[0] = b / beta
    H = torch.zeros(mmax + 1, mmax, dtype=torch.float64, device=dev)
    k_used = mmax
    lin = 1.0
    for k in range(mmax):
        wv = (J(x, V[k].reshape(x.shape)) - V[k].reshape(x.shape)).reshape(-1)
        nfe += 1
        for _ in range(2):
            hh = V[:k + 1] @ wv
            wv = wv - hh @ V[:k + 1]
            H[:k + 1, k] += hh.double()
        hn = wv.norm()
        H[k + 1, k] = hn.double()
        V[k + 1] = wv / hn
        if (k + 1) % 10 == 0 or k + 1 == mmax:
            e1 = torch.zeros(k + 2, 1, dtype=torch.float64, device=dev); e1[0, 0] = beta.double()
            Hk = H[:k + 2, :k + 1].cpu(); e1c = e1.cpu()
            y = torch.linalg.lstsq(Hk, e1c).solution
            lin = float((Hk @ y - e1c).norm() / e1c.norm())
            if lin < eta:
                k_used = k + 1
                break
    dx = (y[:, 0].to(dev, torch.float32) @ V[:y.shape[0]]).reshape(x.shape)
    s, gn0 = 1.0, float(g.norm())
    while True:
        r1, g1 = rel_of(x + s * dx)
        nfe += 1
        if float(g1.norm()) <= (1 - 1e-4 * s) * gn0 or s < 1e-3:
            break
        s *= 0.5
    if float(g1.norm()) < gn0:
        x, g = x + s * dx, g1
    torch.cuda.synchronize()
    print(f"outer {outer}: krylov {k_used} lin {lin:.1e} step {s:.3f} -> rel {r1:.3e} nfe {nfe} t {time.time() - t0:.2f}s", flush=True)
    if r1 < eps:
        break
