"""Probe: how well is s = vT.dg reproduced by -dx.dg + sum_j a_j c_j (and beta = vT.g by -dx.g + sum_j a_j b_j) in fp32?
Plain torch Broyden on the GPU with the HIP f (diagnostic only)."""
import importlib, os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
pkg = importlib.import_module("psi-gnn_amd"); eng = importlib.import_module("psi-gnn_amd.engine")
dev = torch.device("cuda:0")
nodes = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
mesh = pkg.data.make_hex_problem(pkg.data.hex_n_for_nodes(nodes), seed=0, compute_sol=False).to(dev)
w = np.load(os.path.join(ROOT, "tests", "golden", "weights_dirichlet.npz")); sd = {k: torch.from_numpy(w[k]) for k in w.files}
P = "autoencoder.encoder.mlp.mlp."
h0 = eng.mlp2(mesh.x, *[sd[P + k].to(dev) for k in ("0.weight", "0.bias", "2.weight", "2.bias")])
fm = eng.FixedPointMap(eng.MeshPlan(mesh), eng.PackedWeights(sd, dev), h0, mesh.prb_data)
g = lambda x: (fm(x.reshape(h0.shape)) - x.reshape(h0.shape)).reshape(-1)
x = h0.reshape(-1).clone(); gx = g(x); upd = gx.clone()
U, V = [], []
worst_s = worst_b = worst_c = 0.0
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 200):
    xn = x + upd; gn = g(xn); dx = upd; dg = gn - gx; x, gx = xn, gn
    rel = float(gx.norm() / (gx + x).norm())
    if U:
        Um, Vm = torch.stack(U), torch.stack(V)
        a, c, b = Um @ dx, Vm @ dg, Vm @ gx
        vT = -dx + a @ Vm
        D1 = dx + dg - c @ Um
        D2 = gx - b @ Um
        s_f = float(-(dx.double() @ dg.double()) + (a.double() @ c.double()))
        b_f = float(-(dx.double() @ gx.double()) + (a.double() @ b.double()))
        canc = float((dx.abs().double() @ dg.abs().double()) + (a.abs().double() @ c.abs().double()))
    else:
        vT, D1, D2 = -dx, dx + dg, gx.clone()
        s_f, b_f = float(-(dx.double() @ dg.double())), float(-(dx.double() @ gx.double()))
        canc = abs(s_f)
    s_d, b_d = float(vT.double() @ dg.double()), float(vT.double() @ gx.double())
    worst_s = max(worst_s, abs(s_f - s_d) / abs(s_d)); worst_b = max(worst_b, abs(b_f - b_d) / max(abs(b_d), 1e-30))
    worst_c = max(worst_c, canc / abs(s_d))
    if it % 25 == 0 or it < 3:
        print(f"it {it:3d} rel {rel:.2e} s_direct {s_d:.4e} s_formula {s_f:.4e} relerr {abs(s_f - s_d) / abs(s_d):.1e} cancel {canc / abs(s_d):.1e} | beta relerr {abs(b_f - b_d) / max(abs(b_d), 1e-30):.1e}")
    u = D1 / s_d
    U.append(u); V.append(vT)
    upd = D2 - u * b_d
print("worst relerr s", worst_s, "beta", worst_b, "worst cancellation factor", worst_c)
