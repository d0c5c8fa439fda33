import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import torch, numpy as np
from conftest import load_case, load_weights, pkg, rel_l2, CASES
from oracle import psignn_oracle as orc
from test_gpu_parity import bind
dev = torch.device('cuda:0')
solver = pkg("utilities.solver")
for name in CASES:
    g, mesh, md, sd, fmap = bind(name, dev)
    print(name, "ref e7: h", rel_l2(g["broyden_e7_result"], g["fp64_result"]), "u", rel_l2(g["broyden_e7_u"], g["fp64_u"]), "ref lowest", float(g["broyden_e7_lowest"]))
    for eps in (1e-7, 5e-8, 3e-8, 2e-8):
        out = solver.broyden(fmap, fmap.h0, threshold=1500, eps=eps)
        print("   eps", eps, "n_iter", out["n_iter"], "lowest %.2e" % out["lowest"], "h", "%.2e" % rel_l2(out["result"], g["fp64_result"]), "u", "%.2e" % rel_l2(orc.decoder(sd, out["result"].cpu()), g["fp64_u"]), "stop", out["stop_reason"])
data, eng = pkg("data"), pkg("engine")
sd = load_weights("dirichlet")
mesh = data.make_hex_problem(60, seed=0, compute_sol=False); md = mesh.to(dev)
h0 = orc.encoder(sd, mesh.x)
fm = eng.FixedPointMap(eng.MeshPlan(md), eng.PackedWeights(sd, dev), h0.to(dev), md.prb_data)
got = fm(fm.h0)
v = torch.randn(mesh.num_nodes, 10, generator=torch.Generator().manual_seed(1)).to(dev)
jv = fm.jvp(got, v)
for eps in (1e-1, 1e-2, 1e-3, 1e-4):
    fd = (fm(got + eps * v) - fm(got - eps * v)) / (2 * eps)
    print("fd eps", eps, rel_l2(jv, fd))
