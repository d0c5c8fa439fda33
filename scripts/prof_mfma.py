"""Target for rocprofv3 --pmc passes on the kernels that use the matrix cores: stage 1 of the f / JVP tile kernels in their MFMA form
(PSIGNN_STAGE1=mfma, PSIGNN_JVP_STAGE1=mfma; off by default) and the parameter-gradient reduction k_pgrad_outer
(always MFMA: v_mfma_f32_16x16x4_f32 with K = nodes).

    PSIGNN_STAGE1=mfma PSIGNN_JVP_STAGE1=mfma python3 scripts/prof_mfma.py [nodes=1000000] [reps=5]
"""
import importlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
nodes = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
pkg = importlib.import_module("psi-gnn_amd")
eng = importlib.import_module("psi-gnn_amd.engine")
dev = torch.device("cuda:0")
mesh = pkg.data.make_hex_problem(pkg.data.hex_n_for_nodes(nodes), seed=0, compute_sol=False).to(dev)
w = np.load(os.path.join(ROOT, "tests", "golden", "weights_dirichlet.npz"))
sd = {k: torch.from_numpy(w[k]) for k in w.files}
P = "autoencoder.encoder.mlp.mlp."
h0 = eng.mlp2(mesh.x, sd[P + "0.weight"].to(dev), sd[P + "0.bias"].to(dev), sd[P + "2.weight"].to(dev), sd[P + "2.bias"].to(dev))
plan = eng.MeshPlan(mesh)
fm = eng.FixedPointMap(plan, eng.PackedWeights(sd, dev), h0, mesh.prb_data, None)
x = fm.to_plan(h0)
v = torch.randn_like(x)
for _ in range(reps):
    x = fm.fp(x)
    jv = fm.jvp_p(x, v)
    grad, oh = fm.param_vjp_p(x, v)
torch.cuda.synchronize()
print(f"N={plan.N} tiles={plan.n_tiles}: {reps} x (f, jvp, param_vjp) done; |grad| = {float(grad.norm()):.4e}")
