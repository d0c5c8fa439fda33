"""Run the GNN block f (plan order) a few times on a synthetic mesh: target for rocprofv3 --pmc passes.

    python3 scripts/prof_f.py [nodes=1000000] [reps=20] [tile_target=0] [bc=dirichlet]
"""
import importlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
nodes = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
tt = int(sys.argv[3]) if len(sys.argv) > 3 else 0
bc = sys.argv[4] if len(sys.argv) > 4 else "dirichlet"
pkg = importlib.import_module("psi-gnn_amd")
eng = importlib.import_module("psi-gnn_amd.engine")
dev = torch.device("cuda:0")
n = pkg.data.hex_n_for_nodes(nodes)
mesh = pkg.data.make_hex_problem(n, seed=0, mixed=(bc == "mixed"), compute_sol=False).to(dev)
w = np.load(os.path.join(ROOT, "tests", "golden", f"weights_{bc}.npz"))
sd = {k: torch.from_numpy(w[k]) for k in w.files}
P = "autoencoder.encoder.mlp.mlp."
h0 = eng.mlp2(mesh.x, sd[P + "0.weight"].to(dev), sd[P + "0.bias"].to(dev), sd[P + "2.weight"].to(dev), sd[P + "2.bias"].to(dev))
plan = eng.MeshPlan(mesh, tile_target=tt)
fm = eng.FixedPointMap(plan, eng.PackedWeights(sd, dev), h0, mesh.prb_data, getattr(mesh, "unit_normal_vector", None))
x = fm.to_plan(h0)
for _ in range(3):
    x = fm.fp(x)
torch.cuda.synchronize()
t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
t0.record()
for _ in range(reps):
    x = fm.fp(x)
t1.record(); torch.cuda.synchronize()
print(f"N={plan.N} Ep={plan.Ep} tiled={plan.tiled} tiles={plan.n_tiles} max_rows={plan.max_tile_rows} ell_rows={plan.ell_rows} "
      f"f avg {t0.elapsed_time(t1) / reps * 1e3:.1f} us")
if len(sys.argv) > 5 and sys.argv[5] == "adjoint":
    H = fm.from_plan(x)
    V = torch.randn_like(H)
    Vp = fm.to_plan(V)
    lin = fm.linearize_p(x) if fm.can_linearize() else None
    extra = ((("lin build (plan order)", lambda: fm.linearize_p(x, lin)), ("lin jvp (plan order)", lambda: lin.jvp_p(Vp))) if lin else ())
    for name, fn in extra + (("jvp", lambda: fm.jvp(H, V)), ("vjp", lambda: fm.vjp(H, V)), ("f (caller order)", lambda: fm(H)),
                     ("vjp_p (plan order)", lambda: fm.vjp_p(x, Vp)), ("jvp_p (plan order)", lambda: fm.jvp_p(x, Vp))):
        fn(); torch.cuda.synchronize()
        t0.record()
        for _ in range(reps):
            fn()
        t1.record(); torch.cuda.synchronize()
        print(f"{name} avg {t0.elapsed_time(t1) / reps * 1e3:.1f} us")
