"""How often does the on-device Broyden solve reach 1e-7 on the reference's 5 329-node disc within 1 500 steps?  Runs the solve from the
encoder state and from 5 copies perturbed by one float32 ulp of noise; prints lowest residual / step per run (chaos vs regression)."""
import importlib, os, sys, json
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_circlelarge, load_weights, pkg
dev = torch.device("cuda:0")
mesh, band = load_circlelarge()
md = mesh.to(dev)
sd = load_weights("dirichlet")
net = pkg("model_psignn").ModelPSIGNN(dict(latent_dim=10, n_layers=1, fw_tol=1e-7, fw_thres=1500)).eval()
net.load_state_dict(sd); net = net.to(dev)
solver, eng = pkg("utilities.solver"), pkg("engine")
with torch.no_grad():
    h0 = net.autoencoder.encoder(md.x)
fm = net.deqdss.f.bind(h0, md)
gen = torch.Generator().manual_seed(0)
for i in range(6):
    x0 = fm.h0 if i == 0 else fm.h0 * (1 + 1.2e-7 * torch.randn(fm.h0.shape, generator=gen).to(dev))
    f2 = eng.FixedPointMap(fm.plan, fm.weights, x0, fm.prb)
    f2.h0 = x0
    # the map's Dirichlet rows come from h_initial: keep the true h0 there by binding the original map and only moving the start
    out = solver.broyden(lambda H: fm(H), x0, threshold=1500, eps=1e-7, keep_trace=False) if i else solver.broyden(fm, fm.h0, threshold=1500, eps=1e-7, keep_trace=False)
    tr = np.array(out["rel_trace"][:out["n_iter"]])
    print(i, "n_iter", out["n_iter"], "nstep", out["nstep"], "lowest %.3e" % out["lowest"], "stop", out["stop_reason"],
          "first step below 1e-5/1e-6:", int(np.argmax(tr < 1e-5)) if (tr < 1e-5).any() else None, int(np.argmax(tr < 1e-6)) if (tr < 1e-6).any() else None)
