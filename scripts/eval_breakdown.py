"""Where the time of one test-set batch goes (50 graphs of 547 nodes, reference protocol)."""
import importlib, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
pkg = importlib.import_module("psi-gnn_amd"); mod = importlib.import_module("psi-gnn_amd.model_psignn")
ev = importlib.import_module("psi-gnn_amd.evaluation"); eng = importlib.import_module("psi-gnn_amd.engine")
dev = torch.device("cuda:0")
w = np.load(os.path.join(ROOT, "tests", "golden", "weights_dirichlet.npz")); sd = {k: torch.from_numpy(w[k]) for k in w.files}
net = mod.ModelPSIGNN(dict(latent_dim=10, n_layers=1, fw_tol=1e-5, fw_thres=500)); net.load_state_dict(sd); net = net.to(dev).eval()
meshes = [pkg.data.make_hex_problem(13, seed=s, phase=0.1 * (s % 7)) for s in range(200)]
def T():
    torch.cuda.synchronize(); return time.perf_counter()
for rep in range(3):
    acc = dict(collate=0, h2d=0, plan=0, forward=0, metrics=0)
    for b in range(4):
        chunk = meshes[b * 50:(b + 1) * 50]
        t0 = T(); batch = pkg.data.collate(chunk)
        t1 = T(); bd = batch.to(dev)
        t2 = T(); plan = eng.plan_for(bd)
        t3 = T(); u, loss = net(bd)
        t4 = T(); m = ev.errors_batch(u, bd)
        t5 = T()
        for k, v in zip(acc, (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4)): acc[k] += v / 4
    print({k: round(v * 1e3, 2) for k, v in acc.items()}, "ms per batch; nsteps", loss["nsteps"])
