"""The reference's test-set protocol (dirichlet/psignn/test/test_func.py:68-120: 2 000 graphs of ~500 nodes, DataLoader
batch_size 50 = union graphs, per-graph metrics; recorded: 40 batches in 39 s on a Quadro T2000) on synthetic
in-distribution meshes (warped hexagons n=13, 547 nodes, random problems)."""
import importlib, json, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
pkg = importlib.import_module("psi-gnn_amd"); mod = importlib.import_module("psi-gnn_amd.model_psignn")
ev = importlib.import_module("psi-gnn_amd.evaluation")
n_graphs = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
dev = torch.device("cuda:0")
w = np.load(os.path.join(ROOT, "tests", "golden", "weights_dirichlet.npz")); sd = {k: torch.from_numpy(w[k]) for k in w.files}
net = mod.ModelPSIGNN(dict(latent_dim=10, n_layers=1, fw_tol=1e-5, fw_thres=500)); net.load_state_dict(sd); net = net.to(dev).eval()
t0 = time.time()
meshes = [pkg.data.make_hex_problem(13, seed=s, phase=0.1 * (s % 7)) for s in range(n_graphs)]
print(f"generated {n_graphs} meshes in {time.time() - t0:.1f} s")
ev.test_dataset(net, meshes[:50], dev, batch_size=50)   # warm-up (first-call initialisation)
rep = ev.test_dataset(net, meshes, dev, batch_size=50)
rep["nsteps_per_batch"] = [int(s) for s in rep["nsteps_per_batch"]]
print(json.dumps(rep))
json.dump(rep, open(os.path.join(ROOT, "gpurun_out", "eval_dataset.json"), "w"), indent=1)
