"""Backward of the VJP (csrc/gather_backward.hip) on one large mesh: time per call, per-kernel breakdown.

    python3 scripts/jacreg_bench.py [hex_n=577] [reps=10]
"""
import importlib
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = lambda n="": importlib.import_module("psi-gnn_amd" + ("." + n if n else ""))


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 577
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    data, nat, eng = pkg("data"), pkg("_native"), pkg("engine")
    w = np.load(os.path.join(ROOT, "tests", "golden", "weights_dirichlet.npz"))
    sd = {k: torch.from_numpy(w[k]) for k in w.files}
    dev = torch.device("cuda:0")
    mesh = data.make_hex_problem(n, seed=0).to(dev)
    W = eng.PackedWeights(sd, dev)
    enc = pkg("model_psignn").ModelDEQDSS(dict(latent_dim=10, n_layers=1))
    enc.load_state_dict(sd)
    h0 = enc.to(dev).autoencoder.encoder(mesh.x).detach()
    fmap = eng.FixedPointMap(eng.plan_for(mesh), W, h0, mesh.prb_data, None)
    h = fmap(fmap(h0))
    v = torch.randn(h.shape, device=dev, generator=torch.Generator(device=dev).manual_seed(1))
    g = fmap.vjp(h, v)
    gbar = 2.0 * g / h.numel()
    fmap.vjp_backward(h, v, gbar)
    torch.cuda.synchronize()
    nat.prof_enable(True)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        grads, dh = fmap.vjp_backward(h, v, gbar)
    e1.record()
    torch.cuda.synchronize()
    kern = nat.prof_collect()
    nat.prof_enable(False)
    N = h.shape[0]
    print(json.dumps({"workload": f"backward of the VJP, hexagon n={n}: {N} nodes", "ms_per_call": e0.elapsed_time(e1) / reps,
                      "kernels_us": {k: round(1e3 * v[1] / v[0], 1) for k, v in sorted(kern.items(), key=lambda kv: -kv[1][1])},
                      "record_bytes": 2 * N * 320 * 4, "jac_loss": float(g.norm() ** 2 / h.numel()),
                      "grad_norm": float(torch.cat([t.reshape(-1) for t in grads.values()]).norm())}))


if __name__ == "__main__":
    main()
