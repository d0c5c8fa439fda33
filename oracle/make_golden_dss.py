"""Golden vectors of the DSS baseline.  Runs only in the build container (reads /root/reference):
loads dirichlet/dss/results/dss_results/ckpt/best_model.pt with ``torch.load(weights_only=True)``, commits its state_dict
(tests/golden/weights_dss.npz, with k and alpha) and the oracle's ``DeepStatisticalSolver.inference`` restatement on the
fixture meshes in fp32 and fp64 (tests/golden/dss_<case>.npz).

Pin: the reference model needs torch_geometric (absent); the restatement is pinned by the recorded test statistics of
this checkpoint (dirichlet/dss/test/test_notebook.ipynb cell 11: Residual 2.255e-4, MSE 3.0e-2, MSEDirichlet 4.9e-4
over 2000 graphs) on the reference's own 487-node gmsh mesh -- printed below, asserted in tests/test_oracle_golden.py."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import psignn_oracle as orc  # noqa: E402
from conftest import load_case  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
CKPT = "/root/reference/dirichlet/dss/results/dss_results/ckpt/best_model.pt"


def main():
    ck = torch.load(CKPT, map_location="cpu", weights_only=True)
    sd = {k: v.float().contiguous() for k, v in ck["state_dict"].items()}
    k, alpha = int(ck["hyperparameters"]["k"]), float(ck["hyperparameters"]["alpha"])
    np.savez_compressed(os.path.join(OUT, "weights_dss.npz"), k=np.int64(k), alpha=np.float64(alpha),
                        **{n: v.numpy() for n, v in sd.items()})
    for case in ("original_dirichlet_s0", "original_dirichlet_s1", "hex13_dirichlet_s0"):
        _, mesh = load_case(case)
        b = orc.dss_batch(mesh)
        u, h, res = orc.dss_inference(sd, b, k, alpha, trace=True)
        sd64 = {n: v.double() for n, v in sd.items()}
        b64 = orc.dss_batch(type(mesh)(**{f: (getattr(mesh, f).double() if getattr(mesh, f).is_floating_point() else getattr(mesh, f))
                                         for f in mesh.keys()}))
        u64, h64 = orc.dss_inference(sd64, b64, k, alpha)
        idx = torch.where(mesh.tags.reshape(-1) == 1)[0]
        mse = float(torch.mean((u - b.sol) ** 2))
        msd = float(torch.mean((u[idx] - b.sol[idx]) ** 2))
        print(f"{case}: residual {res[-1]:.3e} mse {mse:.3e} mse_dirichlet {msd:.3e} fp32-vs-fp64 H {float((h - h64).norm() / h64.norm()):.2e}")
        np.savez(os.path.join(OUT, f"dss_{case}.npz"), h_k=h.numpy(), u_k=u.numpy(), h_k64=h64.numpy(), u_k64=u64.numpy(),
                 res_trace=np.array(res), mse=np.float64(mse), mse_dirichlet=np.float64(msd),
                 h_1=orc.dss_step(sd, 0, torch.zeros_like(h), b, alpha).numpy())


if __name__ == "__main__":
    main()
