"""Golden vectors for ``broyden(..., ls=True)`` (Armijo line search, utilities/solver.py:20-94).

Runs only where /root/reference exists: imports the reference's own solver module (numpy / torch / scipy only), runs it
with ``ls=True`` on the restated f of one fixture mesh and on a small analytic map, checks that the oracle's restatement
retraces it bit for bit, and stores the reference's outputs in tests/golden/broyden_ls.npz.

    python3 oracle/make_golden_ls.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))

from oracle import psignn_oracle as orc  # noqa: E402
from make_golden import load_reference_solver  # noqa: E402
from conftest import CASES, load_case, load_weights  # noqa: E402


def toy_map(n=16, d=10, seed=5):
    gen = torch.Generator().manual_seed(seed)
    A = torch.randn(n * d, n * d, generator=gen) * (0.9 / (n * d) ** 0.5)
    b = torch.randn(n * d, generator=gen)
    return (lambda x: torch.tanh(A @ x.reshape(-1) * 1.5 + b).reshape(n, d)), torch.zeros(n, d), A, b


def main():
    ref = load_reference_solver()
    out = {}
    same = lambda a, b: torch.equal(a, b)
    name = "hex13_dirichlet_s0"
    g, mesh = load_case(name)
    sd = load_weights(CASES[name])
    with torch.no_grad():
        h0 = orc.encoder(sd, mesh.x)
        f = lambda H: orc.function_forward(sd, H, h0, mesh)
        ft, x0t, A, b = toy_map()
        for tag, fn, x0, thr, eps in (("hex13", f, h0, 300, 1e-5), ("toy", ft, x0t, 60, 1e-6)):
            calls = [0]

            def counted(x, fn=fn):
                calls[0] += 1
                return fn(x)
            r_ref = ref.broyden(counted, x0, threshold=thr, eps=eps, ls=True)
            n_ref = calls[0]
            calls[0] = 0
            r_orc = orc.broyden(counted, x0, threshold=thr, eps=eps, ls=True)
            assert n_ref == calls[0], (tag, n_ref, calls[0])
            assert r_ref["nstep"] == r_orc["nstep"] and r_ref["rel_trace"] == r_orc["rel_trace"], tag
            assert same(r_ref["result"], r_orc["result"]) and r_ref["lowest"] == r_orc["lowest"], tag
            assert all(same(a, c) for a, c in zip(r_ref["xest_trace"], r_orc["xest_trace"]))
            r_plain = ref.broyden(fn, x0, threshold=thr, eps=eps)
            n_it = len(r_ref["xest_trace"]) - 1
            steps = [float(torch.norm(r_ref["xest_trace"][i + 1] - r_ref["xest_trace"][i])) for i in range(n_it)]
            print(f"{tag}: ls=True nstep {r_ref['nstep']} iters {n_it} f-calls {n_ref} lowest {r_ref['lowest']:.3e}; "
                  f"ls=False nstep {r_plain['nstep']} lowest {r_plain['lowest']:.3e}")
            out[f"{tag}_result"] = r_ref["result"].numpy()
            out[f"{tag}_nstep"] = np.int64(r_ref["nstep"])
            out[f"{tag}_niter"] = np.int64(n_it)
            out[f"{tag}_fcalls"] = np.int64(n_ref)
            out[f"{tag}_lowest"] = np.float64(r_ref["lowest"])
            out[f"{tag}_rel_trace"] = np.array(r_ref["rel_trace"][:n_it], dtype=np.float64)
            out[f"{tag}_step_norms"] = np.array(steps, dtype=np.float64)
            out[f"{tag}_x2"] = r_ref["xest_trace"][min(2, n_it)].numpy()
        out["toy_A"], out["toy_b"] = A.numpy(), b.numpy()
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "broyden_ls.npz"), **out)


if __name__ == "__main__":
    main()
