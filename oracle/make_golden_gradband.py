"""Spread of the reference CPU path's OWN gradient error in a training step.  Build container only (test infrastructure).

One training step solves two fixed points (forward, adjoint) to a relative residual of 1e-7 and pushes the result through
f once more; with rho(J) ~ 0.99 the two stops leave an error that depends on the chaotic fp32 trajectory, so the fp32
gradient of a step is a SAMPLE around the float64 truth -- for the reference path (this oracle) as for the HIP path.  This
script measures the reference path's distribution: per fixture, the input x is perturbed by one fp32 ulp of noise (8 draws),
the oracle's restated training step (oracle/psignn_oracle.py::training_step = ModelDEQDSS.forward in train mode + the
backward hook, dirichlet/psignn/model.py:58-99,184-225) runs in fp32 at the test's tolerances and in float64 at 1e-12 on
the SAME perturbed input, and the worst per-tensor relative error is recorded.  Output: tests/golden/grad_error_band.json;
tests/golden/grad_truth_fp64.npz holds the float64 gradients of the first 4 draws; tests/test_gpu_training.py measures the
HIP path's error on the same 4 inputs and compares the two means.

    python oracle/make_golden_gradband.py
"""
import copy
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import CASES, load_case, load_weights  # noqa: E402
from oracle import psignn_oracle as orc  # noqa: E402

FIXTURES = ["hex13_dirichlet_s0", "original_dirichlet_s0", "hex13_mixed_s1"]
DRAWS = 8


def to64(sd, mesh):
    m64 = copy.copy(mesh)
    for k, v in list(vars(mesh).items()):
        if torch.is_tensor(v) and v.is_floating_point():
            setattr(m64, k, v.double())
    return {k: v.double() for k, v in sd.items()}, m64


def worst(got, want):
    scale = max(float(t.norm()) for t in want.values())
    errs = {k: float((got[k].double() - w.double()).norm()) / max(float(w.double().norm()), 1e-4 * scale) for k, w in want.items()}
    k = max(errs, key=errs.get)
    return errs[k], k


def main():
    torch.set_num_threads(8)
    out = {}
    truth = {}
    for name in FIXTURES:
        g, mesh = load_case(name)
        sd = load_weights(CASES[name])
        errs, names = [], []
        for draw in range(DRAWS):
            m = copy.copy(mesh)
            if draw > 0:
                gen = torch.Generator().manual_seed(2000 + draw)
                m.x = mesh.x * (1 + 1e-7 * torch.randn(mesh.x.shape, generator=gen))
            _, _, g32, _, _ = orc.training_step(sd, m, fw_tol=1e-7, fw_thres=600, bw_tol=1e-7, bw_thres=400)
            sd64, m64 = to64(sd, m)
            torch.set_default_dtype(torch.float64)
            try:
                _, _, g64, _, _ = orc.training_step(sd64, m64, fw_tol=1e-12, fw_thres=1500, bw_tol=1e-12, bw_thres=1500)
            finally:
                torch.set_default_dtype(torch.float32)
            if draw < 4:   # float64 truth of the first draws: the GPU test measures the HIP path on the same inputs
                for kk, t in g64.items():
                    truth[f"{name}/{draw}/{kk}"] = t.numpy()
            e, k = worst(g32, g64)
            errs.append(e)
            names.append(k)
            print(f"{name} draw {draw}: worst tensor error {e:.3e} ({k})", flush=True)
        e = np.array(errs)
        out[name] = {"errors": errs, "tensors": names, "mean": float(e.mean()), "std": float(e.std(ddof=1)),
                     "bound": float(e.mean() + 3 * e.std(ddof=1))}
        print(f"{name}: mean {e.mean():.3e} std {e.std(ddof=1):.2e} max {e.max():.3e} -> mean + 3 sigma {out[name]['bound']:.3e}", flush=True)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "grad_truth_fp64.npz"), **truth)
    json.dump({"_doc": "oracle/make_golden_gradband.py: worst per-tensor gradient error of the fp32 reference path (oracle "
                       "training_step, fw_tol = bw_tol = 1e-7) against the float64 step on the same input, 8 ulp-perturbed inputs",
               **out}, open(os.path.join(ROOT, "tests", "golden", "grad_error_band.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
