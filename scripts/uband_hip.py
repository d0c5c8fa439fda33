"""The HIP path's own distribution of converged-solution errors, measured exactly as oracle/make_golden_uband.py measures the
reference CPU path's: 12 starts (encoder state, then +- one fp32 ulp of noise), eps = 1e-7, error of h* and of the decoded u
against the fp64 fixed point of the golden file.  Prints both distributions side by side (GPU; test infrastructure)."""
import json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import CASES, load_case, load_weights, pkg, rel_l2
from oracle import psignn_oracle as orc
dev = torch.device("cuda:0")
eng, solver = pkg("engine"), pkg("utilities.solver")
band = json.load(open(os.path.join(ROOT, "tests", "golden", "u_error_band.json")))
out = {}
for name in ("original_dirichlet_s0", "hex13_dirichlet_s0", "hex13_mixed_s1", "hex26_dirichlet_s0"):
    g, mesh = load_case(name)
    sd = load_weights(CASES[name])
    md = mesh.to(dev)
    h0 = torch.from_numpy(g["h0"])
    fm = eng.FixedPointMap(eng.plan_for(md), eng.PackedWeights(sd, dev), h0.to(dev), md.prb_data, getattr(md, "unit_normal_vector", None))
    eu, eh, ns = [], [], []
    for seed in range(12):
        gen = torch.Generator().manual_seed(1000 + seed)
        x0 = h0 if seed == 0 else h0 * (1 + 1e-7 * torch.randn(h0.shape, generator=gen))
        # a start different from h_initial: generic-callable form of the solver around the same HIP f (x0 != fm.h0)
        r = solver.broyden(fm, x0.to(dev), threshold=1000, eps=1e-7) if seed == 0 else \
            solver.broyden(lambda H: fm(H), x0.to(dev), threshold=1000, eps=1e-7)
        if r["lowest"] >= 1e-7:
            continue
        eu.append(rel_l2(orc.decoder(sd, r["result"].cpu()), g["fp64_u"]))
        eh.append(rel_l2(r["result"], g["fp64_result"]))
        ns.append(r["nstep"])
    b = band[name]
    print(f"{name}: HIP u mean {np.mean(eu):.3e} +- {np.std(eu, ddof=1):.1e} (max {max(eu):.3e}; first {eu[0]:.3e}) | reference u mean {b['u_mean']:.3e} +- {b['u_std']:.1e}"
          f" || HIP h mean {np.mean(eh):.3e} (max {max(eh):.3e}) | reference h mean {b['h_mean']:.3e}; nstep {min(ns)}..{max(ns)}; converged {len(eu)}/12", flush=True)
    out[name] = {"u": eu, "h": eh, "nstep": ns}
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "r2_uband_hip.json"), "w"), indent=1)
