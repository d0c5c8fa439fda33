"""Generate the golden fixtures under tests/golden/.  Runs ONLY in the build container.

Needs /root/reference (read-only): it
  1. imports the reference's own ``dirichlet/psignn/utilities/solver.py`` (numpy/torch/scipy only),
  2. loads the two trained checkpoints with ``torch.load(weights_only=True)`` (the only non-tensor
     global in the pickle is ``utilities.solver.broyden`` → allow-listed),
  3. decodes the real gmsh mesh ``tests/special_geo/mesh_files/original/paraview_mesh.h5``
     (deflate streams, no h5py),
  4. runs the oracle restatement (oracle/psignn_oracle.py) and the reference solver side by side,
     asserts they agree bit-for-bit, and writes inputs + expected outputs as .npz (no pickles).

Nothing under /root/reference is copied: the fixtures are data (weights as plain arrays, mesh
coordinates/connectivity, input tensors, expected outputs).

    python oracle/make_golden.py
"""
from __future__ import annotations

import importlib
import importlib.util
import json
import os
import sys
import types
import zlib

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
OUT = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, ROOT)

from oracle import psignn_oracle as orc  # noqa: E402

data_mod = importlib.import_module("psi-gnn_amd.data")


def load_reference_solver():
    pkg = types.ModuleType("utilities")
    pkg.__path__ = []
    sys.modules["utilities"] = pkg
    spec = importlib.util.spec_from_file_location(
        "utilities.solver", os.path.join(REF, "dirichlet/psignn/utilities/solver.py"))
    mod = importlib.util.module_from_spec(spec)
    sys.modules["utilities.solver"] = mod
    spec.loader.exec_module(mod)
    pkg.solver = mod
    return mod


def load_checkpoint(path, ref_solver):
    torch.serialization.add_safe_globals([ref_solver.broyden])
    ck = torch.load(path, map_location="cpu", weights_only=True)
    sd = {k: v.detach().cpu().contiguous() for k, v in ck["state_dict"].items()}
    hp = {k: (v if isinstance(v, (int, float, str)) else str(v)) for k, v in ck["hyperparameters"].items()}
    return sd, hp, int(ck.get("epoch", -1))


def decode_original_mesh():
    b = open(os.path.join(REF, "tests/special_geo/mesh_files/original/paraview_mesh.h5"), "rb").read()
    streams, i = [], 0
    while i < len(b) - 1:
        if b[i] == 0x78 and b[i + 1] in (0x01, 0x5E, 0x9C, 0xDA):
            d = zlib.decompressobj()
            try:
                out = d.decompress(b[i:])
                if d.eof and len(out) >= 64:
                    streams.append(out)
                    i += len(b[i:]) - len(d.unused_data)
                    continue
            except zlib.error:
                pass
        i += 1
    sizes = [len(s) for s in streams]
    assert sizes[:6] == [11688, 7776, 7776, 7776, 7776, 3888], sizes
    pts = np.frombuffer(streams[0], "<f8").reshape(-1, 3)[:, :2].copy()
    topo = np.frombuffer(b"".join(streams[1:5]), "<i8")
    tris, segs, k = [], [], 0
    while k < len(topo):
        if topo[k] == 4:
            tris.append(topo[k + 1:k + 4]); k += 4
        elif topo[k] == 2:
            assert topo[k + 1] == 2
            segs.append(topo[k + 2:k + 4]); k += 4
        else:
            raise ValueError(topo[k])
    tri = np.array(tris, dtype=np.int64)
    seg = np.array(segs, dtype=np.int64)
    phys = np.frombuffer(streams[5], "<i4")
    assert (phys == 101).sum() == len(seg) == 70 and (phys == 606).sum() == len(tri) == 902
    assert pts.shape[0] == 487
    return pts, tri, seg


def batch_arrays(m, prefix="in_"):
    out = {}
    for k in ("x", "edge_index", "edge_attr", "a_ij", "y", "sol", "prb_data", "tags", "pos", "unit_normal_vector"):
        v = getattr(m, k, None)
        if v is not None:
            out[prefix + k] = v.numpy()
    return out


def same(a, b):
    return torch.equal(a, b)


def golden_case(name, sd, mesh, ref_solver, thr=500, big=False):
    """Inputs + expected outputs for one (weights, mesh) pair."""
    torch.manual_seed(0)
    g = batch_arrays(mesh)
    with torch.no_grad():
        h0 = orc.encoder(sd, mesh.x)
        f = lambda H: orc.function_forward(sd, H, h0, mesh)
        f1 = f(h0)
        f2 = f(f1)
        P = "deqdss.f."
        g.update(h0=h0.numpy(), f1=f1.numpy(), f2=f2.numpy())
        g["mp_to"] = orc.phi(sd, P + "phi_to_list.0", f1, mesh.edge_index, mesh.edge_attr, "source_to_target").numpy()
        g["mp_from"] = orc.phi(sd, P + "phi_from_list.0", f1, mesh.edge_index, mesh.edge_attr, "target_to_source").numpy()
        if orc.is_mixed_state_dict(sd):
            g["mp_neu"] = orc.phi(sd, P + "phi_neumann", f1, mesh.edge_index, mesh.edge_attr, "target_to_source").numpy()

        # reference solver vs restatement, bit for bit, on the same f
        for eps, tag in ((1e-5, "e5"), (1e-7, "e7")):
            r_ref = ref_solver.broyden(f, h0, threshold=thr, eps=eps)
            r_orc = orc.broyden(f, h0, threshold=thr, eps=eps)
            assert r_ref["nstep"] == r_orc["nstep"], (name, tag)
            assert r_ref["rel_trace"] == r_orc["rel_trace"] and r_ref["abs_trace"] == r_orc["abs_trace"], (name, tag)
            assert same(r_ref["result"], r_orc["result"]) and r_ref["lowest"] == r_orc["lowest"], (name, tag)
            assert len(r_ref["xest_trace"]) == len(r_orc["xest_trace"])
            assert all(same(a, b) for a, b in zip(r_ref["xest_trace"], r_orc["xest_trace"]))
            n_it = len(r_ref["xest_trace"]) - 1
            g[f"broyden_{tag}_result"] = r_ref["result"].numpy()
            g[f"broyden_{tag}_nstep"] = np.int64(r_ref["nstep"])
            g[f"broyden_{tag}_niter"] = np.int64(n_it)
            g[f"broyden_{tag}_lowest"] = np.float64(r_ref["lowest"])
            g[f"broyden_{tag}_rel_trace"] = np.array(r_ref["rel_trace"][:n_it], dtype=np.float64)
            g[f"broyden_{tag}_abs_trace"] = np.array(r_ref["abs_trace"][:n_it], dtype=np.float64)
            g[f"broyden_{tag}_u"] = orc.decoder(sd, r_ref["result"]).numpy()
            if tag == "e5" and not big:
                g["broyden_e5_x3"] = r_ref["xest_trace"][3].numpy()  # third iterate: early-divergence probe
            print(f"  {name}: broyden eps={eps:g}: nstep={r_ref['nstep']} iters={n_it} lowest={r_ref['lowest']:.3e}")

        if not big:
            r_ref = ref_solver.forward_iteration(f, h0, eps=1e-5, threshold=60)
            r_orc = orc.forward_iteration(f, h0, eps=1e-5, threshold=60)
            assert r_ref["nstep"] == r_orc["nstep"] and same(r_ref["result"], r_orc["result"])
            g["fwd_result"] = r_ref["result"].numpy()
            g["fwd_nstep"] = np.int64(r_ref["nstep"])
            g["fwd_rel_trace"] = np.array([float(t) for t in r_ref["rel_trace"]], dtype=np.float64)

            r_ref = ref_solver.anderson(f, h0, threshold=80, eps=1e-5)
            r_orc = orc.anderson(f, h0, threshold=80, eps=1e-5)
            assert r_ref["nstep"] == r_orc["nstep"] and same(r_ref["result"], r_orc["result"])
            assert r_ref["rel_trace"] == r_orc["rel_trace"]
            g["anderson_result"] = r_ref["result"].numpy()
            g["anderson_nstep"] = np.int64(r_ref["nstep"])
            g["anderson_rel_trace"] = np.array(r_ref["rel_trace"], dtype=np.float64)

    # fp64 fixed point = the truth for the 1e-5 gate; the reference's broyden
    # allocates Us/VTs with the default dtype, so the default is switched for this call only.
    torch.set_default_dtype(torch.float64)
    try:
        sd64 = {k: v.double() for k, v in sd.items()}
        m64 = mesh.clone()
        for k in m64.keys():
            v = getattr(m64, k)
            if v.is_floating_point():
                setattr(m64, k, v.double())
        with torch.no_grad():
            h064 = orc.encoder(sd64, m64.x)
            f64 = lambda H: orc.function_forward(sd64, H, h064, m64)
            r64 = ref_solver.broyden(f64, h064, threshold=2 * thr, eps=1e-11)
            g["fp64_result"] = r64["result"].numpy()
            g["fp64_lowest"] = np.float64(r64["lowest"])
            g["fp64_u"] = orc.decoder(sd64, r64["result"]).numpy()
            g["fp64_f1"] = f64(h064).numpy()
            print(f"  {name}: fp64 fixed point: nstep={r64['nstep']} lowest={r64['lowest']:.3e}")
        if not big:
            # JVP / VJP probes at the first iterate (fp64 so they serve as truth for fp32 kernels)
            gen = torch.Generator().manual_seed(7)
            v = torch.randn(h064.shape, generator=gen, dtype=torch.float64)
            hp = f64(h064).detach()
            g["jv_point"] = hp.numpy()
            g["jv_dir"] = v.numpy()
            g["jvp64"] = orc.function_jvp(sd64, hp, h064, m64, v).detach().numpy()
            g["vjp64"] = orc.function_vjp(sd64, hp, h064, m64, v).detach().numpy()
    finally:
        torch.set_default_dtype(torch.float32)

    # metrics the reference's notebooks report (band check)
    with torch.no_grad():
        u = torch.from_numpy(g["broyden_e5_u"])
        r, c = mesh.edge_index
        Au = torch.zeros_like(u).index_add_(0, r, mesh.a_ij * u[c])
        res = Au - mesh.y
        g["metric_residual"] = np.float64((res ** 2).mean())
        g["metric_resnorm"] = np.float64(res.norm() / mesh.y.norm())
        g["metric_mse"] = np.float64(((u - mesh.sol) ** 2).mean())
        g["metric_rel"] = np.float64((u - mesh.sol).norm() / mesh.sol.norm())
        bd = (mesh.tags[:, 1] if mesh.tags.shape[1] == 3 else mesh.tags[:, 0]) == 1
        g["metric_mse_bound"] = np.float64(((u - mesh.sol)[bd] ** 2).mean())
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **g)
    return g


def main():
    os.makedirs(OUT, exist_ok=True)
    ref_solver = load_reference_solver()
    sd_d, hp_d, ep_d = load_checkpoint(os.path.join(REF, "dirichlet/psignn/results/constant_dataset/ckpt/best_model.pt"), ref_solver)
    sd_m, hp_m, ep_m = load_checkpoint(os.path.join(REF, "mixed/psignn/results/best_model/ckpt/best_model.pt"), ref_solver)
    print("dirichlet ckpt epoch", ep_d, "params", sum(v.numel() for v in sd_d.values()))
    print("mixed ckpt epoch", ep_m, "params", sum(v.numel() for v in sd_m.values()))
    np.savez(os.path.join(OUT, "weights_dirichlet.npz"), **{k: v.numpy() for k, v in sd_d.items()})
    np.savez(os.path.join(OUT, "weights_mixed.npz"), **{k: v.numpy() for k, v in sd_m.items()})
    json.dump({"dirichlet": {"epoch": ep_d, "hyperparameters": hp_d},
               "mixed": {"epoch": ep_m, "hyperparameters": hp_m}},
              open(os.path.join(OUT, "checkpoints.json"), "w"), indent=1, sort_keys=True)

    pts, tri, seg = decode_original_mesh()
    np.savez_compressed(os.path.join(OUT, "mesh_original.npz"), pos=pts, tri=tri, dirichlet_segments=seg)
    dmask = np.zeros(len(pts), dtype=bool)
    dmask[seg.ravel()] = True

    stats = []
    for seed in range(10):
        mesh = data_mod.make_from_triangulation(pts, tri, dmask, seed=seed, radius=1.0)
        assert mesh.edge_index.shape[1] == 2983  # tests/comparatives.ipynb cell 11
        if seed < 2:
            g = golden_case(f"original_dirichlet_s{seed}", sd_d, mesh, ref_solver)
        else:
            with torch.no_grad():
                u, loss, out = orc.model_forward(sd_d, mesh, solver=ref_solver.broyden, fw_tol=1e-5, fw_thres=500)
                r, c = mesh.edge_index
                res = torch.zeros_like(u).index_add_(0, r, mesh.a_ij * u[c]) - mesh.y
                bd = mesh.tags[:, 0] == 1
                g = {"metric_residual": float((res ** 2).mean()), "metric_resnorm": float(res.norm() / mesh.y.norm()),
                     "metric_mse": float(((u - mesh.sol) ** 2).mean()),
                     "metric_rel": float((u - mesh.sol).norm() / mesh.sol.norm()),
                     "metric_mse_bound": float(((u - mesh.sol)[bd] ** 2).mean()),
                     "broyden_e5_nstep": out["nstep"]}
        stats.append({k: float(g[k]) for k in ("metric_residual", "metric_resnorm", "metric_mse", "metric_rel",
                                                "metric_mse_bound", "broyden_e5_nstep")})
    mean = {k: float(np.mean([s[k] for s in stats])) for k in stats[0]}
    print("original mesh, 10 problems, mean:", mean)
    json.dump({"per_seed": stats, "mean": mean,
               "reference_band": {"metric_residual": 2.694e-3, "metric_resnorm": 1.752e-2, "metric_mse": 8.498e-3,
                                  "metric_rel": 1.448e-2, "metric_mse_bound": 1.978e-5, "nstep_single_sample": 66,
                                  "source": "dirichlet/psignn/test/test.ipynb cells 11,13"}},
              open(os.path.join(OUT, "original_band.json"), "w"), indent=1, sort_keys=True)

    golden_case("hex13_dirichlet_s0", sd_d, data_mod.make_hex_problem(13, seed=0), ref_solver)
    golden_case("hex13_mixed_s1", sd_m, data_mod.make_hex_problem(13, seed=1, mixed=True), ref_solver)
    golden_case("hex26_dirichlet_s0", sd_d, data_mod.make_hex_problem(26, seed=0), ref_solver, big=True)


if __name__ == "__main__":
    main()
