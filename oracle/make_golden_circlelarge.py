"""Second reference-held known answer for the oracle's f: the 5 329-node disc `circlelarge`.  Build container only.

The reference holds (i) the gmsh mesh tests/special_geo/mesh_files/circlelarge/paraview_mesh.h5 (meshio XDMF/HDF5, chunked +
deflate: /data0 5329 x 3 f64 points in chunks of (1333, 1), /data1 42 624 i64 mixed topology in chunks of 1 332, /data2,
/data3 cell tags), (ii) the problem: tests/special_geo/spec_geo.py:39-58 draws the coefficients of f and g with
np.random.seed(0) and divides coordinates by the radius passed in (3.0, tests/comparatives.ipynb cell 5), (iii) the recorded
result of ITS PSIGNN on that problem, tests/comparatives.ipynb cell 6 (fw_tol 1e-6, fw_thres 1000, spec_geo.py:292-293):
  Data(x=[5329, 1], edge_index=[2, 35874])   Residual 1.117e-03   MSE 2.173e+00   MSEDirichlet 2.507e-05   Nstep 231
This script decodes the mesh without h5py (zlib stream scan), rebuilds the problem with the build's own P1 assembler, runs
the oracle (restated f + solver, dirichlet checkpoint) and prints its row beside the recorded one; it writes the decoded
mesh + coefficients (data, no reference source) to tests/golden/mesh_circlelarge.npz and the two rows to
tests/golden/circlelarge_band.json.  The node numbering differs from the reference's (dolfin dof order): every recorded
quantity is invariant to it up to floating-point summation order.

    python oracle/make_golden_circlelarge.py
"""
import json
import os
import sys
import zlib

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_weights, pkg  # noqa: E402
from oracle import psignn_oracle as orc  # noqa: E402

RECORDED = {"nodes": 5329, "edges": 35874, "residual": 1.117e-3, "mse": 2.173, "mse_dirichlet": 2.507e-5, "nstep": 231,
            "source": "tests/comparatives.ipynb cells 5-6", "fw_tol": 1e-6, "fw_thres": 1000}
# the same mesh and problem in the second notebook (spec_geo_2.py: fw_tol 1e-5, fw_thres 1500; the run continued to 976 steps,
# i.e. down to its tolerance), with the two baselines beside it: tests/comparatives_2.ipynb cell 8
RECORDED_2 = {"psignn": {"residual": 3.266e-4, "mse": 1.432, "rel": 7.391e-2, "mse_dirichlet": 2.344e-5, "nstep": 976},
              "dss": {"residual": 2.848e-3, "mse": 1.369e2, "rel": 7.225e-1, "mse_dirichlet": 7.163e-4},
              "dsgps": {"residual": 7.186e-3, "mse": 3.604e1, "rel": 3.708e-1, "mse_dirichlet": 3.736e-5},
              "source": "tests/comparatives_2.ipynb cell 8"}


def zlib_streams(path):
    b = open(path, "rb").read()
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
    return streams


def decode_circlelarge():
    st = zlib_streams(os.path.join(REF, "tests/special_geo/mesh_files/circlelarge/paraview_mesh.h5"))
    sizes = [len(s) for s in st]
    assert sizes == [10664] * 12 + [10656] * 40, sizes
    n = 5329
    pts = np.zeros((4 * 1333, 3))
    for blk in range(4):          # chunk grid (4, 3) of (1333, 1) chunks, row-major
        for c in range(3):
            pts[blk * 1333:(blk + 1) * 1333, c] = np.frombuffer(st[blk * 3 + c], "<f8")
    pts = pts[:n]
    assert np.all(pts[:, 2] == 0.0)
    topo = np.frombuffer(b"".join(st[12:44]), "<i8")[:42624]
    phys = np.frombuffer(b"".join(st[44:48]), "<i4")[:10656]
    tris, segs, k = [], [], 0
    while k < len(topo):
        if topo[k] == 4:
            tris.append(topo[k + 1:k + 4]); k += 4
        elif topo[k] == 2:
            assert topo[k + 1] == 2
            segs.append(topo[k + 2:k + 4]); k += 4
        else:
            raise ValueError(int(topo[k]))
    tri, seg = np.array(tris, np.int64), np.array(segs, np.int64)
    assert len(tri) == 10419 and len(seg) == 237 and (phys == 101).sum() == 237, (len(tri), len(seg))
    assert tri.max() < n and np.unique(tri).size == n
    r = np.hypot(pts[:, 0], pts[:, 1])
    assert abs(r.max() - 3.0) < 1e-6                       # a disc of radius 3 (comparatives.ipynb cell 5)
    return pts[:, :2].copy(), tri, seg


def metrics(sd, mesh, out_dict):
    u = orc.decoder(sd, out_dict["result"])
    res = orc.residual_loss(u, mesh)
    d = mesh.tags[:, 0] == 1
    return {"residual": float(res), "mse": float(((u - mesh.sol) ** 2).mean()),
            "mse_dirichlet": float(((u[d] - mesh.x[d]) ** 2).mean()), "nstep": int(out_dict["nstep"]),
            "lowest": float(out_dict["lowest"])}


def main():
    torch.set_num_threads(8)
    pos, tri, seg = decode_circlelarge()
    dmask = np.zeros(len(pos), bool)
    dmask[seg.ravel()] = True
    rs = np.random.RandomState(0)                          # np.random.seed(0) of spec_geo.py:42
    pf, pg = rs.uniform(-10, 10, 3), rs.uniform(-10, 10, 6)
    data = pkg("data")
    mesh = data.hexmesh.make_from_triangulation(pos, tri, dmask, radius=3.0, coeffs=(pf, pg))
    print("graph:", mesh.num_nodes, "nodes,", mesh.edge_index.shape[1], "edges (recorded 5329 / 35874)")
    assert mesh.num_nodes == RECORDED["nodes"] and mesh.edge_index.shape[1] == RECORDED["edges"]
    sd = load_weights("dirichlet")
    with torch.no_grad():
        h0 = orc.encoder(sd, mesh.x)
        out = orc.broyden(lambda H: orc.function_forward(sd, H, h0, mesh), h0, threshold=1000, eps=1e-6)
    m = metrics(sd, mesh, out)
    u = orc.decoder(sd, out["result"])
    m["rel"] = float((u - mesh.sol).norm() / mesh.sol.norm())
    print("recorded (comparatives.ipynb, 231 steps)   :", {k: RECORDED[k] for k in ("residual", "mse", "mse_dirichlet", "nstep")})
    print("recorded (comparatives_2.ipynb, 976 steps) :", RECORDED_2["psignn"])
    print("oracle                                      :", m)
    for k in ("residual", "mse", "rel", "mse_dirichlet"):   # the converged run of the reference, to its printed digits
        assert abs(m[k] - RECORDED_2["psignn"][k]) <= 0.01 * RECORDED_2["psignn"][k], (k, m[k], RECORDED_2["psignn"][k])
    # the two baselines on the same problem (k = 30 updates; their checkpoints are tests/golden/weights_{dsgps,dss}.npz)
    base = {}
    wd = np.load(os.path.join(ROOT, "tests", "golden", "weights_dsgps.npz"))
    sdg = {k: torch.from_numpy(wd[k]) for k in wd.files if k != "k"}
    ug, _ = orc.dsgps_inference(sdg, mesh, int(wd["k"]))
    d = mesh.tags[:, 0] == 1
    base["dsgps"] = {"residual": float(orc.residual_loss(ug, mesh)), "mse": float(((ug - mesh.sol) ** 2).mean()),
                     "rel": float((ug - mesh.sol).norm() / mesh.sol.norm()), "mse_dirichlet": float(((ug[d] - mesh.x[d]) ** 2).mean())}
    ws = np.load(os.path.join(ROOT, "tests", "golden", "weights_dss.npz"))
    sds = {k: torch.from_numpy(ws[k]) for k in ws.files if k not in ("k", "alpha")}
    db = orc.dss_batch(mesh)
    us, _ = orc.dss_inference(sds, db, int(ws["k"]), float(ws["alpha"]))
    base["dss"] = {"residual": float(orc.dss_residual_loss(us, db.edge_index, db.a_ij, db.b_prime)),
                   "mse": float(((us - mesh.sol) ** 2).mean()), "rel": float((us - mesh.sol).norm() / mesh.sol.norm()),
                   "mse_dirichlet": float(((us[d] - mesh.x[d]) ** 2).mean())}
    for k in ("dsgps", "dss"):
        print(f"recorded {k:5s}:", RECORDED_2[k])
        print(f"oracle   {k:5s}:", base[k])
    # DSS is 30 unrolled updates, no solver: its row is a deterministic known answer -- all four printed digits
    for k in ("residual", "mse", "rel", "mse_dirichlet"):
        assert abs(base["dss"][k] - RECORDED_2["dss"][k]) <= 6e-4 * RECORDED_2["dss"][k], (k, base["dss"][k], RECORDED_2["dss"][k])
    # DS-GPS: the notebook's results_dsgps checkpoint is not in the reference tree (tests/test_multiple.py:51-53 loads it from
    # a directory that is not there); the k = 30 checkpoint under dirichlet/dsgps/results gives a different row -- not compared
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "mesh_circlelarge.npz"), pos=pos, tri=tri.astype(np.int32),
                        dirichlet=dmask, param_f=pf, param_g=pg, radius=np.float64(3.0))
    json.dump({"recorded": RECORDED, "recorded_2": RECORDED_2, "oracle": m, "oracle_baselines": base,
               "_doc": "oracle/make_golden_circlelarge.py: the reference's recorded PSIGNN row on its own 5 329-node disc and the "
                       "oracle's row on the same mesh and problem (fp32, dirichlet checkpoint, fw_tol 1e-6, fw_thres 1000)"},
              open(os.path.join(ROOT, "tests", "golden", "circlelarge_band.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
