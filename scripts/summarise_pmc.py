"""Aggregate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes into per-kernel HBM traffic per launch.

    python3 scripts/summarise_pmc.py <fetch_dir> <write_dir> <out.json> <K> <nodes>

The passes run `bench.py --steps K --warmup 0 --repeats 1 --no-kernel-timing --no-cpu-baseline`: the solver kernels of the
timed solve are then the LAST launches of their kind, in iteration order (k_axpy: k = 0 .. K-1 stored pairs, k_dots:
k = 1 .. K-1).  Per kernel the summary is a linear model  bytes(k) = a + b k  (least squares over those launches; b = 0
for kernels whose traffic does not depend on k), which bench.py evaluates for whatever --steps it is run with.

Corrections follow /opt/skills/guides/MI355X_MICROARCH.md section HBM: counters are in KiB; on gfx950 FETCH_SIZE
reports exactly 1/2 of the bytes of wide (16 B/lane) coalesced streaming reads -> doubled for the
streaming kernels (k_dots, k_axpy, k_sweep_*, k_resid, k_final, k_xnext: float4 loads only).  The tile kernel mixes
16-byte slot loads with 8-byte row loads; its read side is reported raw (true value between raw and 2x raw).
WRITE_SIZE is exact for 16-byte stores.
"""
import collections
import csv
import glob
import json
import sys

import numpy as np


def load(d, counter):
    agg = collections.defaultdict(list)
    for f in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                full = r["Kernel_Name"].split("(")[0].replace("void ", "")
                name = full.split("<")[0]
                if name == "k_f_tile":  # template <P, MIXED, FUSED, MFMA1>: keep the fused / plain instantiations apart
                    targs = [a.strip() for a in full[full.index("<") + 1:full.rindex(">")].split(",")]
                    name = "k_f_tile_fused" if len(targs) > 2 and targs[2] == "true" else "k_f_tile"
                agg[name].append((int(r.get("Dispatch_Id", len(agg[name]))), float(r["Counter_Value"])))
    return {k: [v for _, v in sorted(rows)] for k, rows in agg.items()}


fetch, write = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
K, nodes = int(sys.argv[4]), int(sys.argv[5])
STREAM = {"k_dots", "k_axpy", "k_resid", "k_final", "k_xnext", "k_sweep_u1", "k_sweep_v", "k_sweep_u2", "k_sweep_u2d"}
SWEEPS = ("k_dots", "k_axpy", "k_sweep_u1", "k_sweep_v", "k_sweep_u2", "k_sweep_u2d")
import os
KMAX = int(os.environ.get("PSIGNN_U2D_KMAX", "24"))          # as bench.py: which sweep-3 form ran at each k / what sweep 1 read
KEEP = int(os.environ.get("PSIGNN_U2D_KEEP", "16"))
_fold = [KMAX > 0 and k + 1 < K and (k <= KMAX or KEEP > 0) for k in range(K)]
_keep0 = [(0 if k <= KMAX else k - KEEP) if _fold[k] else k + 1 for k in range(K)]
_u2d = [k for k in range(K) if _fold[k]]
_u1 = [c for c in ((_keep0[k - 1] if _fold[k - 1] else k) for k in range(1, K)) if c > 0]
KS = {"k_dots": list(range(1, K)), "k_axpy": list(range(0, K)), "k_final": list(range(0, K)), "k_f_tile_fused": list(range(0, K)),
      "k_reduce_check": list(range(0, K)), "k_sweep_u1": _u1, "k_sweep_v": list(range(0, K)),
      "k_sweep_u2": [k for k in range(0, K) if k not in _u2d], "k_sweep_u2d": _u2d}
out = {}
for k in sorted(set(fetch) | set(write)):
    f = np.array(fetch.get(k, [0.0])) * 1024.0
    w = np.array(write.get(k, [0.0])) * 1024.0
    n = min(len(f), len(w))
    rd = (2.0 if k in STREAM else 1.0) * f[-n:]
    tot = rd + w[-n:]
    ks = KS.get(k)
    if ks and n >= len(ks) and len(ks) >= 2:
        y = tot[-len(ks):]
        if k in SWEEPS and k not in ("k_sweep_u1", "k_dots") and ks[-1] == K - 1:
            ks, y = ks[:-1], y[:-1]    # iteration K's stop test fires in front of these sweeps: the last launch moves nothing
        b, a = np.polyfit(np.array(ks, dtype=float), y, 1) if k in SWEEPS else (0.0, float(np.mean(y)))
        fit_n = len(ks)
    else:
        a, b, fit_n = float(np.mean(tot)), 0.0, n
    out[k] = {"a": float(a), "b": float(b), "launches_fitted": int(fit_n), "mean_bytes_per_launch": float(np.mean(tot[-fit_n:])),
              "mean_read_raw": float(np.mean(f[-fit_n:])), "mean_write": float(np.mean(w[-fit_n:])),
              "note": "fetch doubled (16 B/lane streaming reads, gfx950)" if k in STREAM else
                      "fetch raw; mixed access widths, true value between raw and 2x raw"}
for k in ("k_f_tile", "k_f_tile_fused"):
    if k in out:
        out[f"f({k})"] = out[k]
json.dump({"_nodes": nodes, "_collected_with": {"steps": K, "warmup": 0, "repeats": 1, "workload": "mesh1m"}, "kernels": out},
          open(sys.argv[3], "w"), indent=1, sort_keys=True)
for k, v in out.items():
    print(f"{k:24s} fitted over {v['launches_fitted']:4d} launches: bytes(k) = {v['a'] / 1e6:9.1f} MB + {v['b'] / 1e6:7.2f} MB * k")
