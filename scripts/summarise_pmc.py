"""Aggregate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes into per-kernel HBM traffic per launch.

    python3 scripts/summarise_pmc.py <fetch_dir> <write_dir> <out.json>

Corrections follow /opt/skills/guides/MI355X_MICROARCH.md §HBM: counters are in KiB; on gfx950 FETCH_SIZE
reports exactly 1/2 of the bytes of wide (16 B/lane) coalesced streaming reads -> doubled for the
streaming kernels (k_dots, k_axpy, k_resid, k_final, k_xnext: float4 loads only).  The tile kernel mixes
16-byte slot loads with 8-byte row loads; its read side is reported both raw and doubled (uncalibrated
between the two).  WRITE_SIZE is exact for 16-byte stores.
"""
import collections
import csv
import glob
import json
import sys


def load(d, counter):
    agg = collections.defaultdict(list)
    for f in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                full = r["Kernel_Name"].split("(")[0].replace("void ", "")
                name = full.split("<")[0]
                if name == "k_f_tile":  # template <P, MIXED, FUSED>: keep the fused / plain instantiations apart
                    targs = [a.strip() for a in full[full.index("<") + 1:full.rindex(">")].split(",")]
                    name = "k_f_tile_fused" if len(targs) > 2 and targs[2] == "true" else "k_f_tile"
                agg[name].append(float(r["Counter_Value"]))
    return agg


fetch, write = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
STREAM = {"k_dots", "k_axpy", "k_resid", "k_final", "k_xnext"}
out = {}
for k in sorted(set(fetch) | set(write)):
    f = fetch.get(k, [0.0])
    w = write.get(k, [0.0])
    rd_raw = 1024.0 * sum(f) / len(f)
    wr = 1024.0 * sum(w) / len(w)
    rd = 2.0 * rd_raw if k in STREAM else rd_raw
    out[k] = {"launches": len(f), "fetch_bytes_raw": rd_raw, "fetch_bytes_corrected": rd, "write_bytes": wr,
              "hbm_bytes_per_launch": rd + wr,
              "note": "fetch doubled (16 B/lane streaming reads, gfx950)" if k in STREAM else
                      "fetch raw; mixed access widths, true value between raw and 2x raw"}
for k in ("k_f_tile", "k_f_tile_fused"):
    if k in out:
        out[f"f({k})"] = out[k]
out["_bench_args"] = {"steps": 50, "warmup": 5, "workload": "mesh1m"}
json.dump(out, open(sys.argv[3], "w"), indent=1, sort_keys=True)
for k, v in out.items():
    if k.startswith("_"):
        continue
    print(f"{k:24s} launches {v['launches']:4d}  read {v['fetch_bytes_corrected'] / 1e6:9.1f} MB  write {v['write_bytes'] / 1e6:8.1f} MB")
