#!/bin/bash
# Runs ON THE GPU BOX: matrix-core utilisation of the kernels that issue MFMA instructions (SQ counters, one --pmc pass)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r3_mfma
rm -rf $OUT && mkdir -p $OUT
export PSIGNN_STAGE1=mfma PSIGNN_JVP_STAGE1=mfma
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $OUT/p1 -- python3 scripts/prof_mfma.py 1000000 5 > $OUT/p1.log 2>&1 || echo "pass failed"
tail -1 $OUT/p1.log
python3 - <<'PY'
import csv, glob, collections, json
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
dur = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob("gpurun_out/r3_mfma/p1/*/*counter_collection.csv"):
    for row in csv.DictReader(open(f)):
        kn = row["Kernel_Name"].split("(")[0].replace("void ", "")
        a = acc[kn][row["Counter_Name"]]; a[0] += float(row["Counter_Value"]); a[1] += 1
for f in glob.glob("gpurun_out/r3_mfma/p1/*/*kernel_trace.csv"):
    for row in csv.DictReader(open(f)):
        kn = row["Kernel_Name"].split("(")[0].replace("void ", "")
        d = dur[kn]; d[0] += float(row["End_Timestamp"]) - float(row["Start_Timestamp"]); d[1] += 1
out = {}
for kn, cs in acc.items():
    m = {c: v / max(n, 1) for c, (v, n) in cs.items()}
    if m.get("SQ_INSTS_VALU_MFMA_MOPS_F32", 0) <= 0: continue
    us = dur[kn][0] / max(dur[kn][1], 1) * 1e-3
    # SQ_VALU_MFMA_BUSY_CYCLES: cycles a SIMD's matrix pipe was busy, summed over the chip's 1024 SIMDs; GRBM_GUI_ACTIVE: cycles summed over the 8 XCDs
    cyc = m.get("GRBM_GUI_ACTIVE", 0) / 8.0
    m["avg_us_under_pmc"] = us
    m["mfma_busy_frac"] = m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (1024.0 * cyc) if cyc else None
    out[kn] = m
    print(kn[:70], {k: (round(v, 4) if isinstance(v, float) else v) for k, v in m.items()})
json.dump(out, open("gpurun_out/r3_mfma/mfma_utilisation.json", "w"), indent=1)
PY
