#!/bin/bash
# Runs ON THE GPU BOX: MFMA counters of the parameter-gradient reduction (k_pgrad_outer) on the 1M-node mesh.
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_pgrad
rm -rf $OUT && mkdir -p $OUT
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_BUSY_CYCLES SQ_WAVE_CYCLES --output-format csv -d $OUT/a -- python3 scripts/jacreg_bench.py 577 3 > $OUT/a.log 2>&1 || \
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES --output-format csv -d $OUT/a -- python3 scripts/jacreg_bench.py 577 3 > $OUT/a.log 2>&1
python3 - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/pmc_pgrad/a/*/*counter_collection.csv")[0]
t = glob.glob("gpurun_out/pmc_pgrad/a/*/*kernel_trace.csv")[0]
acc = collections.defaultdict(lambda: [0.0, 0])
for row in csv.DictReader(open(f)):
    if "k_pgrad_outer" in row["Kernel_Name"]:
        a = acc[row["Counter_Name"]]; a[0] += float(row["Counter_Value"]); a[1] += 1
for k, (v, n) in acc.items(): print(k, v / max(n, 1), n)
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in csv.DictReader(open(t)) if "k_pgrad_outer" in r["Kernel_Name"]]
print("k_pgrad_outer avg us", sum(d) / len(d), len(d))
PY
