#!/bin/bash
# Runs ON THE GPU BOX: SQ counters of the tile kernel on the 1M-node mesh (LDS vs VALU share).  Output: gpurun_out/pmc_f/
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_f
rm -rf $OUT && mkdir -p $OUT
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/a -- python3 scripts/prof_f.py 1000000 5 0 dirichlet > $OUT/a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/b -- python3 scripts/prof_f.py 1000000 5 0 dirichlet > $OUT/b.log 2>&1
python3 - <<'PY'
import csv, glob, collections
for tag in "ab":
    f = glob.glob(f"gpurun_out/pmc_f/{tag}/*/*counter_collection.csv")
    acc = collections.defaultdict(lambda: [0.0, 0])
    for row in csv.DictReader(open(f[0])):
        if "k_f_tile" in row["Kernel_Name"]:
            a = acc[row["Counter_Name"]]
            a[0] += float(row["Counter_Value"]); a[1] += 1
    for k, (v, n) in acc.items():
        print(tag, k, v / max(n, 1), n)
PY
