#!/bin/bash
# Runs ON THE GPU BOX: SQ / cache counters of the plain f tile kernel on the 1M-node mesh, one --pmc pass per group
# (prof_f.py: 3 warm-up + 5 timed evaluations = 8 launches).  Output: gpurun_out/pmc_f/counters.json (mean per launch).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_f
rm -rf $OUT && mkdir -p $OUT
i=0
for grp in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_ACTIVE_INST_LDS" \
           "SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCP_TCC_READ_REQ_sum" \
           "GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/p$i -- python3 scripts/prof_f.py 1000000 5 0 dirichlet > $OUT/p$i.log 2>&1 || echo "pass $i failed: $grp"
done
python3 - <<'PY'
import csv, glob, collections, json
out = {}
for d in sorted(glob.glob("gpurun_out/pmc_f/p*/")):
    fs = glob.glob(d + "*/*counter_collection.csv")
    if not fs: continue
    acc = collections.defaultdict(lambda: [0.0, 0])
    for row in csv.DictReader(open(fs[0])):
        if "k_f_tile" in row["Kernel_Name"]:
            a = acc[row["Counter_Name"]]
            a[0] += float(row["Counter_Value"]); a[1] += 1
    for k, (v, n) in acc.items():
        out[k] = {"mean_per_launch": v / max(n, 1), "launches": n}
json.dump(out, open("gpurun_out/pmc_f/counters.json", "w"), indent=1)
for k, v in out.items(): print(k, v["mean_per_launch"], v["launches"])
PY
