#!/bin/bash
# Runs ON THE GPU BOX: occupancy sweep and SQ counters of the linearised JVP (k_jvp_lin) and of its build pass, 1M nodes
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
echo "== occupancy sweep (workgroups per CU = floor(160 KB / LDS request); default request 25.9 KB -> 6)"
for mn in 0 30000 38000 50000 62000; do
  echo "LDS_MIN=$mn: $(PSIGNN_TILE_LDS_MIN=$mn timeout -k 10 200 python3 scripts/prof_f.py 1000000 50 0 dirichlet adjoint 2>&1 | grep -E 'lin jvp|jvp_p' | tr '\n' ' ')"
done
echo "== SQ counters (mean per launch)"
OUT=gpurun_out/r3_pmc_lin
rm -rf $OUT && mkdir -p $OUT
i=0
for grp in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD" \
           "SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES" \
           "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/p$i -- python3 scripts/prof_f.py 1000000 5 0 dirichlet adjoint > $OUT/p$i.log 2>&1 || echo "pass $i failed: $grp"
done
python3 - <<'PY'
import csv, glob, collections, json
out = collections.defaultdict(dict)
for d in sorted(glob.glob("gpurun_out/r3_pmc_lin/p*/")):
    fs = glob.glob(d + "*/*counter_collection.csv")
    if not fs: continue
    acc = collections.defaultdict(lambda: [0.0, 0])
    for row in csv.DictReader(open(fs[0])):
        kn = row["Kernel_Name"]
        for key in ("k_jvp_lin", "k_lin_build", "k_jvp_tile"):
            if key in kn:
                a = acc[(key, row["Counter_Name"])]; a[0] += float(row["Counter_Value"]); a[1] += 1
    for (k, c), (v, n) in acc.items():
        out[k][c] = v / max(n, 1)
json.dump(out, open("gpurun_out/r3_pmc_lin/counters.json", "w"), indent=1)
for k, cs in out.items():
    w = cs.get("SQ_WAVES", 0)
    print(k, {c: round(v) for c, v in cs.items()})
    if w: print("   per wave: VALU %.0f SALU %.0f SMEM %.0f LDS %.0f VMEM_RD %.0f" % tuple(cs.get(c, 0) / w for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_SMEM", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD")))
PY
