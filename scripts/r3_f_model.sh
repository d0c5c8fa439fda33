#!/bin/bash
# Runs ON THE GPU BOX: the measurements behind DESIGN section 4 "the bound of k_f_tile / k_jvp_tile" (round 3).
#  1. occupancy sweep of the plain f kernel (PSIGNN_TILE_LDS_MIN caps the workgroups per CU)
#  2. launch order: costliest tiles first in every XCD's run (PSIGNN_TILE_ORDER=cost), plain f and the fused step (bench K = 20)
#  3. SQ counters of k_f_tile, k_jvp_tile, k_vjp_tile_* (instruction counts, VALU busy, resident waves)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
echo "== 1. occupancy sweep, plain f, 1M nodes (50 evaluations each; workgroups per CU = floor(160 KB / LDS request))"
for mn in 0 30000 38000 50000 62000; do
  echo "LDS_MIN=$mn: $(PSIGNN_TILE_LDS_MIN=$mn timeout -k 10 200 python3 scripts/prof_f.py 1000000 50 0 dirichlet 2>&1 | tail -1)"
done
echo "== 2. tile launch order"
for rep in 1 2; do
for o in plan cost; do
  echo "order=$o plain f: $(PSIGNN_TILE_ORDER=$o timeout -k 10 200 python3 scripts/prof_f.py 1000000 50 0 dirichlet 2>&1 | tail -1)"
  echo "order=$o bench K=20: $(PSIGNN_TILE_ORDER=$o timeout -k 10 200 python3 bench.py --steps 20 --warmup 5 --repeats 3 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); ks={x['kernel']:x for x in d['kernels']}
print('it/s', round(d['iters_per_sec'],1), 'fused f us', round(ks['k_f_tile_fused']['avg_us'],1), 'frac', round(ks['k_f_tile_fused'].get('frac_of_8TBps',0),3))")"
done
done
echo "== 3. SQ counters (mean per launch)"
OUT=gpurun_out/r3_pmc_fjv
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
for d in sorted(glob.glob("gpurun_out/r3_pmc_fjv/p*/")):
    fs = glob.glob(d + "*/*counter_collection.csv")
    if not fs: continue
    acc = collections.defaultdict(lambda: [0.0, 0])
    for row in csv.DictReader(open(fs[0])):
        kn = row["Kernel_Name"]
        for key in ("k_f_tile", "k_jvp_tile", "k_vjp_tile_a", "k_vjp_tile_b"):
            if key in kn:
                a = acc[(key, row["Counter_Name"])]; a[0] += float(row["Counter_Value"]); a[1] += 1
    for (k, c), (v, n) in acc.items():
        out[k][c] = v / max(n, 1)
json.dump(out, open("gpurun_out/r3_pmc_fjv/counters.json", "w"), indent=1)
for k, cs in out.items():
    w = cs.get("SQ_WAVES", 0)
    print(k, {c: round(v) for c, v in cs.items()})
    if w: print("   per wave: VALU %.0f SALU %.0f SMEM %.0f LDS %.0f VMEM_RD %.0f" % tuple(cs.get(c, 0) / w for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_SMEM", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD")))
PY
