#!/bin/bash
# Runs ON THE GPU BOX: the round's measurement set.  Profiles first (so that bench.py finds the PMC traffic model), then
# the bench lines, then the phase stamps (stamped build, restored afterwards).
cd "$GRAFT_REPO_ROOT"
bash scripts/collect_profiles.sh r2 > gpurun_out/r2_collect.log 2>&1
cp gpurun_out/prof_r2/pmc_traffic_model.json profiles/pmc_traffic_model.json
timeout -k 10 300 python bench.py > gpurun_out/r2_bench_final.json 2> gpurun_out/r2_bench_final.err
for w in dir100k mixed100k batch50k; do timeout -k 10 200 python bench.py --workload $w --steps 100 --no-cpu-baseline > gpurun_out/r2_bench_$w.json 2> gpurun_out/r2_bench_$w.err; done
PSIGNN_BENCH_STREAMS=1 timeout -k 10 200 python bench.py --workload batch50k --steps 100 --no-cpu-baseline > gpurun_out/r2_bench_batch50k_streams.json 2>/dev/null
timeout -k 10 100 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r2_bench_k20.json 2>/dev/null
(cd psi-gnn_amd/csrc && rm -f fgnn_tile.o && make EXTRA="-DTILE_STAMPS=1" > /dev/null 2>&1) && timeout -k 10 120 python scripts/tile_phases.py > gpurun_out/r2_tile_phases.txt 2>&1
(cd psi-gnn_amd/csrc && rm -f fgnn_tile.o && make > /dev/null 2>&1)
timeout -k 10 200 python scripts/prof_f.py 1000000 50 0 dirichlet adjoint > gpurun_out/r2_prof_f_adjoint.txt 2>&1
timeout -k 10 200 python scripts/prof_f.py 1000000 50 0 mixed adjoint > gpurun_out/r2_prof_f_adjoint_mixed.txt 2>&1
python - <<PY
import json
for f in ("r2_bench_final","r2_bench_k20","r2_bench_dir100k","r2_bench_mixed100k","r2_bench_batch50k","r2_bench_batch50k_streams"):
    try:
        d=json.loads(open("gpurun_out/%s.json"%f).read().strip().splitlines()[-1])
        r=d.get("roofline",{}); rf=d.get("roofline_f",{})
        print(f, "it/s %.1f"%d["iters_per_sec"], "value %.3e"%d["value"], "ms/step %.4f"%d["ms_per_step"], d.get("ms_per_step_spread"), "| dom", r.get("kernel"), round(r.get("frac",0),3), "traffic", r.get("traffic"), "| f_fused", round(rf.get("avg_launch_us",0),1), round(rf.get("frac",0),3), "traffic", rf.get("traffic"), "| f_only", d.get("f_only",{}).get("us_per_eval"), "jvp", d.get("roofline_jvp",{}).get("avg_launch_us"), "| iter frac", round(d["roofline_iter"]["frac"],3))
    except Exception as e: print(f, "ERR", e)
PY
tail -14 gpurun_out/r2_tile_phases.txt; cat gpurun_out/r2_prof_f_adjoint.txt | tail -6; tail -6 gpurun_out/r2_prof_f_adjoint_mixed.txt
