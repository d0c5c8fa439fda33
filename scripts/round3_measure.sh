#!/bin/bash
# Runs ON THE GPU BOX: the round's bench lines (after scripts/collect_profiles.sh r3, whose PMC model sits in profiles/), the
# phase stamps of k_f_tile (stamped build, restored afterwards) and the evaluation / adjoint timings.
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3_final; mkdir -p $O
timeout -k 10 300 python bench.py > $O/r3_bench_default.json 2> $O/r3_bench_default.err
timeout -k 10 200 python bench.py --steps 20 --warmup 5 > $O/r3_bench_k20.json 2> $O/r3_bench_k20.err
timeout -k 10 200 python bench.py --steps 100 --warmup 5 --no-cpu-baseline > $O/r3_bench_k100.json 2>/dev/null
for w in dir100k mixed100k batch50k mixedbatch50k; do timeout -k 10 200 python bench.py --workload $w --steps 100 --no-cpu-baseline > $O/r3_bench_$w.json 2> $O/r3_bench_$w.err; done
PSIGNN_BENCH_BACKEND=gloo timeout -k 10 300 python bench.py --gpus 2 --workload batch50k --nodes 20000 --steps 30 --no-cpu-baseline > $O/r3_bench_selflaunch_2ranks_one_gpu.json 2> $O/r3_bench_selflaunch.err
(cd psi-gnn_amd/csrc && rm -f fgnn_tile.o && make EXTRA="-DTILE_STAMPS=1" > /dev/null 2>&1) && timeout -k 10 120 python scripts/tile_phases.py > $O/r3_tile_phases.txt 2>&1
(cd psi-gnn_amd/csrc && rm -f fgnn_tile.o && make > /dev/null 2>&1)
timeout -k 10 200 python scripts/prof_f.py 1000000 50 0 dirichlet adjoint > $O/r3_prof_f_adjoint.txt 2>&1
timeout -k 10 200 python scripts/prof_f.py 1000000 50 0 mixed adjoint > $O/r3_prof_f_adjoint_mixed.txt 2>&1
python - <<PY
import json
for f in ("r3_bench_default","r3_bench_k20","r3_bench_k100","r3_bench_dir100k","r3_bench_mixed100k","r3_bench_batch50k","r3_bench_mixedbatch50k","r3_bench_selflaunch_2ranks_one_gpu"):
    try:
        d=json.loads(open("$O/%s.json"%f).read().strip().splitlines()[-1])
        r=d.get("roofline",{}); rf=d.get("roofline_f",{})
        print(f, "it/s %.1f"%d["iters_per_sec"], "value %.3e"%d["value"], "ms/step %.4f"%d["ms_per_step"], d.get("ms_per_step_spread"), "first", d.get("ms_per_step_first"), "| dom", r.get("kernel"), round(r.get("frac",0),3), "traffic", r.get("traffic"), "| f_fused", round(rf.get("avg_launch_us",0),1), round(rf.get("frac",0),3), "traffic", rf.get("traffic"), "| f_only", d.get("f_only",{}).get("us_per_eval"), "jvp", d.get("roofline_jvp",{}).get("avg_launch_us"), d.get("roofline_jvp",{}).get("frac"), "nk", d.get("newton_krylov",{}).get("frac"), "| iter frac", round(d["roofline_iter"]["frac"],3), "cpu", d.get("cpu_baseline",{}).get("value"), "ranks", d.get("ranks"), d.get("n_gpus"))
    except Exception as e: print(f, "ERR", e)
PY
tail -14 $O/r3_tile_phases.txt; tail -8 $O/r3_prof_f_adjoint.txt; tail -6 $O/r3_prof_f_adjoint_mixed.txt
