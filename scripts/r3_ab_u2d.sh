#!/bin/bash
# Runs ON THE GPU BOX: register vs LDS form of the folded sweep 3 (solver.hip k_sweep_u2r<KB> / k_sweep_u2d), run-time knob,
# interleaved A/B on one box: mesh1m at the driver's K = 20 and at K = 50 / 100; batch50k at K = 100.
cd "$GRAFT_REPO_ROOT"
line() { python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
ks={x['kernel']:x for x in d['kernels']}
g=lambda n:((round(ks[n]['avg_us'],1), round(ks[n].get('frac_of_8TBps',0),3)) if n in ks else None)
print('it/s', round(d['iters_per_sec'],1), 'ms/step', d['ms_per_step_all'], 'u2d', g('k_sweep_u2d'), 'v', g('k_sweep_v'), 'u1', g('k_sweep_u1'), 'f', g('k_f_tile_fused'), 'red', g('k_reduce_check'), g('k_reduce_cb'))"; }
for rep in 1 2; do
for form in reg lds; do
  for K in 20 50; do
    echo "form=$form K=$K mesh1m: $(PSIGNN_U2D_FORM=$form timeout -k 10 200 python3 bench.py --steps $K --warmup 5 --repeats 3 --no-cpu-baseline 2>/dev/null | line)"
  done
done
done
for form in reg lds; do
  echo "form=$form K=100 mesh1m: $(PSIGNN_U2D_FORM=$form timeout -k 10 200 python3 bench.py --steps 100 --warmup 5 --repeats 3 --no-cpu-baseline 2>/dev/null | line)"
  echo "form=$form K=100 batch50k: $(PSIGNN_U2D_FORM=$form timeout -k 10 200 python3 bench.py --workload batch50k --steps 100 --warmup 5 --repeats 3 --no-cpu-baseline 2>/dev/null | line)"
  echo "form=$form K=100 dir100k: $(PSIGNN_U2D_FORM=$form timeout -k 10 200 python3 bench.py --workload dir100k --steps 100 --warmup 5 --repeats 3 --no-cpu-baseline 2>/dev/null | line)"
done
