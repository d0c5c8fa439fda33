#!/bin/bash
# Runs ON THE GPU BOX: sweep 2 with one row in flight per wave vs two (solver.hip SWEEP_V_AHEAD), interleaved
cd "$GRAFT_REPO_ROOT"
line() { python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
ks={x['kernel']:x for x in d['kernels']}
g=lambda n:((round(ks[n]['avg_us'],1), round(ks[n].get('frac_of_8TBps',0),3)) if n in ks else None)
print('it/s', round(d['iters_per_sec'],1), 'v', g('k_sweep_v'), 'u2d', g('k_sweep_u2d'), 'u1', g('k_sweep_u1'))"; }
for v in ${AB_V:-0 1 0 1}; do
  (cd psi-gnn_amd/csrc && rm -f solver.o && make EXTRA="-DSWEEP_V_AHEAD=$v" > /dev/null 2>&1) || { echo "build failed: $v"; continue; }
  for K in 20 50 100; do
    echo "AHEAD=$v K=$K: $(timeout -k 10 200 python3 bench.py --steps $K --warmup 5 --repeats 3 --no-cpu-baseline 2>/dev/null | line)"
  done
done
(cd psi-gnn_amd/csrc && rm -f solver.o && make > /dev/null 2>&1)
