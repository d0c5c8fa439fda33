#!/bin/bash
# Runs ON THE GPU BOX: build-time variants of the folded sweep 3 (solver.hip k_sweep_u2d): mesh1m K = 20 and K = 50.
cd "$GRAFT_REPO_ROOT"
line() { python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
ks={x['kernel']:x for x in d['kernels']}
g=lambda n:((round(ks[n]['avg_us'],1), round(ks[n].get('frac_of_8TBps',0),3)) if n in ks else None)
print('it/s', round(d['iters_per_sec'],1), 'u2d', g('k_sweep_u2d'), 'v', g('k_sweep_v'), 'u1', g('k_sweep_u1'))"; }
for v in ${AB_VARIANTS:-"-DU2D_UNROLL=8"}; do
  v=${v//,/ }
  (cd psi-gnn_amd/csrc && rm -f solver.o && make EXTRA="$v" > /dev/null 2>&1) || { echo "build failed: $v"; continue; }
  a=$(timeout -k 10 200 python3 bench.py --steps 20 --warmup 5 --repeats 3 --no-cpu-baseline 2>/dev/null | line)
  b=$(timeout -k 10 200 python3 bench.py --steps 50 --warmup 5 --repeats 3 --no-cpu-baseline 2>/dev/null | line)
  echo "$v | K=20: $a | K=50: $b"
done
(cd psi-gnn_amd/csrc && rm -f solver.o && make > /dev/null 2>&1)
