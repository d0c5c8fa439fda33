#!/bin/bash
# Runs ON THE GPU BOX: A/B of build-time variants of the solver's sweep kernels (solver.hip): mesh1m K=50 and batch50k K=100.
cd "$GRAFT_REPO_ROOT"
line() { python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
ks={x['kernel']:x for x in d['kernels']}
g=lambda n:((round(ks[n]['avg_us'],1), round(ks[n].get('frac_of_8TBps',0),3)) if n in ks else None)
print('it/s', round(d['iters_per_sec'],1), 'dots', g('k_dots'), 'axpy', g('k_axpy'), 'final', g('k_final'))"; }
for v in ${AB_VARIANTS:-"-DSWEEP_U16=1"}; do
  v=${v//,/ }
  (cd psi-gnn_amd/csrc && rm -f solver.o && make EXTRA="$v" > /dev/null 2>&1) || { echo "build failed: $v"; continue; }
  a=$(timeout -k 10 200 python3 bench.py --steps 50 --warmup 5 --repeats 3 --no-cpu-baseline 2>/dev/null | line)
  b=$(timeout -k 10 200 python3 bench.py --workload batch50k --steps 100 --warmup 5 --repeats 3 --no-cpu-baseline 2>/dev/null | line)
  c=$(PSIGNN_JGROUPS=4 PSIGNN_VEC_AX4=1 timeout -k 10 200 python3 bench.py --workload batch50k --steps 100 --warmup 5 --repeats 3 --no-cpu-baseline 2>/dev/null | line)
  echo "$v | mesh1m: $a | batch50k: $b | batch50k j4/ax4: $c"
done
(cd psi-gnn_amd/csrc && rm -f solver.o && make > /dev/null 2>&1)
