#!/bin/bash
# Runs ON THE GPU BOX: build-time variants of the vector kernels (AB_VARIANTS), mesh1m K = 50 and batch50k K = 100.
cd "$GRAFT_REPO_ROOT"
line() { python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
ks={x['kernel']:x for x in d['kernels']}
g=lambda n:((round(ks[n]['avg_us'],1), round(ks[n].get('frac_of_8TBps',0),3)) if n in ks else None)
print('it/s', round(d['iters_per_sec'],1), 'dots', g('k_dots'), 'axpy', g('k_axpy'), 'final', g('k_final'), 'f', g('f(k_f_tile_fused)'))"; }
for v in ${AB_VARIANTS:-"-DWAVE_SUM_SHFL=1" "-DWAVE_SUM_SHFL=0"}; do
  (cd psi-gnn_amd/csrc && rm -f solver.o fpiter.o krylov.o && make EXTRA="$v" > /dev/null 2>&1) || { echo "build failed: $v"; continue; }
  a=$(timeout -k 10 200 python3 bench.py --steps 50 --warmup 5 --repeats 3 --no-cpu-baseline 2>gpurun_out/ab_wsum_err.log | line)
  b=$(timeout -k 10 200 python3 bench.py --workload batch50k --steps 100 --warmup 5 --repeats 3 --no-cpu-baseline 2>gpurun_out/ab_wsum_err.log | line)
  echo "$v | mesh1m: $a | batch50k: $b"
done
