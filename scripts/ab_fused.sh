#!/bin/bash
# Runs ON THE GPU BOX: build-time variants of the f tile kernel judged by the fused Broyden step (bench mesh1m K = 50, batch50k K = 100).
cd "$GRAFT_REPO_ROOT"
line() { python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
ks={x['kernel']:x for x in d['kernels']}
g=lambda n:(round(ks[n]['avg_us'],1) if n in ks else None)
print('it/s', round(d['iters_per_sec'],1), 'fused', g('f(k_f_tile_fused)'), 'dots', g('k_dots'), 'axpy', g('k_axpy'))"; }
for v in ${AB_VARIANTS:-"-DFUSED_WPE=0" "-DFUSED_WPE=6"}; do
  v=${v//,/ }
  (cd psi-gnn_amd/csrc && rm -f fgnn_tile.o && make EXTRA="$v" > /dev/null 2>&1) || { echo "build failed: $v"; continue; }
  a=$(timeout -k 10 200 python3 bench.py --steps 50 --warmup 5 --repeats 3 --no-cpu-baseline 2>/dev/null | line)
  b=$(timeout -k 10 200 python3 bench.py --workload batch50k --steps 100 --warmup 5 --repeats 3 --no-cpu-baseline 2>/dev/null | line)
  echo "$v | mesh1m: $a | batch50k: $b"
done
(cd psi-gnn_amd/csrc && rm -f fgnn_tile.o && make > /dev/null 2>&1)
