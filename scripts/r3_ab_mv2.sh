#!/bin/bash
# Runs ON THE GPU BOX: A/B of where the weight loads of mv2 are pinned (tile_helpers.h MV2_LAUNDER / MV2_CH; static counts of the
# SGPR-spill lane operations per variant in DESIGN section 4): plain f, fused step (bench K = 20), linearised and direct JVP.
cd "$GRAFT_REPO_ROOT"
run() {
  f=""
  for i in 1 2; do f="$f $(timeout -k 10 120 python3 scripts/prof_f.py 1000000 50 0 dirichlet adjoint 2>/dev/null | grep -E '^N=|lin jvp|jvp_p' | grep -o 'avg [0-9.]* us' | grep -o '[0-9.]*' | tr '\n' '/')"; done
  b=$(timeout -k 10 200 python3 bench.py --steps 20 --warmup 5 --repeats 3 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=[x for x in d['kernels'] if 'fused' in x['kernel']][0]
print('it/s', round(d['iters_per_sec'],1), 'fused_us', round(k['avg_us'],1))")
  b2=$(timeout -k 10 200 python3 bench.py --workload batch50k --steps 100 --warmup 5 --repeats 3 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=[x for x in d['kernels'] if 'fused' in x['kernel']][0]
print('batch it/s', round(d['iters_per_sec'],1), 'fused_us', round(k['avg_us'],1))")
  echo "$1 | f / lin jvp / jvp_p us:$f | $b | $b2"
}
for v in "-DMV2_LAUNDER=0" "-DMV2_LAUNDER=1" "-DMV2_LAUNDER=2" "-DMV2_LAUNDER=2,-DMV2_CH=10" "-DMV2_LAUNDER=1,-DMV2_CH=2"; do
  vv=${v//,/ }
  (cd psi-gnn_amd/csrc && rm -f fgnn_tile.o fgnn_tile_lin.o fgnn_tile_jvp.o && make EXTRA="$vv" > /dev/null 2>&1) || { echo "build failed: $v"; continue; }
  run "$v"
done
(cd psi-gnn_amd/csrc && rm -f fgnn_tile.o fgnn_tile_lin.o fgnn_tile_jvp.o && make > /dev/null 2>&1)
