#!/bin/bash
# Runs ON THE GPU BOX: A/B of build-time variants of the f tile kernel (plain f at 1M nodes, 3 runs each; fused step via bench).
cd "$GRAFT_REPO_ROOT"
for v in ${AB_VARIANTS:-"-DMV2_CH=0" "-DMV2_CH=4"}; do
  v=${v//,/ }
  (cd psi-gnn_amd/csrc && rm -f fgnn_tile.o && make EXTRA="$v" > /dev/null 2>&1) || { echo "build failed: $v"; continue; }
  f=""
  for i in 1 2 3; do f="$f $(timeout -k 10 120 python3 scripts/prof_f.py 1000000 50 0 dirichlet 2>/dev/null | grep -o 'f avg [0-9.]* us' | grep -o '[0-9.]*')"; done
  b=$(timeout -k 10 200 python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=[x for x in d['kernels'] if 'fused' in x['kernel']][0]
print('it/s', round(d['iters_per_sec'],1), 'fused_us', round(k['avg_us'],1))")
  echo "$v | plain f us:$f | $b"
done
(cd psi-gnn_amd/csrc && rm -f fgnn_tile.o && make > /dev/null 2>&1)
