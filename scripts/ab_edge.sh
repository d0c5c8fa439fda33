#!/bin/bash
# Runs ON THE GPU BOX: A/B of build-time variants of k_f_tile (rebuilds fgnn_tile.o per variant; plain f at 1M nodes
# and the fused kernel inside the default bench).  AB_VARIANTS="-DX=0 -DX=1" overrides the list.
cd "$GRAFT_REPO_ROOT"
for v in ${AB_VARIANTS:-"-DEDGE_CLAMP=0" "-DEDGE_CLAMP=1"}; do
  v=${v//,/ }   # "-DA=1,-DB=2" = one variant with two defines
  (cd psi-gnn_amd/csrc && rm -f fgnn_tile.o && make EXTRA="$v" > /dev/null 2>&1) || { echo "build failed: $v"; continue; }
  f=$(timeout -k 10 120 python3 scripts/prof_f.py 1000000 20 0 dirichlet 2>/dev/null | grep -o "f avg [0-9.]* us")
  b=$(timeout -k 10 200 python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=[x for x in d['kernels'] if 'fused' in x['kernel']][0]
print('it/s', round(d['iters_per_sec'],1) if 'iters_per_sec' in d else d['value'], 'fused_us', round(k['avg_us'],1))")
  echo "$v | $f | $b"
done
(cd psi-gnn_amd/csrc && rm -f fgnn_tile.o && make > /dev/null 2>&1)
