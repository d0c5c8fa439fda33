#!/bin/bash
# Runs ON THE GPU BOX: wave priorities per phase of k_f_tile (fgnn_tile.hip TILE_PRIO = p0 + 4 p1 + 16 p2 + 64 p3)
cd "$GRAFT_REPO_ROOT"
for v in ${AB_PRIOS:-0 3 12 48 192 15 60 51 195}; do
  (cd psi-gnn_amd/csrc && rm -f fgnn_tile.o && make EXTRA="-DTILE_PRIO=$v" > /dev/null 2>&1) || { echo "build failed: $v"; continue; }
  f=""
  for i in 1 2; do f="$f $(timeout -k 10 120 python3 scripts/prof_f.py 1000000 50 0 dirichlet 2>/dev/null | grep -o 'f avg [0-9.]* us' | grep -o '[0-9.]*')"; done
  b=$(timeout -k 10 200 python3 bench.py --steps 20 --warmup 5 --repeats 3 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=[x for x in d['kernels'] if 'fused' in x['kernel']][0]
print('it/s', round(d['iters_per_sec'],1), 'fused_us', round(k['avg_us'],1))")
  echo "TILE_PRIO=$v (stage1 $((v&3)) walk $(((v>>2)&3)) update $(((v>>4)&3)) epilogue $(((v>>6)&3))) | plain f us:$f | $b"
done
(cd psi-gnn_amd/csrc && rm -f fgnn_tile.o && make > /dev/null 2>&1)
