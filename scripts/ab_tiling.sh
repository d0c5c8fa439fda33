#!/bin/bash
# Runs ON THE GPU BOX: plain f / JVP / VJP at 1M nodes and the default bench for each tiling scheme.
cd "$GRAFT_REPO_ROOT"
for v in PSIGNN_TILING=strips PSIGNN_TILING=hilbert; do
  f=$(env $v timeout -k 10 200 python3 scripts/prof_f.py 1000000 50 0 dirichlet adjoint 2>/dev/null | tr '\n' ' ')
  b=$(env $v timeout -k 10 200 python3 bench.py --steps 50 --warmup 5 --repeats 3 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=[x for x in d['kernels'] if 'fused' in x['kernel']][0]
print('it/s', round(d['iters_per_sec'],1), 'fused_us', round(k['avg_us'],1), 'f_only', round(d['f_only']['us_per_eval'],1), 'jvp', round(d['roofline_jvp']['avg_launch_us'],1), 'tiles', d['config']['tiles'], 'max_rows', d['config']['max_tile_rows'])")
  echo "$v | $f | $b"
done
