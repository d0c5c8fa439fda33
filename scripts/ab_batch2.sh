#!/bin/bash
# per-kernel achieved fractions of bench.py for "<env> -- <bench args>" entries
cd "$(dirname "$0")/.."
for cfg in "$@"; do
  envs="${cfg%%--*}"; args="${cfg#*--}"
  out=$(env $envs timeout -k 10 300 python3 bench.py $args --repeats 3 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
ks={x['kernel']:x for x in d['kernels']}
g=lambda n:((round(ks[n]['avg_us'],1), round(ks[n].get('frac_of_8TBps',0),3)) if n in ks else None)
print('it/s', round(d['iters_per_sec'],1), 'dots', g('k_dots'), 'axpy', g('k_axpy'), 'f', g('f(k_f_tile_fused)'), 'final', g('k_final'), 'rc', g('k_reduce_check'))")
  echo "$cfg | $out"
done
