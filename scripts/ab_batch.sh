#!/bin/bash
# A/B of the sweep shapes of the batched shard solver (bench.py --workload batch50k, K = 100): one line per environment.
#   bash scripts/ab_batch.sh "PSIGNN_JGROUPS=2" "PSIGNN_VEC16_MIN=99999999" ...
cd "$(dirname "$0")/.."
for cfg in "$@"; do
  out=$(env $cfg timeout -k 10 200 python3 bench.py --workload batch50k --steps 100 --warmup 5 --repeats 3 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
ks={x['kernel']:x for x in d['kernels']}
g=lambda n:(round(ks[n]['avg_us'],1) if n in ks else None)
print('it/s', round(d['iters_per_sec'],1), 'dots', g('k_dots'), 'axpy', g('k_axpy'), 'comb', g('k_axpy_combine'), 'f', g('f(k_f_tile_fused)'), 'final', g('k_final'), 'rc', g('k_reduce_check'))")
  echo "$cfg | $out"
done
