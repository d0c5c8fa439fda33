#!/bin/bash
# Runs ON THE GPU BOX: the bench at the driver's K = 20 and at K = 50 / 100, plus the batched shard (one line each)
cd "$GRAFT_REPO_ROOT"
line() { python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
ks={x['kernel']:x for x in d['kernels']}
g=lambda n:((round(ks[n]['avg_us'],1), round(ks[n].get('frac_of_8TBps',0),3)) if n in ks else None)
print('it/s', round(d['iters_per_sec'],1), 'ms/step', d['ms_per_step_all'], 'u2d', g('k_sweep_u2d'), 'v', g('k_sweep_v'), 'u1', g('k_sweep_u1'), 'f', g('k_f_tile_fused'), 'red', g('k_reduce_check'), g('k_reduce_cb'))"; }
for K in 20 50 100; do
  echo "K=$K mesh1m: $(timeout -k 10 200 python3 bench.py --steps $K --warmup 5 --repeats 3 --no-cpu-baseline 2>/dev/null | line)"
done
for w in batch50k mixedbatch50k dir100k mixed100k; do
  echo "K=100 $w: $(timeout -k 10 200 python3 bench.py --workload $w --steps 100 --warmup 5 --repeats 3 --no-cpu-baseline 2>/dev/null | line)"
done
