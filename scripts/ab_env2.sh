#!/bin/bash
# Runs ON THE GPU BOX: whole-iteration time of bench.py under environment variants ("ENV=.. -- bench args"), 2 runs each.
cd "$GRAFT_REPO_ROOT"
for cfg in "$@"; do
  envs="${cfg%%--*}"; args="${cfg#*--}"
  r=""
  for i in 1 2; do
    r="$r | $(env $envs timeout -k 10 300 python3 bench.py $args --repeats 5 --no-cpu-baseline --no-kernel-timing 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('it/s', round(d['iters_per_sec'],1), 'ms', d['ms_per_step_spread'], 'rel', d.get('rel_residual_after_K'))")"
  done
  echo "$cfg $r"
done
