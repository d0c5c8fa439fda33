#!/bin/bash
# Runs ON THE GPU BOX: batched fused f kernel free (87 VGPRs, five waves per SIMD, 1 280 workgroup slots for the shard's 1 576 tiles) vs held to
# six waves (-DBATCH_WPE=6: 80 VGPRs + 36 bytes of scratch, 1 536 slots), interleaved
cd "$GRAFT_REPO_ROOT"
for v in 0 6 0 6; do
  (cd psi-gnn_amd/csrc && rm -f fgnn_tile.o && make EXTRA="-DBATCH_WPE=$v" > /dev/null 2>&1) || { echo "build failed: $v"; continue; }
  for w in batch50k mixedbatch50k; do
  echo "BATCH_WPE=$v $w: $(timeout -k 10 200 python3 bench.py --workload $w --steps 100 --warmup 5 --repeats 3 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=[x for x in d['kernels'] if 'fused' in x['kernel']][0]
print('it/s', round(d['iters_per_sec'],1), 'fused_us', round(k['avg_us'],1), round(k.get('frac_of_8TBps',0),3))")"
  done
done
(cd psi-gnn_amd/csrc && rm -f fgnn_tile.o && make > /dev/null 2>&1)
