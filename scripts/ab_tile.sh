#!/bin/bash
# Runs ON THE GPU BOX: plain f at 1M nodes for several tile sizes (nodes per tile), 2 runs each.
cd "$GRAFT_REPO_ROOT"
for t in ${AB_TILES:-256 240 224 208 200 196 192 176 160}; do
  f=""
  for i in 1 2; do f="$f $(timeout -k 10 120 python3 scripts/prof_f.py ${AB_NODES:-1000000} 50 $t dirichlet 2>/dev/null | grep -o 'tiles=[0-9]* max_rows=[0-9]*.*f avg [0-9.]* us')"; done
  echo "tile_target=$t | $f"
done
