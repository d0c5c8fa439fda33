#!/bin/bash
# Runs ON THE GPU BOX: A/B of build-time variants of the tiled JVP / VJP kernels (plan-order products at 1M nodes, 2 runs each).
cd "$GRAFT_REPO_ROOT"
for v in ${AB_VARIANTS:-"-DJVP_WAVES=0"}; do
  v=${v//,/ }
  (cd psi-gnn_amd/csrc && rm -f fgnn_tile_jvp.o fgnn_tile_vjp.o && make EXTRA="$v" > /dev/null 2>&1) || { echo "build failed: $v"; continue; }
  r=""
  for i in 1 2; do r="$r | $(timeout -k 10 120 python3 scripts/prof_f.py 1000000 30 0 ${AB_BC:-dirichlet} adjoint 2>/dev/null | grep -E 'plan order' | sed 's/ (plan order) avg//' | tr '\n' ' ')"; done
  echo "$v $r"
done
(cd psi-gnn_amd/csrc && rm -f fgnn_tile_jvp.o fgnn_tile_vjp.o && make > /dev/null 2>&1)
