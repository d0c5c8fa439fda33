#!/bin/bash
# Runs ON THE GPU BOX: wave priority in the memory-issuing phases of k_jvp_lin / k_jvp_tile (LIN_PRIO, JVP_PRIO), interleaved
cd "$GRAFT_REPO_ROOT"
for v in 0 1 0 1 0 1; do
  (cd psi-gnn_amd/csrc && rm -f fgnn_tile_lin.o fgnn_tile_jvp.o && make EXTRA="-DLIN_PRIO=$v -DJVP_PRIO=$v" > /dev/null 2>&1) || { echo "build failed: $v"; continue; }
  echo "PRIO=$v: $(timeout -k 10 200 python3 scripts/prof_f.py 1000000 50 0 dirichlet adjoint 2>&1 | grep -E 'lin jvp|jvp_p|lin build' | grep -o '[a-z_ ]*(plan order) avg [0-9.]* us' | tr '\n' ' ')"
done
(cd psi-gnn_amd/csrc && rm -f fgnn_tile_lin.o fgnn_tile_jvp.o && make > /dev/null 2>&1)
