"""Node-block-diagonal Newton map for ``utilities.solver.newton`` (reference: utilities/solver.py:349-366).

The reference forms the dense (N,d,N,d) Jacobian of ``f_root(z) = f(z) - z`` with autograd, keeps the
per-node d x d blocks ``J[n,:,n,:]`` and iterates ``z <- z - blocks^{-1} f_root(z)``.  Here the blocks come
from the analytic JVP kernel: with a proper colouring of the mesh graph (no edge joins two nodes of one
colour; f[n] only depends on h[n] and h[neighbours of n]), the JVP along ``e_j`` restricted to one colour
class gives column j of every block of that class at once: d x (#colours) JVPs, O(N d^2) memory.
"""
from __future__ import annotations

import numpy as np
import torch

from ..engine import D, FixedPointMap


def greedy_colouring(csr_ptr, csr_nbr, csc_ptr, csc_nbr, n):
    """Greedy colouring of the undirected support of the edge set (host side, one-time, integer work)."""
    colour = np.full(n, -1, dtype=np.int32)
    for v in range(n):
        nb = np.concatenate([csr_nbr[csr_ptr[v]:csr_ptr[v + 1]], csc_nbr[csc_ptr[v]:csc_ptr[v + 1]]])
        used = set(colour[nb][colour[nb] >= 0].tolist())
        c = 0
        while c in used:
            c += 1
        colour[v] = c
    return colour


def node_jacobian_blocks(fmap: FixedPointMap, z):
    """(N, d, d) tensor B with B[n, i, j] = d f[n, i] / d z[n, j]."""
    plan = fmap.plan
    if not hasattr(plan, "_colours"):
        col = greedy_colouring(plan.export("csr_ptr"), plan.export("csr_nbr"), plan.export("csc_ptr"),
                               plan.export("csc_nbr"), plan.N)
        plan._colours = torch.from_numpy(col).to(z.device)
    colours = plan._colours
    nc = int(colours.max().item()) + 1
    B = torch.zeros(plan.N, D, D, device=z.device, dtype=z.dtype)
    for c in range(nc):
        mask = (colours == c)
        for j in range(D):
            v = torch.zeros_like(z)
            v[mask, j] = 1.0
            jv = fmap.jvp(z, v)
            B[mask, :, j] = jv[mask]
    return B


def block_newton_map(fmap: FixedPointMap):
    """z -> z - (blockdiag(J_f) - I)^{-1} (f(z) - z)."""
    eye = None

    def g(z):
        nonlocal eye
        if eye is None:
            eye = torch.eye(D, device=z.device, dtype=z.dtype)[None]
        blocks = node_jacobian_blocks(fmap, z) - eye
        r = fmap(z) - z
        return z - torch.linalg.solve(blocks, r[:, :, None])[:, :, 0]

    return g
