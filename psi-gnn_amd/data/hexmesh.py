"""Closed-form synthetic P1 Poisson meshes in the reference reader's tensor schema.

The reference builds its datasets with gmsh + FEniCS (``dirichlet/dataset/build_mesh.py``,
``extract_data.py:16-102``; mixed: ``mixed/dataset/extract_data.py:18-117``), neither of which
exists here.  This module reproduces the *schema and statistics* of those datasets on a
deterministic mesh: a regular hexagon tiled with equilateral triangles of side
``h = 0.08`` (the reference's ``hsize``), smoothly warped, with

* own P1 stiffness assembly and consistent-mass load ``b = M f`` (FEniCS: ``a = inner(grad u, grad v) dx``,
  ``L = f v dx``, ``extract_data.py:56-59``),
* Dirichlet rows replaced by identity rows with ``b_i = g_i`` (``bc.apply(A, b)``, ``:64``),
* edges = non-zero pattern of ``A`` *including the diagonal*, row-major (``find(A)``, ``:86-87``),
* ``distance_e = (pos[row]-pos[col], |.|)`` (``compute_position``, ``:92-102``),
* ``prb_data``/``tags`` as ``extract_data.py:75-84`` (mixed ``:82-100``),
* the reader's hard-coded normalisation and initial guess (``reader.py:73-110``; mixed ``:74-118``).

Sizes: ``N = 3 n^2 + 3 n + 1`` nodes, ``6 n^2`` triangles.  n=26 -> 2 107, n=129 -> 50 311,
n=182 -> 99 919, n=577 -> 1 000 519.
"""
from __future__ import annotations

import numpy as np
import scipy.sparse as sp
import torch

from .meshdata import MeshData

HSIZE = 0.08  # dirichlet/dataset/generate_data.py:37

# reader.py:73-77 (dirichlet)
DIR_PRB_MEAN = np.array([0.0464, -0.0006])
DIR_PRB_STD = np.array([9.6267, 3.2935])
DIR_DIST_MEAN = np.array([0.0, 0.0, 0.0655])
DIR_DIST_STD = np.array([0.0507, 0.0507, 0.0293])
# mixed reader.py:74-81
MIX_PRB_MEAN = np.array([-0.4319, 0.0289, -0.0189])
MIX_PRB_STD = np.array([8.4245, 2.1942, 2.8585])
MIX_DIST_MEAN = np.array([0.0, 0.0, 0.0572])
MIX_DIST_STD = np.array([0.0445, 0.0443, 0.0258])
MIX_NRM_MEAN = np.array([0.0007, -0.0004])
MIX_NRM_STD = np.array([0.2773, 0.2959])


def hex_n_for_nodes(target_nodes: int) -> int:
    """Smallest n with 3n^2+3n+1 >= target_nodes."""
    n = int(np.ceil((-3 + np.sqrt(max(9 - 12 * (1 - target_nodes), 0))) / 6))
    while 3 * n * n + 3 * n + 1 < target_nodes:
        n += 1
    return max(n, 1)


def hex_lattice(n: int, h: float = HSIZE, warp: float = 0.15, phase: float = 0.0):
    """Nodes and triangles of the warped hexagon lattice.

    Returns (pos (N,2) f64, tri (T,3) i64, q (N,), r (N,)).  Nodes are numbered row by row
    (r ascending, then q ascending).
    """
    qs, rs = [], []
    for r in range(-n, n + 1):
        q0, q1 = max(-n, -n - r), min(n, n - r)
        qq = np.arange(q0, q1 + 1)
        qs.append(qq)
        rs.append(np.full_like(qq, r))
    q = np.concatenate(qs)
    r = np.concatenate(rs)
    N = q.size
    # (q, r) -> id map on a dense (2n+2)^2 grid
    W = 2 * n + 3
    gid = -np.ones((W, W), dtype=np.int64)
    gid[r + n, q + n] = np.arange(N)

    def node(qq, rr):
        ok = (np.abs(qq) <= n) & (np.abs(rr) <= n) & (np.abs(qq + rr) <= n)
        out = np.full(qq.shape, -1, dtype=np.int64)
        out[ok] = gid[rr[ok] + n, qq[ok] + n]
        return out

    a = np.arange(N)
    up = np.stack([a, node(q + 1, r), node(q, r + 1)], axis=1)
    dn = np.stack([a, node(q + 1, r - 1), node(q + 1, r)], axis=1)
    tri = np.concatenate([up[(up >= 0).all(1)], dn[(dn >= 0).all(1)]], axis=0)

    x = h * (q + 0.5 * r)
    y = h * (np.sqrt(3.0) / 2.0) * r
    L = n * h
    if warp:
        dx = warp * h * np.sin(2.3 * np.pi * y / L + 0.3 + phase)
        dy = warp * h * np.sin(1.7 * np.pi * x / L + 1.1 + phase)
        x, y = x + dx, y + dy
    pos = np.stack([x, y], axis=1)
    return pos, tri, q, r


def p1_assemble(pos: np.ndarray, tri: np.ndarray):
    """P1 stiffness K and consistent mass M (scipy CSR, float64).  SURVEY Appendix A.2."""
    p0, p1, p2 = pos[tri[:, 0]], pos[tri[:, 1]], pos[tri[:, 2]]
    b = np.stack([p1[:, 1] - p2[:, 1], p2[:, 1] - p0[:, 1], p0[:, 1] - p1[:, 1]], axis=1)
    c = np.stack([p2[:, 0] - p1[:, 0], p0[:, 0] - p2[:, 0], p1[:, 0] - p0[:, 0]], axis=1)
    area2 = (p1[:, 0] - p0[:, 0]) * (p2[:, 1] - p0[:, 1]) - (p2[:, 0] - p0[:, 0]) * (p1[:, 1] - p0[:, 1])
    area = 0.5 * np.abs(area2)
    Kloc = (b[:, :, None] * b[:, None, :] + c[:, :, None] * c[:, None, :]) / (4.0 * area)[:, None, None]
    Mloc = (area / 12.0)[:, None, None] * (np.ones((3, 3)) + np.eye(3))[None]
    I = np.repeat(tri[:, :, None], 3, axis=2).ravel()
    J = np.repeat(tri[:, None, :], 3, axis=1).ravel()
    N = pos.shape[0]
    K = sp.coo_matrix((Kloc.ravel(), (I, J)), shape=(N, N)).tocsr()
    M = sp.coo_matrix((Mloc.ravel(), (I, J)), shape=(N, N)).tocsr()
    return K, M


def boundary_edges(tri: np.ndarray):
    """Edges that belong to exactly one triangle, as (B,2) node pairs (oriented as in the triangle)."""
    e = np.concatenate([tri[:, [0, 1]], tri[:, [1, 2]], tri[:, [2, 0]]], axis=0)
    key = np.sort(e, axis=1)
    N1 = int(key.max()) + 1
    code = key[:, 0] * N1 + key[:, 1]
    _, inv, cnt = np.unique(code, return_inverse=True, return_counts=True)
    return e[cnt[inv] == 1]


def _problem_coeffs(seed: int):
    rng = np.random.default_rng(seed)
    return rng.uniform(-10, 10, 3), rng.uniform(-10, 10, 6)


def _f_expr(pf, x, y):  # extract_data.py:23-26
    return pf[0] * (x - 1.0) ** 2 + pf[1] * y ** 2 + pf[2]


def _g_expr(pg, x, y):  # extract_data.py:28-32
    return pg[0] * x * x + pg[1] * x * y + pg[2] * y * y + pg[3] * x + pg[4] * y + pg[5]


def _apply_dirichlet(K: sp.csr_matrix, rhs: np.ndarray, dmask: np.ndarray, gvals: np.ndarray):
    """FEniCS ``bc.apply(A, b)``: Dirichlet rows -> identity rows (columns untouched), b_i = g_i."""
    keep = sp.diags((~dmask).astype(np.float64))
    A = (keep @ K + sp.diags(dmask.astype(np.float64))).tocsr()
    A.eliminate_zeros()  # csr_matrix(dense) + find() drop exact zeros (extract_data.py:70-71,86)
    A.sort_indices()
    rhs = rhs.copy()
    rhs[dmask] = gvals[dmask]
    return A, rhs


def _edges_of(A: sp.csr_matrix, pos: np.ndarray):
    A = A.tocoo()
    order = np.lexsort((A.col, A.row))  # row-major, as this scipy's find() returns
    row, col, val = A.row[order].astype(np.int64), A.col[order].astype(np.int64), A.data[order]
    d = pos[row] - pos[col]
    dist = np.concatenate([d, np.sqrt((d ** 2).sum(1, keepdims=True))], axis=1)
    return row, col, val, dist


def _solve(A, rhs, compute_sol):
    if not compute_sol:
        return np.zeros_like(rhs)
    from scipy.sparse.linalg import spsolve
    return spsolve(A.tocsc(), rhs)


def make_from_triangulation(pos, tri, dirichlet_mask, seed=0, radius=1.0, mixed=False,
                            normals=None, compute_sol=True, dtype=torch.float32, coeffs=None) -> MeshData:
    """Assemble one Poisson problem on an arbitrary P1 triangulation and emit the reader schema.
    ``coeffs`` = (param_f[3], param_g[6]) overrides the seeded draw (to rebuild a problem the reference recorded)."""
    N = pos.shape[0]
    K, M = p1_assemble(pos, tri)
    pf, pg = _problem_coeffs(seed) if coeffs is None else (np.asarray(coeffs[0], float), np.asarray(coeffs[1], float))
    xs, ys = pos[:, 0] / radius, pos[:, 1] / radius
    fv = _f_expr(pf, xs, ys)
    gv = _g_expr(pg, xs, ys)
    rhs = M @ fv
    A, rhs = _apply_dirichlet(K, rhs, dirichlet_mask, gv)
    sol = _solve(A, rhs, compute_sol)
    row, col, val, dist = _edges_of(A, pos)

    if not mixed:
        prb = np.stack([np.where(dirichlet_mask, 0.0, fv), np.where(dirichlet_mask, gv, 0.0)], axis=1)
        tags = dirichlet_mask.astype(np.float64)[:, None]
        prb = (prb - DIR_PRB_MEAN) / DIR_PRB_STD
        ea = (dist - DIR_DIST_MEAN) / DIR_DIST_STD
        unv = None
    else:
        bnd = np.zeros(N, dtype=bool)
        bnd[boundary_edges(tri).ravel()] = True
        neu = bnd & ~dirichlet_mask
        interior = ~bnd
        prb = np.stack([np.where(interior, fv, 0.0), np.where(dirichlet_mask, gv, 0.0),
                        np.where(neu, fv, 0.0)], axis=1)
        tags = np.stack([interior, dirichlet_mask, neu], axis=1).astype(np.float64)
        prb = (prb - MIX_PRB_MEAN) / MIX_PRB_STD
        ea = (dist - MIX_DIST_MEAN) / MIX_DIST_STD
        unv = (normals - MIX_NRM_MEAN) / MIX_NRM_STD

    t = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=dtype)
    x0 = np.where(dirichlet_mask, rhs, 0.0)[:, None]
    data = MeshData(
        x=t(x0), edge_index=torch.tensor(np.stack([row, col]), dtype=torch.long),
        edge_attr=t(ea), a_ij=t(val[:, None]), y=t(rhs[:, None]), sol=t(sol[:, None]),
        prb_data=t(prb), tags=t(tags), pos=t(pos))
    if unv is not None:
        data.unit_normal_vector = t(unv)
    return data


def vertex_normals(pos, tri):
    """Outward unit normals at boundary vertices (length-weighted average of the two adjacent
    boundary-edge normals, renormalised); zero at interior vertices.  Stands in for the
    reference's L2-projected FacetNormal (``mixed/dataset/extract_data.py:119-137``)."""
    N = pos.shape[0]
    be = boundary_edges(tri)
    t = pos[be[:, 1]] - pos[be[:, 0]]
    nrm = np.stack([t[:, 1], -t[:, 0]], axis=1)  # right of the ccw-oriented edge = outward
    # orientation: triangles may be cw; decide by the domain centroid
    mid = 0.5 * (pos[be[:, 0]] + pos[be[:, 1]])
    flip = ((mid - pos.mean(0)) * nrm).sum(1) < 0
    nrm[flip] *= -1
    acc = np.zeros((N, 2))
    np.add.at(acc, be[:, 0], nrm)
    np.add.at(acc, be[:, 1], nrm)
    ln = np.linalg.norm(acc, axis=1, keepdims=True)
    return np.divide(acc, ln, out=np.zeros_like(acc), where=ln > 0)


def make_hex_problem(n: int, seed: int = 0, mixed: bool = False, warp: float = 0.15,
                     phase: float | None = None, compute_sol: bool = True,
                     dtype=torch.float32, hsize: float = HSIZE, radius: float | None = None) -> MeshData:
    """One synthetic Poisson problem on the warped hexagon with ``3n^2+3n+1`` nodes.

    dirichlet: whole boundary ring is Dirichlet.  mixed: the six sides alternate
    Dirichlet / Neumann (corner nodes Dirichlet), homogeneous Neumann.
    ``hsize``: triangle side (default 0.08, the dirichlet dataset's ``hsize``, dirichlet/dataset/generate_data.py:37; the mixed
    training set's mean edge length is 0.068 by its normalisation constants, mixed/psignn/utilities/reader.py:77).
    ``radius``: what the coordinates are divided by inside the problem's f and g (default: the hexagon's side n * hsize).
    """
    if phase is None:
        phase = 0.0
    pos, tri, q, r = hex_lattice(n, hsize, warp, phase)
    s = -q - r
    ring = np.maximum(np.maximum(np.abs(q), np.abs(r)), np.abs(s)) == n
    if radius is None:   # the (f, g) fields are functions of pos / radius (extract_data.py:19-32; the reference passes its nominal domain radius 1.0)
        radius = n * hsize
    if not mixed:
        return make_from_triangulation(pos, tri, ring, seed=seed, radius=radius, mixed=False,
                                       compute_sol=compute_sol, dtype=dtype)
    # side k of the hexagon: the constraint that is active; corners belong to two sides
    sides = [q == n, s == -n, r == n, q == -n, s == n, r == -n]
    dmask = np.zeros(q.size, dtype=bool)
    for k in (0, 2, 4):
        dmask |= sides[k]
    normals = vertex_normals(pos, tri)
    return make_from_triangulation(pos, tri, dmask, seed=seed, radius=radius, mixed=True,
                                   normals=normals, compute_sol=compute_sol, dtype=dtype)
