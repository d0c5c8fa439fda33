"""numpy reference of the mesh plan's integer structures (what csrc/plan.hip must reproduce bit-exactly)."""
import numpy as np


def plan_reference(edge_index, n_nodes):
    row, col = np.asarray(edge_index[0], dtype=np.int64), np.asarray(edge_index[1], dtype=np.int64)
    eid = np.arange(row.size, dtype=np.int64)
    keep = row != col  # remove_self_loops (model.py:342,360)
    r, c, e = row[keep], col[keep], eid[keep]
    o = np.lexsort((e, c, r))  # group by row, then (col, edge id)
    csr_ptr = np.concatenate([[0], np.cumsum(np.bincount(r, minlength=n_nodes))]).astype(np.int32)
    o2 = np.lexsort((e, r, c))  # group by col, then (row, edge id)
    csc_ptr = np.concatenate([[0], np.cumsum(np.bincount(c, minlength=n_nodes))]).astype(np.int32)
    oa = np.lexsort((eid, col, row))
    a_ptr = np.concatenate([[0], np.cumsum(np.bincount(row, minlength=n_nodes))]).astype(np.int32)
    return {"csr_ptr": csr_ptr, "csr_nbr": c[o].astype(np.int32), "csr_eid": e[o].astype(np.int32),
            "csc_ptr": csc_ptr, "csc_nbr": r[o2].astype(np.int32), "csc_eid": e[o2].astype(np.int32),
            "a_ptr": a_ptr, "a_col": col[oa].astype(np.int32), "a_eid": eid[oa].astype(np.int32)}


def tile_reference(edge_index, n_nodes, perm, tile_ptr, edge_attr, halo_cap=512):
    """Given the plan's node order (perm[new] = old) and tile ranges, the halo lists, slice degrees and
    sliced-ELL arrays that csrc/tiles.hip must produce (bit-exact)."""
    ref = plan_reference(edge_index, n_nodes)
    perm = np.asarray(perm, dtype=np.int64)
    inv = np.empty(n_nodes, dtype=np.int64)
    inv[perm] = np.arange(n_nodes)
    n_tiles = len(tile_ptr) - 1
    halos, halo_cnt = [], np.zeros(n_tiles, dtype=np.int32)
    tile_slice = [0]
    slice_deg, rows_idx, rows_attr = [], [], []
    ea = np.asarray(edge_attr, dtype=np.float32)
    for t in range(n_tiles):
        t0, t1 = int(tile_ptr[t]), int(tile_ptr[t + 1])
        nbrs = []
        lists = []
        for new in range(t0, t1):
            old = perm[new]
            i_in = slice(ref["csc_ptr"][old], ref["csc_ptr"][old + 1])
            i_out = slice(ref["csr_ptr"][old], ref["csr_ptr"][old + 1])
            n_in, n_out = inv[ref["csc_nbr"][i_in]], inv[ref["csr_nbr"][i_out]]
            lists.append((n_in, ref["csc_eid"][i_in], n_out, ref["csr_eid"][i_out]))
            nbrs += [n_in, n_out]
        alln = np.concatenate(nbrs) if nbrs else np.zeros(0, dtype=np.int64)
        h = np.unique(alln[(alln < t0) | (alln >= t1)])
        halos.append(h)
        halo_cnt[t] = len(h)
        n_t = t1 - t0
        ns = (n_t + 63) // 64
        tile_slice.append(tile_slice[-1] + ns)
        for w in range(ns):
            lanes = lists[64 * w:64 * w + 64]
            din = max(len(l[0]) for l in lanes)
            dout = max(len(l[2]) for l in lanes)
            slice_deg.append((din, dout))
            for which, dmax in ((0, din), (2, dout)):
                for r in range(dmax):
                    idx = np.full(64, 0xFFFF, dtype=np.uint16)
                    at = np.zeros((3, 64), dtype=np.float32)
                    for lane, l in enumerate(lanes):
                        if r < len(l[which]):
                            nb = int(l[which][r])
                            idx[lane] = nb - t0 if t0 <= nb < t1 else n_t + int(np.searchsorted(h, nb))
                            at[:, lane] = ea[l[which + 1][r]]
                    rows_idx.append(idx)
                    rows_attr.append(at)
    return {"halo": halos, "halo_cnt": halo_cnt, "tile_slice": np.array(tile_slice, dtype=np.int32),
            "slice_deg": np.array(slice_deg, dtype=np.uint8).reshape(-1, 2),
            "ell_idx": np.stack(rows_idx) if rows_idx else np.zeros((0, 64), np.uint16),
            "ell_attr": np.stack(rows_attr) if rows_attr else np.zeros((0, 3, 64), np.float32)}
