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


SLOT_IN, SLOT_OUT, ELL_EMPTY = 0x10000, 0x20000, 0xFFFF


def merge_slots(n_in, e_in, n_out, e_out, ea_bits):
    """Pair-merge of a node's canonical in-list and out-list (csrc/tiles.hip merge_slots).
    n_* = neighbour ids (OLD numbering, the order key), e_* = edge ids; ea_bits = edge_attr viewed as uint32.
    Returns [(kind, pos_in, pos_out)] with kind 1 in-only, 2 out-only, 3 merged."""
    out, i, j = [], 0, 0
    big = np.iinfo(np.int64).max
    while i < len(n_in) or j < len(n_out):
        ni = n_in[i] if i < len(n_in) else big
        nj = n_out[j] if j < len(n_out) else big
        if ni == nj:
            single = (i + 1 >= len(n_in) or n_in[i + 1] != ni) and (j + 1 >= len(n_out) or n_out[j + 1] != nj)
            ai, aj = ea_bits[e_in[i]], ea_bits[e_out[j]]
            mirror = ai[0] == (aj[0] ^ 0x80000000) and ai[1] == (aj[1] ^ 0x80000000) and ai[2] == aj[2]
            if single and mirror:
                out.append((3, i, j)); i += 1; j += 1
            else:
                out.append((1, i, j)); i += 1
        elif ni < nj:
            out.append((1, i, j)); i += 1
        else:
            out.append((2, i, j)); j += 1
    return out


def tile_reference(edge_index, n_nodes, perm, tile_ptr, edge_attr, halo_cap=512):
    """Given the plan's node order (perm[new] = old) and tile ranges, the halo lists, slice slot counts and
    pair-merged ELL slots that csrc/tiles.hip must produce (bit-exact)."""
    ref = plan_reference(edge_index, n_nodes)
    perm = np.asarray(perm, dtype=np.int64)
    inv = np.empty(n_nodes, dtype=np.int64)
    inv[perm] = np.arange(n_nodes)
    n_tiles = len(tile_ptr) - 1
    halos, halo_cnt = [], np.zeros(n_tiles, dtype=np.int32)
    tile_slice = [0]
    slice_deg, rows = [], []
    ea_bits = np.ascontiguousarray(np.asarray(edge_attr, dtype=np.float32)).view(np.uint32)
    for t in range(n_tiles):
        t0, t1 = int(tile_ptr[t]), int(tile_ptr[t + 1])
        nbrs, lists = [], []
        for new in range(t0, t1):
            old = perm[new]
            i_in = slice(ref["csc_ptr"][old], ref["csc_ptr"][old + 1])
            i_out = slice(ref["csr_ptr"][old], ref["csr_ptr"][old + 1])
            o_in, o_out = ref["csc_nbr"][i_in].astype(np.int64), ref["csr_nbr"][i_out].astype(np.int64)
            lists.append((o_in, ref["csc_eid"][i_in], o_out, ref["csr_eid"][i_out]))
            nbrs += [inv[o_in], inv[o_out]]
        alln = np.concatenate(nbrs) if nbrs else np.zeros(0, dtype=np.int64)
        h = np.unique(alln[(alln < t0) | (alln >= t1)])
        halos.append(h)
        halo_cnt[t] = len(h)
        n_t = t1 - t0
        ns = (n_t + 63) // 64
        tile_slice.append(tile_slice[-1] + ns)
        for w in range(ns):
            lanes = lists[64 * w:64 * w + 64]
            merged = [merge_slots(l[0], l[1], l[2], l[3], ea_bits) for l in lanes]
            dmax = max(len(m) for m in merged)
            slice_deg.append(dmax)
            for r in range(dmax):
                row = np.zeros((64, 4), dtype=np.uint32)
                row[:, 0] = ELL_EMPTY
                for lane, (l, m) in enumerate(zip(lanes, merged)):
                    if r < len(m):
                        kind, i, j = m[r]
                        nb = int(inv[l[2][j]] if kind & 2 else inv[l[0][i]])
                        li = nb - t0 if t0 <= nb < t1 else n_t + int(np.searchsorted(h, nb))
                        row[lane, 0] = li | (SLOT_IN if kind & 1 else 0) | (SLOT_OUT if kind & 2 else 0)
                        if kind & 2:
                            row[lane, 1:] = ea_bits[l[3][j]]
                        else:
                            a = ea_bits[l[1][i]]
                            row[lane, 1:] = (a[0] ^ 0x80000000, a[1] ^ 0x80000000, a[2])
                rows.append(row)
    return {"halo": halos, "halo_cnt": halo_cnt, "tile_slice": np.array(tile_slice, dtype=np.int32),
            "slice_deg": np.array(slice_deg, dtype=np.uint8),
            "ell": np.stack(rows) if rows else np.zeros((0, 64, 4), np.uint32)}
