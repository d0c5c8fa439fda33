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
