"""Dataset reader: the reference's raw ``.npy`` lists -> lists of ``MeshData`` (no torch_geometric).

Mirror of ``BuildDataset.process`` (``dirichlet/psignn/utilities/reader.py:61-131``; mixed ``:61-139``):

* edges = ``scipy.sparse.find(A)`` incl. the diagonal, ``a_ij`` its values (``:82-85``);
* ``prb_data`` and ``distance`` normalised with the reference's hard-coded means / stds (``:73-77``; mixed ``:74-81``
  plus the unit normals);
* initial guess ``x``: zeros, ``b`` on the Dirichlet rows (``:108-110``; mixed ``tags[:,1]``);
* split 60 / 20 / 20 by ``train_test_split(test_size=0.2)`` then ``(test_size=0.25)``: ``shuffle=False`` for the dirichlet
  reader (``:116-117``), ``shuffle=True`` (unseeded) for the mixed one (``:125-126``) — reproduced for ``shuffle=False``;
  for the mixed family pass a ``seed`` or take all samples.

The reference stores Python lists of scipy matrices / arrays with ``np.save`` (pickled object arrays).  ``load_raw_dir``
therefore needs ``allow_pickle=True`` and is meant for datasets YOU generated with the reference's ``generate_data.py``;
``from_arrays`` takes already-loaded objects.
"""
from __future__ import annotations

import os

import numpy as np
import scipy.sparse as sp
import torch

from ..data.hexmesh import (DIR_DIST_MEAN, DIR_DIST_STD, DIR_PRB_MEAN, DIR_PRB_STD, MIX_DIST_MEAN, MIX_DIST_STD,
                            MIX_NRM_MEAN, MIX_NRM_STD, MIX_PRB_MEAN, MIX_PRB_STD)
from ..data.meshdata import MeshData

RAW_FILES = ["A_sparse_matrix.npy", "b_matrix.npy", "sol.npy", "prb_data.npy", "tags.npy", "coordinates.npy",
             "distance.npy"]
RAW_FILES_MIXED = RAW_FILES + ["unit_normal_vector.npy"]


def from_arrays(A, b, sol, prb_data, tags, coordinates, distance, unit_normal_vector=None, dtype=torch.float32) -> MeshData:
    """One sample of the raw lists -> ``MeshData`` (the body of the reader's per-sample loop)."""
    mixed = unit_normal_vector is not None
    r, c, v = sp.find(sp.csr_matrix(A))
    order = np.lexsort((c, r))  # row-major; the models do not depend on the edge order
    r, c, v = r[order].astype(np.int64), c[order].astype(np.int64), v[order]
    distance = np.asarray(distance, dtype=np.float64)
    if distance.shape[0] != r.shape[0]:
        raise ValueError(f"distance has {distance.shape[0]} rows but A has {r.shape[0]} non-zeros")
    # distance rows follow find(A)'s order at generation time (extract_data.py:86-88); re-key them by (row, col)
    pos = np.asarray(coordinates, dtype=np.float64)
    d_expected = pos[r] - pos[c]
    if not np.allclose(distance[:, :2], d_expected, atol=1e-9):
        r0, c0, _ = sp.find(sp.csr_matrix(A))
        key = {(int(a), int(bb)): i for i, (a, bb) in enumerate(zip(r0, c0))}
        distance = distance[[key[(int(a), int(bb))] for a, bb in zip(r, c)]]
    t = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=dtype)
    b = np.asarray(b, dtype=np.float64).reshape(-1, 1)
    tags = np.asarray(tags, dtype=np.float64)
    if mixed:
        prb = (np.asarray(prb_data) - MIX_PRB_MEAN) / MIX_PRB_STD
        ea = (distance - MIX_DIST_MEAN) / MIX_DIST_STD
        dmask = tags[:, 1] == 1
    else:
        prb = (np.asarray(prb_data) - DIR_PRB_MEAN) / DIR_PRB_STD
        ea = (distance - DIR_DIST_MEAN) / DIR_DIST_STD
        tags = tags.reshape(-1, 1)
        dmask = tags[:, 0] == 1
    x = np.where(dmask[:, None], b, 0.0)
    m = MeshData(x=t(x), edge_index=torch.tensor(np.stack([r, c]), dtype=torch.long), edge_attr=t(ea),
                 a_ij=t(v[:, None]), y=t(b), sol=t(np.asarray(sol).reshape(-1, 1)), prb_data=t(prb), tags=t(tags), pos=t(pos))
    if mixed:
        m.unit_normal_vector = t((np.asarray(unit_normal_vector) - MIX_NRM_MEAN) / MIX_NRM_STD)
    return m


def split_indices(n, mode, shuffle=False, seed=None):
    """60/20/20 split of ``range(n)`` as two chained ``train_test_split`` calls (reader.py:116-117)."""
    from sklearn.model_selection import train_test_split
    idx = list(range(n))
    kw = dict(shuffle=shuffle, random_state=seed if shuffle else None)
    train_, test = train_test_split(idx, test_size=0.2, **kw)
    train, val = train_test_split(train_, test_size=0.25, **kw)
    return {"train": train, "val": val, "test": test, "all": idx}[mode]


def load_raw_dir(root, mode="test", mixed=False, shuffle=None, seed=None, dtype=torch.float32, allow_pickle=False):
    """``<root>/data/*.npy`` as written by the reference's ``generate_data.py`` -> list of ``MeshData``.

    Those files are pickled object arrays (lists of scipy CSR matrices and ragged arrays), which numpy can only read by
    UNPICKLING -- i.e. by running code from the file.  That is never done implicitly: the default ``allow_pickle=False``
    makes numpy refuse such files with its own error, and only a caller who generated the files themselves opts in with
    ``allow_pickle=True``.  Plain numeric ``.npy`` / ``.npz`` exports load without it."""
    raw = os.path.join(root, "data")
    names = RAW_FILES_MIXED if mixed else RAW_FILES
    try:
        lists = [np.load(os.path.join(raw, f), allow_pickle=bool(allow_pickle)) for f in names]
    except ValueError as e:
        raise ValueError(f"{e}.  The reference's raw dataset files are pickled object arrays; pass allow_pickle=True only for "
                         "files you generated yourself") from e
    n = len(lists[0])
    if shuffle is None:
        shuffle = mixed
    keep = split_indices(n, mode, shuffle=shuffle, seed=seed)
    return [from_arrays(*(l[i] for l in lists), dtype=dtype) for i in keep]
