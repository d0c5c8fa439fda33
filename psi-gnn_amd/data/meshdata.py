"""Mesh containers that stand in for torch_geometric ``Data`` / ``Batch``.

The reference hands its models a PyG ``Data`` object built by
``dirichlet/psignn/utilities/reader.py:103-110`` (mixed: ``:112-118``).  PyG is not a
dependency of this package, so the same attribute surface is provided here:

    x (N,1)  edge_index (2,E) int64  edge_attr (E,3)  a_ij (E,1)  y (N,1)  sol (N,1)
    prb_data (N,2|3)  tags (N,1)|(N,3)  pos (N,2)  [unit_normal_vector (N,2)]  [batch (N,)]
    num_nodes, num_edges, .to(device), .clone()

Any object with these attributes (including a real PyG ``Data``) is accepted by the
models; ``MeshData`` is only what our own generators and tests produce.
"""
from __future__ import annotations

import torch

_TENSOR_FIELDS = ("x", "edge_index", "edge_attr", "a_ij", "y", "sol", "prb_data", "tags",
                  "pos", "unit_normal_vector", "batch", "ptr")


class MeshData:
    """One mesh problem (or a disjoint union of several) in the reader's tensor schema."""

    def __init__(self, **fields):
        for k in _TENSOR_FIELDS:
            setattr(self, k, None)
        for k, v in fields.items():
            setattr(self, k, v)

    # -- PyG-like surface -------------------------------------------------
    @property
    def num_nodes(self) -> int:
        return int(self.x.shape[0])

    @property
    def num_edges(self) -> int:
        return int(self.edge_index.shape[1])

    @property
    def is_mixed(self) -> bool:
        return self.tags is not None and self.tags.dim() == 2 and self.tags.shape[1] == 3

    def keys(self):
        return [k for k, v in self.__dict__.items() if torch.is_tensor(v)]

    def to(self, device, non_blocking: bool = False) -> "MeshData":
        out = MeshData()
        for k, v in self.__dict__.items():
            setattr(out, k, v.to(device, non_blocking=non_blocking) if torch.is_tensor(v) else v)
        return out

    def cpu(self) -> "MeshData":
        return self.to("cpu")

    def clone(self) -> "MeshData":
        out = MeshData()
        for k, v in self.__dict__.items():
            setattr(out, k, v.clone() if torch.is_tensor(v) else v)
        return out

    def __repr__(self) -> str:
        parts = [f"{k}={list(v.shape)}" for k, v in self.__dict__.items() if torch.is_tensor(v)]
        return f"MeshData({', '.join(parts)})"


def collate(meshes) -> MeshData:
    """Disjoint union of meshes, as PyG ``Batch.from_data_list`` would build it.

    The reference solves a batch of graphs as ONE union graph (one Broyden problem whose
    norms and stop test are global over the union: ``dirichlet/psignn/test/test_func.py:84-96``).
    Node-level tensors are concatenated, ``edge_index`` is offset by the running node count,
    ``batch`` holds the graph id of every node and ``ptr`` the node offsets.
    """
    meshes = list(meshes)
    out = MeshData()
    offs = [0]
    for m in meshes:
        offs.append(offs[-1] + m.num_nodes)
    # the last four: the DSS schema (dirichlet/dss/utilities/reader.py:61-92; tests/model_dss.py spells the edge field edge_attr_norm)
    for k in ("x", "edge_attr", "a_ij", "y", "sol", "prb_data", "tags", "pos", "unit_normal_vector",
              "a_ij_norm", "edge_attr_norm", "b_prime", "b_prime_norm"):
        vals = [getattr(m, k, None) for m in meshes]
        if all(v is not None for v in vals):
            setattr(out, k, torch.cat(vals, dim=0))
    out.edge_index = torch.cat([m.edge_index + o for m, o in zip(meshes, offs)], dim=1)
    dev = meshes[0].x.device
    out.batch = torch.cat([torch.full((m.num_nodes,), i, dtype=torch.long, device=dev)
                           for i, m in enumerate(meshes)])
    out.ptr = torch.tensor(offs, dtype=torch.long, device=dev)
    out.num_graphs = len(meshes)
    return out
