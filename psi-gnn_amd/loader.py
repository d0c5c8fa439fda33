"""Stand-ins for the three torch_geometric pieces the reference's scripts wrap around the model
(``dirichlet/psignn/main.py:9-10,70-78,106``; ``test/test_func.py:134-140``): ``DataListLoader``, ``DataLoader`` and
``DataParallel``.  No torch_geometric is needed.

* ``DataListLoader(dataset, batch_size, shuffle)`` yields python lists of graphs (as PyG's does);
* ``DataLoader(dataset, batch_size, shuffle)`` yields disjoint-union batches (``data.collate``);
* ``DataParallel(module)`` keeps the reference's calling convention -- ``model(list_of_graphs)``, ``model.module`` -- but
  is one process per GPU by design: the list is collated into ONE union batch on the module's device (what PyG's
  DataParallel does on each of its devices) and handed to the module.  Multi-GPU data parallelism is done with one process
  per GPU and ``training_class.allreduce_mean_grads`` (RCCL), not with replica threads.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from .data.meshdata import collate


class DataListLoader:
    def __init__(self, dataset, batch_size=1, shuffle=False, generator=None, **_ignored):
        self.dataset, self.batch_size, self.shuffle, self.generator = dataset, int(batch_size), bool(shuffle), generator

    def __len__(self):
        return (len(self.dataset) + self.batch_size - 1) // self.batch_size

    def _order(self):
        n = len(self.dataset)
        return torch.randperm(n, generator=self.generator).tolist() if self.shuffle else list(range(n))

    def __iter__(self):
        order = self._order()
        for i in range(0, len(order), self.batch_size):
            yield [self.dataset[j] for j in order[i:i + self.batch_size]]


class DataLoader(DataListLoader):
    def __iter__(self):
        for graphs in super().__iter__():
            yield collate(graphs)


class DataParallel(nn.Module):
    def __init__(self, module, device_ids=None, output_device=None):
        super().__init__()
        self.module = module

    def _device(self):
        return next(self.module.parameters()).device

    def forward(self, data_list):
        if isinstance(data_list, (list, tuple)):
            if len(data_list) == 0:
                raise ValueError("DataParallel received an empty list of graphs")
            batch = data_list[0] if len(data_list) == 1 else collate(list(data_list))
        else:
            batch = data_list
        return self.module(batch.to(self._device()))
