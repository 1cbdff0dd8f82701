"""Data parallelism for the flat gradient buffer: one process per GPU, RCCL (torch.distributed
backend "nccl" on ROCm) over xGMI; replaces the reference's single-process ``nn.DataParallel``
(``src/02_train.py:109``).  Semantics kept from DataParallel: the loss is a mean over the GLOBAL
batch, so per-rank gradients (means over the local shard) are summed and divided by the world
size; BatchNorm statistics stay per replica.

The gradient buffer is one contiguous fp32 tensor in state_dict order; buckets are contiguous
slices of it, issued last-layers-first (the order in which backward finishes them) as async
all-reduces on the communication stream so that they overlap with the rest of backward.
"""
from __future__ import annotations

from typing import List, Optional

import torch
import torch.distributed as dist


class FlatAllReduce:
    def __init__(self, flat: torch.Tensor, process_group=None, bucket_mb: float = 32.0, bounds=None):
        self.flat = flat
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if (dist.is_available() and dist.is_initialized()) else 1
        n = flat.numel()
        if bounds is not None:
            # the engine's gradient buckets, in the order backward completes them (engine.buckets)
            self.bounds = [(int(a), int(b)) for a, b in bounds]
            assert sorted(self.bounds) == sorted(set(self.bounds)) and sum(b - a for a, b in self.bounds) == n
        else:
            nb = max(1, int(round(n * flat.element_size() / (bucket_mb * 2 ** 20))))
            edges = [int(round(i * n / nb)) for i in range(nb + 1)]
            # reversed: the tail of the buffer (last layers) is ready first in backward
            self.bounds = [(edges[i], edges[i + 1]) for i in reversed(range(nb))]
        self.buckets: List[torch.Tensor] = [flat[a:b] for a, b in self.bounds]
        self._works: List = []

    def launch(self, upto: Optional[int] = None, force: bool = False):
        """Start async all-reduces for buckets [len(started), upto)."""
        if self.world == 1 and not force:
            return
        upto = len(self.buckets) if upto is None else upto
        for b in self.buckets[len(self._works):upto]:
            self._works.append(dist.all_reduce(b, op=dist.ReduceOp.SUM, group=self.pg, async_op=True))

    def wait(self):
        for w in self._works:
            w.wait()
        self._works = []

    def reduce_bucket(self, i: int, force: bool = False):
        """All-reduce bucket i now and make the CURRENT stream wait for it (stream-ordered for RCCL)."""
        if self.world == 1 and not force:
            return
        dist.all_reduce(self.buckets[i], op=dist.ReduceOp.SUM, group=self.pg, async_op=True).wait()

    def all_reduce(self):
        self.launch()
        self.wait()

    @property
    def grad_scale(self) -> float:
        """Factor the optimiser applies to the summed gradients (mean over the global batch)."""
        return 1.0 / self.world
