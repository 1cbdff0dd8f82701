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
    """bf16_buckets=True: every bucket travels as a bf16 copy (half the bytes over the point-to-point xGMI links,
    where a ring step is per-link bound) -- the local fp32 gradients are pre-scaled by 1/world and rounded to bf16,
    summed by the collective, and written back into the fp32 buffer, so the optimiser still accumulates in fp32
    (``grad_scale`` is then 1).  Per-element error: bf16 rounding of each rank's contribution (2^-9 relative)."""

    def __init__(self, flat: torch.Tensor, process_group=None, bucket_mb: float = 32.0, bounds=None, bf16_buckets: bool = False):
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
        self.bf16 = bool(bf16_buckets)
        self._stage: List[torch.Tensor] = ([torch.empty(b.numel(), dtype=torch.bfloat16, device=flat.device) for b in self.buckets]
                                           if self.bf16 else [])
        self._works: List = []

    def _send(self, i: int):
        """The tensor that goes on the wire for bucket i (stream-ordered behind the bucket's producer)."""
        if not self.bf16:
            return self.buckets[i]
        st = self._stage[i]
        if self.world > 1:
            torch.mul(self.buckets[i], 1.0 / self.world, out=self.buckets[i])   # mean before rounding: no bf16 overflow in the sum
        st.copy_(self.buckets[i])
        return st

    def _recv(self, i: int):
        if self.bf16:
            self.buckets[i].copy_(self._stage[i])

    def launch(self, upto: Optional[int] = None, force: bool = False):
        """Start async all-reduces for buckets [len(started), upto)."""
        if self.world == 1 and not force:
            return
        upto = len(self.buckets) if upto is None else upto
        for i in range(len(self._works), upto):
            self._works.append(dist.all_reduce(self._send(i), op=dist.ReduceOp.SUM, group=self.pg, async_op=True))

    def wait(self):
        for i, w in enumerate(self._works):
            w.wait()
            self._recv(i)
        self._works = []

    def reduce_bucket(self, i: int, force: bool = False):
        """All-reduce bucket i now and make the CURRENT stream wait for it (stream-ordered for RCCL)."""
        if self.world == 1 and not force:
            return
        dist.all_reduce(self._send(i), op=dist.ReduceOp.SUM, group=self.pg, async_op=True).wait()
        self._recv(i)

    def all_reduce(self):
        self.launch()
        self.wait()

    @property
    def grad_scale(self) -> float:
        """Factor the optimiser applies to the summed gradients (mean over the global batch); the bf16 buckets
        carry the mean already."""
        return 1.0 if self.bf16 else 1.0 / self.world
