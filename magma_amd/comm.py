"""Data-parallel exchange step: one gradient all-reduce per optimizer step over
the flat fp32 gradient blocks (383.8 M elements for MAGMA_v1), through
torch.distributed -- backend "nccl" IS RCCL on ROCm, over xGMI inside a node;
"gloo" on CPU for the world_size-2 tests.  Large blocks are cut into buckets of at most
BUCKET_ELEMS so that the first one can leave while the later ones are still being cast; on ONE
process group the buckets are issued to one RCCL stream and run back to back (they do not
overlap each other -- what overlaps is the exchange with the backward pass, see
MagmaEngine._reduce_params_async).  Driving several xGMI links at once is RCCL's own business
(its channels / rings inside one all-reduce), not this file's.  Semantics preserved from
DeepSpeed ZeRO-2 [UNVENDORED]: gradients are SUMMED here and divided by world size inside the
fused AdamW kernel (mean)."""
from __future__ import annotations

from typing import List

import torch
import torch.distributed as dist

BUCKET_ELEMS = 32 * 1024 * 1024      # 128 MiB of fp32 per bucket


def allreduce_grads(flat: List[torch.Tensor], async_op: bool = False):
    """In-place SUM all-reduce of every tensor in ``flat`` (bucketed).  Returns the
    list of work handles when async_op, else waits."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return []
    works = []
    for t in flat:
        n = t.numel()
        for s in range(0, n, BUCKET_ELEMS):
            works.append(dist.all_reduce(t[s:min(n, s + BUCKET_ELEMS)], op=dist.ReduceOp.SUM, async_op=True))
    if async_op:
        return works
    for w in works:
        w.wait()
    return []
