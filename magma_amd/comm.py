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


class TorchExchange:
    """Default transport: torch.distributed (backend nccl = RCCL on a GPU, gloo on the CPU tests)."""
    name = "torch.distributed"

    def all_reduce(self, t: torch.Tensor):
        return dist.all_reduce(t, op=dist.ReduceOp.SUM, async_op=True)       # work handle (stream-side wait)

    def broadcast(self, t: torch.Tensor, src: int = 0):
        dist.broadcast(t, src=src)

    def close(self):
        pass


class RcclExchange:
    """The same exchange through the C ABI (include/magma_hip.h mg_comm_*: RCCL reached from libmagma_hip.so itself), the
    seam a reference maintainer binds in place of DeepSpeed.  MAGMA_DP_BACKEND=rccl selects it; torch.distributed is then
    used once, as the side channel for the 128-byte unique id.  Collectives are enqueued on the CURRENT stream."""
    name = "mg_comm (RCCL via the C ABI)"

    def __init__(self, device):
        import ctypes as C
        from . import lib as L
        self._L, self._C = L, C
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        torch.cuda.set_device(device)          # before ANY collective: the id broadcast below may itself use the GPU (nccl backend)
        ident = [None]
        if self.rank == 0:
            buf = (C.c_uint8 * 128)()
            L.check(L.load().mg_comm_unique_id(buf), "mg_comm_unique_id")
            ident[0] = bytes(buf)
        dist.broadcast_object_list(ident, src=0)
        h = C.c_void_p()
        idbuf = (C.c_uint8 * 128).from_buffer_copy(ident[0])
        L.check(L.load().mg_comm_init(C.byref(h), idbuf, self.rank, self.world), "mg_comm_init")
        self._h = h

    @staticmethod
    def _dtype(t):
        if t.dtype == torch.float32:
            return 0
        if t.dtype == torch.bfloat16:
            return 1
        raise TypeError(f"mg_comm exchanges fp32 / bf16 gradients, not {t.dtype}")

    def all_reduce(self, t: torch.Tensor):
        assert t.is_cuda and t.is_contiguous()
        self._L.check(self._L.load().mg_comm_allreduce_sum(self._h, t.data_ptr(), t.numel(), self._dtype(t),
                                                           torch.cuda.current_stream().cuda_stream), "mg_comm_allreduce_sum")
        return None                                   # ordered by the stream it was enqueued on

    def broadcast(self, t: torch.Tensor, src: int = 0):
        assert t.is_cuda
        buf = t if t.is_contiguous() else t.contiguous()      # a strided (e.g. transposed-view) parameter goes through a staging copy
        self._L.check(self._L.load().mg_comm_broadcast(self._h, buf.data_ptr(), buf.numel() * buf.element_size(), 2, src,
                                                       torch.cuda.current_stream().cuda_stream), "mg_comm_broadcast")
        if buf is not t:
            t.copy_(buf)

    def close(self):
        if getattr(self, "_h", None):
            self._L.load().mg_comm_destroy(self._h)
            self._h = None

    def __del__(self):       # the engine calls close(); this is the net under an engine that is dropped without it
        try:
            self.close()
        except Exception:  # noqa: BLE001  (interpreter shutdown)
            pass


def make_exchange(device):
    import os
    kind = os.environ.get("MAGMA_DP_BACKEND", "torch")
    if kind == "rccl":
        if device.type != "cuda":
            raise ValueError("MAGMA_DP_BACKEND=rccl needs a GPU")
        return RcclExchange(device)
    if kind != "torch":
        raise ValueError(f"MAGMA_DP_BACKEND must be 'torch' or 'rccl', got {kind!r}")
    return TorchExchange()


def allreduce_grads(flat: List[torch.Tensor], async_op: bool = False, exchange=None, always: bool = False):
    """In-place SUM all-reduce of every tensor in ``flat`` (bucketed).  Returns the
    list of work handles when async_op, else waits."""
    if not dist.is_initialized() or (dist.get_world_size() == 1 and not always):
        return []
    ex = exchange or TorchExchange()
    works = []
    for t in flat:
        n = t.numel()
        for s in range(0, n, BUCKET_ELEMS):
            w = ex.all_reduce(t[s:min(n, s + BUCKET_ELEMS)])
            if w is not None:
                works.append(w)
    if async_op:
        return works
    for w in works:
        w.wait()
    return []


class ReservedCUStream:
    """A compute stream that never dispatches to ``reserve`` of the device's CUs (mg_stream_create_cu_mask), as a
    torch.cuda.ExternalStream.  The training engine runs forward / backward / step on it when MAGMA_DP_RESERVE_CUS=k, so the
    RCCL kernels of the gradient buckets -- enqueued on the exchange stream while the tile GEMMs occupy every CU -- always find
    k CUs to start on.  Off by default: the first multi-GPU hardware run can A/B it."""

    def __init__(self, device: torch.device, reserve: int):
        import ctypes as C
        from . import lib as L
        self.reserve, self._h = int(reserve), C.c_void_p()
        with torch.cuda.device(device):
            L.check(L.load().mg_stream_create_cu_mask(C.byref(self._h), self.reserve), "mg_stream_create_cu_mask")
        self.stream = torch.cuda.ExternalStream(self._h.value, device=device)

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            from . import lib as L
            self.stream.synchronize()
            L.load().mg_stream_destroy(self._h)
            self._h.value = None
