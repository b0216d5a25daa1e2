"""RCCL path on the GPU box: a 1-rank process group (backend nccl == RCCL on ROCm) drives the
bucketed, backward-overlapped gradient all-reduce; results must equal the run without a group."""
import os

import pytest
import torch
import torch.distributed as dist

pytestmark = pytest.mark.gpu


def _grads(dev, with_group):
    from magma_amd.testing import build_reduced_magma
    from magma_amd.train_engine import MagmaEngine
    torch.manual_seed(0)
    model = build_reduced_magma(dev, n_positions=128)
    model.config.gradient_accumulation_steps = 2
    eng = MagmaEngine(model)
    assert eng._dist == with_group
    if with_group:
        assert eng._exchange.name.startswith("mg_comm") == (os.environ.get("MAGMA_DP_BACKEND") == "rccl")
    eng.train()
    g = torch.Generator().manual_seed(1)
    out = []
    for micro in range(2):
        images = torch.randn(2, 3, 64, 64, generator=g).to(dev)
        caps = torch.full((2, 128), model.eos_token, dtype=torch.int64)
        caps[:, :15] = torch.randint(0, 1000, (2, 15), generator=g)
        mask = (torch.rand(2, 4, 512, generator=g) < 0.9).float() / 0.9
        o = eng(images, caps.to(dev), dropout_mask=mask.to(dev))
        eng.backward(o.loss)
        if micro == 1 and with_group:
            assert any(eng._reduced), "no bucket was handed to RCCL during backward"
        if micro == 1:
            out = [grp.grad.clone() for grp in eng.groups]
        eng.step()
    return out, [grp.master.clone() for grp in eng.groups]


@pytest.mark.parametrize("backend", ["torch", "rccl"])
def test_overlapped_allreduce_single_rank(dev, backend, monkeypatch):
    """backend "rccl": the same exchange through the C ABI (mg_comm_unique_id / _init / _allreduce_sum / _destroy over the RCCL
    library, include/magma_hip.h) instead of torch.distributed's collectives."""
    g0, m0 = _grads(dev, False)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29541")
    monkeypatch.setenv("MAGMA_DP_BACKEND", backend)
    dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        g1, m1 = _grads(dev, True)
    finally:
        dist.destroy_process_group()
    # SUM over one rank is the identity; the overlap must not corrupt anything.  Not bitwise: the
    # column-sum kernels accumulate with fp32 atomics, whose order differs from run to run.
    for a, b in zip(g0, g1):
        assert float((a - b).norm() / (a.norm() + 1e-20)) < 1e-5
    for a, b in zip(m0, m1):
        assert float((a - b).norm() / (a.norm() + 1e-20)) < 1e-5


def test_compute_stream_with_reserved_cus(dev, monkeypatch):
    """MAGMA_DP_RESERVE_CUS=8: forward / backward / step run on a stream whose CU mask leaves 8 CUs to the exchange
    (mg_stream_create_cu_mask, hipExtStreamCreateWithCUMask); gradients and masters equal the run without a group, the
    engine's kernels really went to that stream, and timing the exchange stream (bench.py: comm_stream_busy_ms) works."""
    g0, m0 = _grads(dev, False)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29541")
    monkeypatch.setenv("MAGMA_DP_RESERVE_CUS", "8")
    dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        from magma_amd.train_engine import MagmaEngine
        seen = {}
        orig = MagmaEngine._forward_impl

        def spy(self, *a, **k):
            seen["stream"] = torch.cuda.current_stream(self.device).cuda_stream
            seen["masked"] = self._compute_stream.stream.cuda_stream
            self.time_comm = True
            return orig(self, *a, **k)
        monkeypatch.setattr(MagmaEngine, "_forward_impl", spy)
        g1, m1 = _grads(dev, True)
    finally:
        dist.destroy_process_group()
    assert seen["stream"] == seen["masked"] != torch.cuda.default_stream(dev).cuda_stream
    for a, b in zip(g0, g1):
        assert float((a - b).norm() / (a.norm() + 1e-20)) < 1e-5
    for a, b in zip(m0, m1):
        assert float((a - b).norm() / (a.norm() + 1e-20)) < 1e-5


def test_mg_comm_collectives_single_rank(dev):
    """The comm entry points on their own: a 1-rank communicator (SUM over one rank = identity, broadcast from rank 0 = identity),
    fp32 and bf16, then destroy; a bad dtype / root is refused with MG_ERR_SHAPE."""
    import ctypes as C
    from magma_amd import lib as L
    dll = L.load()
    ident = (C.c_uint8 * 128)()
    L.check(dll.mg_comm_unique_id(ident), "mg_comm_unique_id")
    h = C.c_void_p()
    torch.cuda.set_device(dev)
    L.check(dll.mg_comm_init(C.byref(h), ident, 0, 1), "mg_comm_init")
    try:
        s = torch.cuda.current_stream().cuda_stream
        for dt, code in ((torch.float32, 0), (torch.bfloat16, 1)):
            x = torch.randn(1 << 16, device=dev).to(dt)
            ref = x.clone()
            L.check(dll.mg_comm_allreduce_sum(h, x.data_ptr(), x.numel(), code, s), "mg_comm_allreduce_sum")
            L.check(dll.mg_comm_broadcast(h, x.data_ptr(), x.numel() * x.element_size(), 2, 0, s), "mg_comm_broadcast")
            torch.cuda.synchronize()
            assert torch.equal(x, ref)
        assert dll.mg_comm_allreduce_sum(h, x.data_ptr(), x.numel(), 7, s) == -1
        assert dll.mg_comm_broadcast(h, x.data_ptr(), 16, 2, 3, s) == -1
    finally:
        L.check(dll.mg_comm_destroy(h), "mg_comm_destroy")
