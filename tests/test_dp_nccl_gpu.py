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


def test_overlapped_allreduce_single_rank(dev):
    g0, m0 = _grads(dev, False)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29541")
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
