"""Two ranks driving the REAL MagmaEngine on one GPU (both processes on cuda:0; backend gloo, which moves the CUDA
buckets through the host -- RCCL refuses two ranks on one device): the data-parallel step end to end.

  * every replica starts from rank 0's model although each process seeds its random init differently: frozen
    GPT-J weights, trainable masters, BatchNorm statistics (the broadcast DeepSpeed's initialize() does, reference
    train.py:103-111);
  * buckets are handed to the process group DURING backward (overlap path), as bf16;
  * mean over ranks of the half-batch gradients == gradient of the full batch on one process (equal label counts per
    rank, so mean-of-means == global mean; reference train_loop.py:7-21 + ZeRO-2 averaging);
  * identical fp32 masters on both ranks after step(), equal to the single-process full-batch step."""
import os
import socket
import tempfile

import pytest
import torch

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _batch():
    g = torch.Generator().manual_seed(7)
    images = torch.randn(4, 3, 64, 64, generator=g)
    caps = torch.full((4, 128), 1054, dtype=torch.int64)            # eos of the reduced vocabulary
    for b, n in enumerate((17, 9, 9, 17)):                           # 26 label tokens on each rank
        caps[b, :n] = torch.randint(0, 1000, (n,), generator=g)
    mask = (torch.rand(4, 4, 512, generator=g) < 0.9).float() / 0.9
    return images, caps, mask


def _build(dev, seed):
    from magma_amd.testing import build_reduced_magma
    torch.manual_seed(seed)
    model = build_reduced_magma(dev, n_positions=128)
    model.lm.init_weights(seed)                      # frozen LM differs per seed
    model.config.gradient_accumulation_steps = 1
    return model


def _amplify_adapters(model):
    with torch.no_grad():
        for n, p in model.named_parameters():
            if ".adapter." in n:
                p.mul_(20.0)


def _run(eng, images, caps, mask, dev):
    eng.train()
    out = eng(images.to(dev), caps, dropout_mask=mask.to(dev), captions_host=caps)
    eng.backward(out.loss)
    handed = sum(len(r) for r in eng._reduced)     # ranges given to the process group while backward was still running
    return out, handed


def _worker(rank, world, port, outdir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from magma_amd.train_engine import MagmaEngine
        model = _build(dev, seed=100 + rank)         # DIFFERENT random init per rank: the engine must broadcast rank 0's
        if rank == 0:
            _amplify_adapters(model)
        eng = MagmaEngine(model)
        assert eng._dist and eng.world == 2 and eng.exchange_bf16
        frozen = torch.stack([p.detach().float().sum() for n, p in model.named_parameters() if not p.requires_grad])
        bn = torch.stack([b.float().sum() for b in model.buffers() if b.is_floating_point()])
        images, caps, mask = _batch()
        sl = slice(rank * 2, rank * 2 + 2)
        out, handed = _run(eng, images[sl], caps[sl], mask[sl], dev)
        eng.step()                                                    # joins the exchange, then clip + AdamW on the bf16 sums
        grads = [g.comm.float().clone() / world for g in eng.groups]  # exchanged buckets = SUM over ranks -> mean
        torch.save({"frozen": frozen.cpu(), "bn": bn.cpu(), "loss": float(out.loss), "handed": handed,
                    "grads": [g.cpu() for g in grads], "masters": [g.master.cpu() for g in eng.groups]},
                   os.path.join(outdir, f"rank{rank}.pt"))
    finally:
        dist.destroy_process_group()


def test_two_rank_engine_step(dev):
    import torch.multiprocessing as mp
    from magma_amd.train_engine import MagmaEngine
    outdir = tempfile.mkdtemp(prefix="magma_dp_")
    port = _free_port()
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_worker, args=(r, 2, port, outdir)) for r in range(2)]
    for p in procs:
        p.start()
    # single-process reference on the full batch, from the same model rank 0 builds
    model = _build(dev, seed=100)
    _amplify_adapters(model)
    eng = MagmaEngine(model)
    assert not eng._dist
    images, caps, mask = _batch()
    out, _ = _run(eng, images, caps, mask, dev)
    ref_grads = [g.grad.clone().cpu() for g in eng.groups]
    frozen = torch.stack([p.detach().float().sum() for n, p in model.named_parameters() if not p.requires_grad]).cpu()
    eng.step()
    ref_masters = [g.master.cpu() for g in eng.groups]
    for p in procs:
        p.join(300)
        assert p.exitcode == 0, f"rank process failed (exit code {p.exitcode})"
    r0, r1 = (torch.load(os.path.join(outdir, f"rank{r}.pt")) for r in range(2))
    # (1) one model on every rank, the one rank 0 built
    assert torch.equal(r0["frozen"], r1["frozen"]) and torch.equal(r0["bn"], r1["bn"])
    assert torch.allclose(r0["frozen"], frozen, rtol=1e-6, atol=1e-6)
    # (2) overlap path taken, on both ranks
    assert r0["handed"] > 0 and r1["handed"] > 0
    # (3) mean of the half-batch gradients == full-batch gradient (bf16 buckets: ~2^-8 per element)
    for a, b, ref in zip(r0["grads"], r1["grads"], ref_grads):
        assert torch.equal(a, b)
        cos = float((a * ref).sum() / (a.norm() * ref.norm() + 1e-30))
        assert cos > 0.9995 and float((a - ref).norm() / ref.norm()) < 2e-2, (cos, float((a - ref).norm() / ref.norm()))
    assert abs(0.5 * (r0["loss"] + r1["loss"]) - float(out.loss)) < 5e-3 * abs(float(out.loss))
    # (4) identical masters on both ranks, and the same update as the single-process step (first Adam step ~ lr * sign(g):
    #     compare the update direction where the gradient is not tiny)
    for a, b, ref, g in zip(r0["masters"], r1["masters"], ref_masters, ref_grads):
        assert torch.equal(a, b)
        big = g.abs() > 1e-2 * g.abs().max()
        assert float((a[big] - ref[big]).norm() / (ref[big].norm() + 1e-30)) < 1e-3


def test_bench_two_ranks_self_launched(dev):
    """`python bench.py --gpus 2 ...` -- no launcher around it -- is a 2-rank run: bench.py starts the ranks itself (one-GPU
    rehearsal mode: both on device 0, gloo instead of RCCL, which refuses two ranks per device), every rank issues the same
    collectives in the same order, rank 0 prints ONE line with n_gpus = 2 and the data-parallel fields of the training leg.
    The numbers of such a run mean nothing; the control flow is what N real GPUs execute."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MAGMA_BENCH_BACKEND="gloo", MAGMA_BENCH_DEVICE="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--layers", "1",
                        "--train-steps", "1", "--train-warmup", "1", "--train-batch", "2", "--no-cpu-baseline", "--fp8", "off",
                        "--no-train-truncate"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["train_ranks"] == 2 and line["value"] > 0
    dp = line["train"]["data_parallel"]
    assert dp["ranks"] == 2 and dp["rccl_ranks"] == 2 and dp["global_batch"] == 4
    assert dp["exposed_comm_ms_per_step"] is not None and dp["elements_handed_over_during_backward"] > 0
    assert line["train_images_per_s"] > 0
