"""world_size-2 data-parallel logic on CPU (gloo): the bucketed gradient
all-reduce and the loss reduction give every rank the same mean, and averaging
the gradients of two half batches equals the full-batch gradient."""
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from magma_amd import comm
    from magma_amd.utils import reduce_losses
    comm.BUCKET_ELEMS = 1000                     # force several buckets
    torch.manual_seed(0)
    w = torch.randn(37, 11)
    x = torch.randn(8, 11)
    xs = x[rank * 4:(rank + 1) * 4]              # this rank's shard of the global batch
    w_r = w.clone().requires_grad_(True)
    loss = (xs @ w_r.t()).pow(2).mean()          # per-rank mean loss (DeepSpeed semantics: mean of means)
    loss.backward()
    flat = [w_r.grad.reshape(-1).clone(), torch.full((2500,), float(rank + 1))]
    comm.allreduce_grads(flat)
    grad_mean = flat[0] / world
    w_f = w.clone().requires_grad_(True)
    (x @ w_f.t()).pow(2).mean().backward()
    ok = torch.allclose(grad_mean, w_f.grad.reshape(-1), atol=1e-6)
    ok &= bool((flat[1] == 3.0).all())
    red = reduce_losses(loss.detach())
    q.put((rank, bool(ok), float(red)))
    dist.destroy_process_group()


def test_dp_allreduce_two_ranks():
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(60)
    assert all(r[1] for r in res), res
    assert abs(res[0][2] - res[1][2]) < 1e-7          # every rank logs the same reduced loss


def test_lr_schedule_shape():
    from magma_amd.train_engine import LRScheduler
    s = LRScheduler({"warmup_min_lr": [0.0, 0.0], "warmup_num_steps": 100, "total_num_steps": 1000}, "WarmupDecayLR",
                    [8e-4, 2e-6])
    lr0 = s.get_lr()
    assert lr0 == [0.0, 0.0]
    for _ in range(100):
        s.step()
    assert abs(s.get_lr()[0] - 8e-4) < 1e-12 and abs(s.get_lr()[1] - 2e-6) < 1e-15
    for _ in range(450):
        s.step()
    assert abs(s.get_lr()[0] - 4e-4) < 1e-9          # half way down the linear decay


def _run_bench(args, env_extra, timeout=180):
    import json
    import subprocess
    import sys
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(env_extra)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=timeout)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    return r.returncode, (json.loads(lines[-1]) if lines else None), r.stderr


def test_bench_gpus_n_launches_n_ranks_by_itself():
    """`python bench.py --gpus 2` with no launcher around it starts 2 ranks (torch.distributed.run on 127.0.0.1), which join
    one process group and run the bench's barrier / MAX-over-ranks collectives; rank 0 prints ONE line with n_gpus = 2.
    (--rendezvous-only: the launch path without the model -- the model needs a GPU; the same path with the model is
    tests/test_dp_engine_gpu.py::test_bench_two_ranks_self_launched.)"""
    rc, line, err = _run_bench(["--gpus", "2", "--rendezvous-only"], {"MAGMA_BENCH_BACKEND": "gloo"})
    assert rc == 0, err[-2000:]
    assert line["n_gpus"] == 2 and line["launched_by"] == "self" and line["backend"] == "gloo"


def test_bench_gpus_8_launches_the_node_sized_job():
    """The driver's scaling run: `python bench.py --gpus 8` -- 8 ranks (one per GPU of a node, reference README.md:121) join,
    run the barrier / MAX-over-ranks path and rank 0 alone prints the line, n_gpus = 8, with each rank's time in it."""
    rc, line, err = _run_bench(["--gpus", "8", "--rendezvous-only"], {"MAGMA_BENCH_BACKEND": "gloo"})
    assert rc == 0, err[-2000:]
    assert line["n_gpus"] == 8 and line["launched_by"] == "self" and line["backend"] == "gloo"
    assert len(line["per_rank_ms"]) == 8 and abs(line["ms_per_step"] - max(line["per_rank_ms"])) < 1e-3


def test_multi_rank_line_carries_cpu_baseline_and_the_flat_training_keys():
    """The first multi-GPU record must be self-sufficient: at N > 1 rank 0 still measures cpu_baseline -- alone, after the last
    collective of the timed part, the other ranks parked at the final barrier -- and the line carries the keys a scaling curve is
    computed from at its top level.  (--rendezvous-only: that control flow without the model; the same tail of main() with the
    model runs in tests/test_dp_engine_gpu.py::test_bench_two_ranks_self_launched and tools/gpu_bench_2rank_rehearsal.sh.)"""
    rc, line, err = _run_bench(["--gpus", "2", "--rendezvous-only", "--with-cpu-baseline", "--cpu-seconds", "2", "--gen", "2"],
                               {"MAGMA_BENCH_BACKEND": "gloo"}, timeout=600)
    assert rc == 0, err[-2000:]
    assert line["n_gpus"] == 2
    cb = line["cpu_baseline"]
    assert cb["value"] > 0 and cb["unit"] == "tokens/s" and cb["kind"] == "port" and cb["cores"] >= 1 and "sample" in cb
    for k in ("train_images_per_s", "train_ms_per_step", "train_ranks", "train_per_gpu_batch", "train_per_rank_step_ms",
              "train_exposed_comm_ms", "train_comm_stream_busy_ms"):
        assert k in line, k
    assert line["train_ranks"] == 2


def test_bench_refuses_a_mislabelled_launch():
    """--gpus must equal the number of ranks: a 1-rank run asked for 2 GPUs exits non-zero instead of printing n_gpus = 1."""
    rc, line, err = _run_bench(["--gpus", "2", "--rendezvous-only"], {"WORLD_SIZE": "1", "RANK": "0"})
    assert rc != 0 and line is None and "--gpus 2 but WORLD_SIZE=1" in err
