"""Rank helpers, tokenizer, label building, param groups, checkpoint paths --
the parts of reference magma/utils.py the hot path touches.  wandb / gdown /
DeepSpeed argument plumbing is out of scope."""
from __future__ import annotations

import argparse
import os
from collections import defaultdict
from pathlib import Path

import torch
import torch.distributed as dist


def is_main():
    return (not dist.is_initialized()) or dist.get_rank() == 0


def print_main(*msg):
    if is_main():
        print(*msg, flush=True)


def reduce_losses(losses):
    """Mean of a loss tensor over all ranks (reference utils.py:26-34): SUM
    all-reduce (RCCL over xGMI when the backend is nccl) then / world size."""
    if dist.is_initialized():
        losses = losses.detach().clone()
        dist.all_reduce(losses, dist.ReduceOp.SUM)
        return losses / dist.get_world_size()
    return losses


def cycle(loader):
    while True:
        for data in loader:
            yield data


def get_tokenizer(name="gpt2", sequence_length=2048):
    from .tokenizer import get_tokenizer as _gt
    return _gt(name, sequence_length)


def parse_args(argv=None):
    """reference utils.py:61-76 minus deepspeed.add_config_arguments."""
    parser = argparse.ArgumentParser()
    parser.add_argument("--config", type=str, required=False, help="path to your training config")
    parser.add_argument("--local_rank", type=int, default=int(os.environ.get("LOCAL_RANK", -1)),
                        help="local rank passed from distributed launcher")
    parser.add_argument("--synthetic_steps", type=int, default=None,
                        help="override train_steps (synthetic-data smoke runs)")
    parser.add_argument("--micro_batch", type=int, default=None,
                        help="per-GPU micro-batch (default: batch_size / (grad-accum * world), as DeepSpeed derives it)")
    parser.add_argument("--grad_accum", type=int, default=None, help="override gradient_accumulation_steps")
    args, _ = parser.parse_known_args(argv)
    args.deepspeed = False
    return args


def get_world_info():
    return int(os.environ.get("LOCAL_RANK", 0)), int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))


def init_distributed(backend=None):
    """One process per GPU; backend 'nccl' IS RCCL on ROCm (reference
    utils.py:262-269 called deepspeed.init_distributed)."""
    local_rank, rank, world_size = get_world_info()
    if world_size > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        dist.init_process_group(backend=backend, rank=rank, world_size=world_size)
    if torch.cuda.is_available():
        torch.cuda.set_device(local_rank)
    return local_rank, rank, world_size


def build_labels(input_embeddings, captions, eos_token, device=None):
    """Same signature as reference utils.py:334-339; the arithmetic runs in the
    HIP kernel mg_build_labels_i64 (one launch, no per-token host syncs)."""
    from . import ops
    return ops.build_labels(captions.contiguous(), int(input_embeddings.shape[1]), int(eos_token))


def count_parameters(model):
    return sum(p.numel() for p in model.parameters() if p.requires_grad)


def get_params_for_weight_decay_optimization(module, config):
    """Two groups: decayed weights / undecayed (LayerNorm, Embedding, biases, or
    everything when weight_decay == 0) -- semantics of reference utils.py:120-161."""
    decay, no_decay = {"params": []}, {"params": [], "weight_decay": 0.0}
    skip_types = (torch.nn.LayerNorm, torch.nn.Embedding)
    for m in module.modules():
        own = [(n, p) for n, p in m._parameters.items() if p is not None and p.requires_grad]
        if isinstance(m, skip_types) or config.weight_decay == 0.0:
            no_decay["params"].extend(p for _, p in own)
        else:
            for n, p in own:
                (no_decay if n == "bias" else decay)["params"].append(p)
    n_named = sum(1 for _, p in module.named_parameters() if p.requires_grad)
    assert len(decay["params"]) + len(no_decay["params"]) == n_named, \
        "Number of params in both groups != total number of trainable params"
    return [no_decay] if config.weight_decay == 0.0 else [decay, no_decay]


def configure_param_groups(model, config):
    """Groups keyed by (lr, weight_decay): image encoder at ``image_enc_lr``,
    everything else at ``lr`` (reference utils.py:164-238); also patches the
    per-group warm-up min/max LR lists into the scheduler block."""
    if config.image_enc_lr is not None:
        enc = get_params_for_weight_decay_optimization(model.image_prefix.enc, config)
        for g in enc:
            g["lr"] = config.image_enc_lr
        proj = get_params_for_weight_decay_optimization(model.image_prefix.proj, config)
        if config.use_image_embed_layernorm:
            proj += get_params_for_weight_decay_optimization(model.image_prefix.ln, config)
        lm = get_params_for_weight_decay_optimization(model.lm, config)
        groups = [g for g in enc + lm + proj if g["params"]]
    else:
        groups = get_params_for_weight_decay_optimization(model, config)
    merged = defaultdict(dict)
    for g in groups:
        lr, wd = g.get("lr"), g.get("weight_decay")
        slot = merged[f"lr_{lr}_wd_{wd}"]
        slot.setdefault("params", []).extend(g["params"])
        if lr is not None:
            slot["lr"] = lr
        if wd is not None:
            slot["weight_decay"] = wd
    groups = list(merged.values())
    n_trainable = sum(1 for _, p in model.named_parameters() if p.requires_grad)
    n_grouped = sum(len(g["params"]) for g in groups)
    assert n_grouped == n_trainable, f"Some parameters are missing from param groups ({n_grouped} | {n_trainable})"
    sched = config.deepspeed_config_params["scheduler"]["params"]
    sched["warmup_min_lr"] = [config.min_lr for _ in groups]
    sched["warmup_max_lr"] = [g.get("lr", config.lr) for g in groups]
    return groups


def infer_checkpoint_path_from_config(config):
    """<save>/latest -> <save>/<tag>/mp_rank_00_model_states.pt (reference utils.py:285-308)."""
    folder = config.save
    if folder is None:
        raise ValueError("No checkpoint folder specified in config. Please provide a checkpoint.")
    latest = Path(folder) / "latest"
    if not latest.exists():
        raise ValueError(f"No checkpoint found in {folder}. Please provide a checkpoint.")
    path = Path(folder) / latest.read_text().strip() / "mp_rank_00_model_states.pt"
    if not path.exists():
        raise ValueError(f"No checkpoint found in {path}. Please provide a checkpoint.")
    return str(path)


def save_model(model_engine, save_dir, global_step, config=None):
    os.makedirs(save_dir, exist_ok=True)
    cfg = config.to_dict() if config is not None else None
    if cfg is not None and is_main():
        import yaml
        with open(str(Path(save_dir) / "config.yml"), "w") as f:
            yaml.dump(cfg, f, default_flow_style=False)
    model_engine.save_checkpoint(save_dir, client_state={"global_step": global_step, "config": cfg})


def load_model(model_engine, load_dir, load_optimizer_states=True, load_lr_scheduler_states=True):
    """Returns the global step to resume from, 0 when nothing could be loaded
    (reference utils.py:99-117)."""
    try:
        load_path, sd = model_engine.load_checkpoint(
            load_dir, load_optimizer_states=load_optimizer_states,
            load_lr_scheduler_states=load_lr_scheduler_states)
    except (AssertionError, FileNotFoundError) as e:
        load_path, sd = None, None
        print(e)
    if load_path is None:
        print("Model loading failed - starting from global step 0")
        return 0
    return sd["global_step"]
