"""MultimodalConfig for the MI355X MAGMA path.

Field names, defaults and the YAML lookup rules follow the reference's
dataclass (reference magma/config.py:20-94, lookup :10-17) so that a user's
existing MAGMA_v1.yml / MAGMA_v2.yml parse unchanged.  What differs:
  * YAML keys that are not fields (MAGMA_v2.yml carries ``dataset_type``,
    ``vqa_dir``, ``gqa_dir``) land in ``extra`` instead of raising;
  * ``deepspeed_config_params`` keeps the reference's shape (train.py and
    configure_param_groups read/patch its scheduler block) but is consumed by
    our RCCL data-parallel engine; bf16 replaces fp16 so the loss-scale entry
    is inert.
"""
from __future__ import annotations

import dataclasses
import uuid
from pathlib import Path
from pprint import pprint
from typing import Any

import yaml

_REQUIRED = dataclasses.MISSING

# (name, type, default) -- grouped as in the reference's sections
_SPEC = [
    # training
    ("batch_size", int, _REQUIRED), ("train_steps", int, _REQUIRED),
    ("optimizer_name", str, "AdamW"), ("lr", float, 8.0e-4), ("image_enc_lr", float, None),
    ("min_lr", float, 0.0), ("lr_decay_iters", int, None), ("gradient_accumulation_steps", int, 1),
    ("image_size", int, 256), ("eval_every", int, 250), ("eval_steps", int, 25), ("zero_stage", int, 2),
    ("gradient_clipping", float, 1.0), ("warmup_num_steps", int, 100), ("weight_decay", float, 0.0),
    ("run_blind", bool, False), ("fine_tune", bool, False), ("load_optimizer", bool, True),
    # checkpointing
    ("save_every", int, 2500), ("save", str, None), ("load", str, None),
    # data
    ("train_dataset_name", str, "conceptual_captions"), ("eval_dataset_name", str, "/data/conceptual_captions"),
    ("train_dataset_dir", Any, "/data/coco_data"), ("eval_dataset_dir", Any, "/data/coco_data"),
    ("eval_dataset_pct", float, 0.1),
    # architecture
    ("encoder_name", str, "clip"), ("tokenizer_name", str, "gpt2"), ("lm_name", str, "EleutherAI/gpt-j-6B"),
    ("image_seq_len", int, 2), ("pretrained_img_encoder", bool, False), ("seq_len", int, None),
    # freezing / prefix
    ("freeze_lm", bool, True), ("freeze_img_encoder", bool, True),
    ("image_embed_dropout_prob", float, 0.0), ("use_image_embed_layernorm", bool, False),
    # adapters, classification, logging
    ("adapter_config", dict, None), ("class_dict", dict, None),
    ("name", str, None), ("log_every", int, 1), ("wandb_project", str, "magma"),
]


def load_config(path, config_dir=Path("configs")):
    """'<name>' -> '<name>.yml' -> cwd, then ``config_dir``, then this checkout's configs/."""
    p = str(path)
    if not p.endswith(".yml"):
        p += ".yml"
    candidates = [Path(p), Path(config_dir) / p, Path(__file__).resolve().parent.parent / "configs" / Path(p).name]
    for c in candidates:
        if c.exists():
            with open(c, "r") as fh:
                return yaml.safe_load(fh)
    raise FileNotFoundError(f"config {path!r} not found (tried {[str(c) for c in candidates]})")


class _ConfigMethods:
    def print(self):
        from .utils import is_main
        if is_main():
            print("-" * 100)
            pprint(self.__dict__, indent=4)
            print("-" * 100)

    def __post_init__(self):
        self.is_classifier = self.class_dict is not None
        self.adapter_config = self.adapter_config or {}
        sched = {"warmup_min_lr": self.min_lr, "warmup_max_lr": self.lr, "warmup_num_steps": self.warmup_num_steps}
        if self.lr_decay_iters is None:
            self.lr_scheduler = "WarmupLR"
        else:
            self.lr_scheduler = "WarmupDecayLR"
            sched = {"total_num_steps": self.lr_decay_iters, **sched}
        self.scheduler_dict = {"type": self.lr_scheduler, "params": sched}
        self.deepspeed_config_params = {
            "train_batch_size": self.batch_size,
            "gradient_accumulation_steps": self.gradient_accumulation_steps,
            "gradient_clipping": self.gradient_clipping,
            "fp16": {"enabled": True, "loss_scale_window": 250},
            "scheduler": self.scheduler_dict,
            "zero_optimization": {"stage": self.zero_stage, "load_from_fp32_weights": False},
        }
        if self.name is None:
            self.name = str(uuid.uuid4())[:8]

    @classmethod
    def from_yml(cls, path):
        raw = load_config(path)
        names = {s[0] for s in _SPEC}
        return cls(**{k: v for k, v in raw.items() if k in names},
                   extra={k: v for k, v in raw.items() if k not in names})

    def to_dict(self):
        return dataclasses.asdict(self)


MultimodalConfig = dataclasses.make_dataclass(
    "MultimodalConfig",
    [(n, t) if d is _REQUIRED else (n, t, dataclasses.field(default=d)) for n, t, d in _SPEC]
    + [("extra", dict, dataclasses.field(default_factory=dict))],
    bases=(_ConfigMethods,),
)
MultimodalConfig.__module__ = __name__
MultimodalConfig.__doc__ = "All knobs of a MAGMA run (see module docstring)."
