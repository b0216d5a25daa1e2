"""Optimizer-step semantics of the reference's DeepSpeed configuration (reference train.py:96-101,
config.py:113-134: AdamW(betas=(0.9, 0.95), eps 1e-8, weight_decay from the config), gradient clipping 1.0 on the
global norm, WarmupDecayLR), checked as a multi-step TRAJECTORY on every trainable tensor: the fused
mg_sumsq_f32 + mg_adamw_f32 kernels over the engine's flat fp32 state against torch.optim.AdamW +
torch.nn.utils.clip_grad_norm_ + the schedule restated here from DeepSpeed's documented WarmupDecayLR formula
(DeepSpeed is un-vendored: this pins the arithmetic to PyTorch's own optimizer, not to DeepSpeed's source)."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


def warmup_decay_lr(step, lr_max, lr_min, warmup, total):
    """DeepSpeed WarmupDecayLR (warmup_type "log"): lr_min + (lr_max - lr_min) * log(step + 1) / log(warmup) while
    step < warmup, then lr_max * (total - step) / (total - warmup), floored at 0."""
    if step < warmup:
        return lr_min + (lr_max - lr_min) * math.log(step + 1) / math.log(warmup)
    return lr_max * max(0.0, (total - step) / max(1.0, total - warmup))


@pytest.mark.parametrize("wd", [0.0, 0.05])
def test_five_step_trajectory_matches_torch_adamw(dev, wd):
    from magma_amd.testing import tiny_multimodal_config
    from magma_amd.train_engine import MagmaEngine
    from magma_amd.magma import Magma
    from magma_amd.image_encoders import ModifiedResNetTrunk
    from magma_amd.language_model import GPTJConfig
    torch.manual_seed(0)
    cfg = tiny_multimodal_config(weight_decay=wd, lr=8e-4, min_lr=1e-5, image_enc_lr=2e-5, lr_decay_iters=12,
                                 warmup_num_steps=4)
    lm_cfg = GPTJConfig(vocab_size=1056, hidden_size=512, num_layers=2, num_heads=2, rotary_dim=64, intermediate_size=2048,
                        max_position_embeddings=128)
    enc = ModifiedResNetTrunk((1, 1, 2, 1), 16, 64, device=dev, dtype=torch.bfloat16)
    model = Magma(cfg, device=dev, lm_config=lm_cfg, enc=enc)
    model.config.gradient_accumulation_steps = 1
    eng = MagmaEngine(model)
    sched = cfg.deepspeed_config_params["scheduler"]["params"]
    warmup, total = eng.lr_scheduler.warmup, eng.lr_scheduler.total
    assert warmup == 4 and total == sched["total_num_steps"]
    # torch side: one fp32 parameter per flat group (AdamW is elementwise; the clip norm is global over all groups)
    tparams = [torch.nn.Parameter(g.master.clone()) for g in eng.groups]
    opt = torch.optim.AdamW([{"params": [p], "lr": g.lr_max, "weight_decay": g.wd} for p, g in zip(tparams, eng.groups)],
                            betas=(0.9, 0.95), eps=1e-8)
    n_train = sum(g.n for g in eng.groups)
    assert n_train >= sum(p.numel() for p in model.parameters() if p.requires_grad)
    gen = torch.Generator(device=dev).manual_seed(3)
    for step in range(7):           # 4 warm-up steps, then 3 on the decay branch
        scale = 10.0 if step % 2 == 0 else 1e-3          # norm far above / far below the clip threshold
        for g, tp in zip(eng.groups, tparams):
            gr = torch.randn(g.n, device=dev, generator=gen) * scale / math.sqrt(n_train)
            g.grad.copy_(gr)
            tp.grad = gr.clone()
        lrs = [warmup_decay_lr(step, g.lr_max, cfg.min_lr, warmup, total) for g in eng.groups]
        assert eng.lr_scheduler.get_lr() == pytest.approx(lrs, rel=1e-12)
        for pg, lr in zip(opt.param_groups, lrs):
            pg["lr"] = lr
        norm = torch.nn.utils.clip_grad_norm_(tparams, 1.0)
        opt.step()
        eng.micro_steps += 1
        eng.step()
        assert eng.grad_norm() == pytest.approx(float(norm), rel=1e-4)
        for g, tp in zip(eng.groups, tparams):
            err = float((g.master - tp.detach()).abs().max() / tp.detach().abs().max())
            assert err < 2e-6, (step, err)
            assert torch.equal(g.model, g.master.to(torch.bfloat16))          # bf16 copy the kernels read
            assert float(g.grad.abs().sum()) == 0.0
    assert eng.global_steps == 7
