"""Helpers shared by tests/, smoke() and bench.py to build reduced-size models
whose structure (head_dim 256, rotary 64, adapters, RN trunk) is identical to
MAGMA_v1/v2.  Nothing here touches the oracle."""
from __future__ import annotations

import torch

from .config import MultimodalConfig
from .image_encoders import ModifiedResNetTrunk
from .language_model import GPTJConfig
from .magma import Magma


def tiny_multimodal_config(mlp_factor=4, attn_factor=None, **kw) -> MultimodalConfig:
    ad = {"mlp": {"adapter_type": "normal", "downsample_factor": mlp_factor}}
    if attn_factor:
        ad["attention"] = {"adapter_type": "normal", "downsample_factor": attn_factor}
    base = dict(batch_size=2, train_steps=1, encoder_name="clip_resnet_large", adapter_config=ad,
                freeze_img_encoder=False, use_image_embed_layernorm=True, image_embed_dropout_prob=0.1,
                image_size=64, image_enc_lr=2.0e-6, lr_decay_iters=1000)
    base.update(kw)
    return MultimodalConfig(**base)


def build_reduced_magma(device, n_layer=2, n_head=2, d_ff=2048, vocab=1056, n_positions=256, enc_width=16,
                        enc_layers=(1, 1, 2, 1), mlp_factor=4, attn_factor=None, resolution=64) -> Magma:
    d = n_head * 256
    lm_cfg = GPTJConfig(vocab_size=vocab, hidden_size=d, num_layers=n_layer, num_heads=n_head, rotary_dim=64,
                        intermediate_size=d_ff, max_position_embeddings=n_positions)
    enc = ModifiedResNetTrunk(enc_layers, enc_width, resolution, device=device, dtype=torch.bfloat16)
    return Magma(tiny_multimodal_config(mlp_factor, attn_factor), device=device, lm_config=lm_cfg, enc=enc)
