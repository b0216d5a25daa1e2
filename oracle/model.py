"""Oracle: functional restatement of the MAGMA model graph (TEST INFRASTRUCTURE).

Every tensor lives in a flat ``dict[str, Tensor]`` keyed by the reference's own
state-dict names, so the same dict drives the oracle and the
HIP product path:

  lm.transformer.wte.weight                       reference magma/magma.py:52
  lm.transformer.h.{i}.ln_1.{weight,bias}
  lm.transformer.h.{i}.attn.attention.{q,k,v,out}_proj.weight            (v1)
  lm.transformer.h.{i}.attn.attn_block.attention.{q,k,v,out}_proj.weight (v2,
        reference magma/adapters.py:107 wraps the block as ``attn_block``)
  lm.transformer.h.{i}.attn.adapter.{0,2}.{weight,bias}                  (v2)
  lm.transformer.h.{i}.mlp.0.{c_fc,c_proj}.{weight,bias}  (reference
        magma/magma.py:143-148: mlp -> Sequential(mlp, Adapter))
  lm.transformer.h.{i}.mlp.1.adapter.{0,2}.{weight,bias}
  lm.transformer.ln_f.{weight,bias} ; lm.lm_head.{weight,bias}
  image_prefix.enc.*  (openai/CLIP ModifiedResNet names)
  image_prefix.proj.{weight,bias} ; image_prefix.ln.{weight,bias}
                                                  reference magma/image_prefix.py:72,76
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F

Params = Dict[str, torch.Tensor]


@dataclass
class OracleConfig:
    """Shape knobs.  Defaults = MAGMA_v1 (reference magma/language_model.py:12-24,
    configs/MAGMA_v1.yml:3-4, magma/image_prefix.py:11-21)."""

    n_layer: int = 28
    n_head: int = 16
    head_dim: int = 256
    rotary_dim: int = 64
    d_ff: int = 16384
    vocab_in: int = 50258  # wte rows   (resize_token_embeddings(len(tokenizer)))
    vocab_out: int = 50258  # lm_head rows (untied, with bias)
    n_positions: int = 2048
    ln_eps: float = 1e-5
    mlp_adapter_hidden: int = 1024  # 4096 // downsample_factor 4 ; 0 = none
    attn_adapter_hidden: int = 0  # v2: 512 ; 0 = none
    # adapter placement (reference magma/magma.py:102-174): "normal" = after the block's output with its own residual
    # (adapters.py:38-39, 109-116); "parallel" / "scaled_parallel" = beside the block, reading its INPUT
    # (adapters.py:42-92), the latter with the trainable scalar ``adapter_scale``
    mlp_adapter_type: str = "normal"
    attn_adapter_type: str = "normal"
    # adapter options (reference magma/adapters.py:11-24): the bottleneck activation ("relu" = nn.ReLU, the default; "gelu" =
    # nn.GELU(), erf; "gelu_tanh" = nn.GELU(approximate="tanh")) and a LayerNorm(dim) in front of the down-projection
    # (add_layernorm: the Sequential's indices shift -- 0 = LayerNorm, 1 = down, 3 = up -- and with them the state-dict keys)
    adapter_act: str = "relu"
    adapter_layernorm: bool = False
    # image side
    enc_width: int = 96  # RN50x16
    enc_layers: Tuple[int, int, int, int] = (6, 8, 18, 8)
    use_prefix_ln: bool = True
    bn_eps: float = 1e-5
    eos_token: int = 50256
    image_token: int = 50257

    @property
    def d_model(self) -> int:
        return self.n_head * self.head_dim

    @property
    def enc_out_dim(self) -> int:
        return self.enc_width * 32  # 3072 for RN50x16, reference image_prefix.py:20

    @staticmethod
    def magma_v1() -> "OracleConfig":
        return OracleConfig()

    @staticmethod
    def magma_v2() -> "OracleConfig":
        # configs/MAGMA_v2.yml:4 -- mlp AND attention adapters at downsample 8
        return OracleConfig(mlp_adapter_hidden=512, attn_adapter_hidden=512)

    @staticmethod
    def tiny(**kw) -> "OracleConfig":
        """Small but structurally identical config used by the parity tests
        (head_dim and rotary_dim stay at the full-size values the kernels are
        specialised for)."""
        base = dict(
            n_layer=2, n_head=2, head_dim=256, rotary_dim=64, d_ff=2048,
            vocab_in=1056, vocab_out=1056, n_positions=256,
            mlp_adapter_hidden=128, attn_adapter_hidden=0,
            enc_width=16, enc_layers=(1, 1, 2, 1),
            eos_token=1054, image_token=1055,
        )
        base.update(kw)
        return OracleConfig(**base)


# ----------------------------------------------------------------------------
# parameter naming helpers
# ----------------------------------------------------------------------------

def attn_prefix(cfg: OracleConfig, i: int) -> str:
    if cfg.attn_adapter_hidden:
        # AdapterWrapper keeps the block as ``attn_block`` (adapters.py:107), ParallelAdapterWrapper as ``module`` (:56)
        holder = "attn_block" if cfg.attn_adapter_type == "normal" else "module"
        return f"lm.transformer.h.{i}.attn.{holder}.attention."
    return f"lm.transformer.h.{i}.attn.attention."


def mlp_prefix(cfg: OracleConfig, i: int) -> str:
    if cfg.mlp_adapter_hidden:
        # Sequential(mlp, Adapter) (magma.py:143-148) vs ParallelAdapter(module=mlp) (magma.py:129-136)
        return f"lm.transformer.h.{i}.mlp." + ("0." if cfg.mlp_adapter_type == "normal" else "module.")
    return f"lm.transformer.h.{i}.mlp."


def mlp_adapter_prefix(cfg: OracleConfig, i: int) -> str:
    return f"lm.transformer.h.{i}.mlp." + ("1.adapter." if cfg.mlp_adapter_type == "normal" else "adapter.")


def enc_conv_specs(cfg: OracleConfig) -> List[Tuple[str, int, int, int]]:
    """(name, cin, cout, k) for every conv of the ModifiedResNet trunk, in
    forward order.  [UNVENDORED openai/CLIP model.py ModifiedResNet]"""
    w = cfg.enc_width
    specs = [("conv1", 3, w // 2, 3), ("conv2", w // 2, w // 2, 3), ("conv3", w // 2, w, 3)]
    inplanes = w
    for li, (mult, blocks) in enumerate(zip((1, 2, 4, 8), cfg.enc_layers), start=1):
        planes = w * mult
        for j in range(blocks):
            stride = 2 if (li > 1 and j == 0) else 1
            pre = f"layer{li}.{j}."
            specs.append((pre + "conv1", inplanes, planes, 1))
            specs.append((pre + "conv2", planes, planes, 3))
            specs.append((pre + "conv3", planes, planes * 4, 1))
            if stride > 1 or inplanes != planes * 4:
                specs.append((pre + "downsample.0", inplanes, planes * 4, 1))
            inplanes = planes * 4
    return specs


def bn_name_for(conv_name: str) -> str:
    if conv_name.endswith("downsample.0"):
        return conv_name[:-1] + "1"
    return conv_name.replace("conv", "bn")


def init_params(cfg: OracleConfig, seed: int = 0, dtype=torch.float32,
                lm_std: float = 0.02, randomize_bn: bool = True, layer_seeds: bool = False) -> Params:
    """Seeded synthetic weights: N(0, lm_std) for LM/proj, Kaiming for
    convs, adapters per reference magma/adapters.py:28-33 (N(0,1e-3) clamped to
    +-2e-3).  BatchNorm statistics are randomised (not the identity) so that
    the parity tests actually exercise the scale/shift arithmetic."""
    g = torch.Generator().manual_seed(seed)
    p: Params = {}
    d = cfg.d_model

    def normal(*shape, std):
        return torch.randn(*shape, generator=g, dtype=torch.float32) * std

    def adapter(prefix, hidden, normal=None):
        normal = normal or globals_normal[0]
        dn_i, up_i = ("1", "3") if cfg.adapter_layernorm else ("0", "2")
        for idx, (o, i_) in ((dn_i, (hidden, d)), (up_i, (d, hidden))):
            p[f"{prefix}{idx}.weight"] = normal(o, i_, std=1e-3).clamp_(-2e-3, 2e-3)
            p[f"{prefix}{idx}.bias"] = normal(o, std=1e-3).clamp_(-2e-3, 2e-3)
        if cfg.adapter_layernorm:      # (the reference initialises it to identity, adapters.py:34-36; perturbed here so that it matters)
            p[f"{prefix}0.weight"] = 1.0 + normal(d, std=0.05)
            p[f"{prefix}0.bias"] = normal(d, std=0.02)

    globals_normal = [normal]

    def block(i, normal, adapter):
        h = f"lm.transformer.h.{i}."
        p[h + "ln_1.weight"] = 1.0 + normal(d, std=0.05)
        p[h + "ln_1.bias"] = normal(d, std=0.02)
        ap = attn_prefix(cfg, i)
        for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
            p[ap + n + ".weight"] = normal(d, d, std=lm_std)
        mp = mlp_prefix(cfg, i)
        p[mp + "c_fc.weight"] = normal(cfg.d_ff, d, std=lm_std)
        p[mp + "c_fc.bias"] = normal(cfg.d_ff, std=lm_std)
        p[mp + "c_proj.weight"] = normal(d, cfg.d_ff, std=lm_std)
        p[mp + "c_proj.bias"] = normal(d, std=lm_std)
        if cfg.mlp_adapter_hidden:
            adapter(mlp_adapter_prefix(cfg, i), cfg.mlp_adapter_hidden)
            if cfg.mlp_adapter_type == "scaled_parallel":
                p[h + "mlp.adapter_scale"] = 1.0 + normal(1, std=0.3)
        if cfg.attn_adapter_hidden:
            adapter(h + "attn.adapter.", cfg.attn_adapter_hidden)
            if cfg.attn_adapter_type == "scaled_parallel":
                p[h + "attn.adapter_scale"] = 1.0 + normal(1, std=0.3)

    p["lm.transformer.wte.weight"] = normal(cfg.vocab_in, d, std=lm_std)
    if layer_seeds:
        # full-depth fixtures (28 x 0.2 B values): one generator per block, blocks drawn concurrently (torch.randn is
        # single-threaded per call and releases the GIL) -- a different stream than the sequential default, which the
        # margin-searched fixtures of tests/fullwidth_common.py depend on and which therefore stays as it was
        from concurrent.futures import ThreadPoolExecutor

        def one(i):
            gi = torch.Generator().manual_seed(seed * 1000003 + 7919 * (i + 1))

            def normal_i(*shape, std):
                return torch.randn(*shape, generator=gi, dtype=torch.float32) * std

            def adapter_i(prefix, hidden):
                adapter(prefix, hidden, normal_i)
            block(i, normal_i, adapter_i)

        with ThreadPoolExecutor(max_workers=min(16, max(1, cfg.n_layer))) as ex:
            list(ex.map(one, range(cfg.n_layer)))
    else:
        for i in range(cfg.n_layer):
            block(i, normal, adapter)
    p["lm.transformer.ln_f.weight"] = 1.0 + normal(d, std=0.05)
    p["lm.transformer.ln_f.bias"] = normal(d, std=0.02)
    p["lm.lm_head.weight"] = normal(cfg.vocab_out, d, std=lm_std)
    p["lm.lm_head.bias"] = normal(cfg.vocab_out, std=lm_std)

    e = "image_prefix.enc."
    for name, cin, cout, k in enc_conv_specs(cfg):
        fan_in = cin * k * k
        p[e + name + ".weight"] = normal(cout, cin, k, k, std=math.sqrt(2.0 / fan_in))
        bn = e + bn_name_for(name)
        if randomize_bn:
            # last BN of each bottleneck gets a small gain so the residual
            # trunk stays O(1) through 40 blocks
            gain = 0.5 if (name.endswith("conv3") and "layer" in name) else 1.0
            p[bn + ".weight"] = (1.0 + normal(cout, std=0.1)) * gain
            p[bn + ".bias"] = normal(cout, std=0.1)
            p[bn + ".running_mean"] = normal(cout, std=0.1)
            p[bn + ".running_var"] = 1.0 + 0.2 * torch.rand(cout, generator=g)
        else:
            p[bn + ".weight"] = torch.ones(cout)
            p[bn + ".bias"] = torch.zeros(cout)
            p[bn + ".running_mean"] = torch.zeros(cout)
            p[bn + ".running_var"] = torch.ones(cout)
    p["image_prefix.proj.weight"] = normal(d, cfg.enc_out_dim, std=lm_std)
    p["image_prefix.proj.bias"] = normal(d, std=lm_std)
    if cfg.use_prefix_ln:
        p["image_prefix.ln.weight"] = 1.0 + normal(d, std=0.05)
        p["image_prefix.ln.bias"] = normal(d, std=0.02)
    if dtype != torch.float32:
        p = {k: v.to(dtype) for k, v in p.items()}
    return p


# ----------------------------------------------------------------------------
# GPT-J arithmetic  [UNVENDORED fork; restated from the published GPT-J
# algorithm, cross-checked against HF modeling_gptj.py in tests]
# ----------------------------------------------------------------------------

def gelu_new(x: torch.Tensor) -> torch.Tensor:
    # HF activations.NewGELUActivation (tanh form)
    return 0.5 * x * (1.0 + torch.tanh(math.sqrt(2.0 / math.pi) * (x + 0.044715 * x.pow(3))))


def rotary_tables(rotary_dim: int, n_pos: int):
    """sin/cos of pos * 10000^(-2i/rotary_dim), i < rotary_dim/2 (HF
    modeling_gptj.py:47-53 create_sinusoidal_positions)."""
    inv_freq = 1.0 / (10000 ** (torch.arange(0, rotary_dim, 2, dtype=torch.int64).float() / rotary_dim))
    ang = torch.einsum("i,j->ij", torch.arange(n_pos, dtype=torch.int64).float(), inv_freq)
    return torch.sin(ang), torch.cos(ang)


def apply_rotary(x: torch.Tensor, pos: torch.Tensor, rotary_dim: int) -> torch.Tensor:
    """x: (B, S, H, Dh). GPT-J interleaved pairs on the first rotary_dim dims
    (HF modeling_gptj.py:56-67 rotate_every_two / apply_rotary_pos_emb)."""
    sin_t, cos_t = rotary_tables(rotary_dim, int(pos.max().item()) + 1)
    sin = sin_t[pos].to(x.dtype).repeat_interleave(2, dim=-1)[None, :, None, :]
    cos = cos_t[pos].to(x.dtype).repeat_interleave(2, dim=-1)[None, :, None, :]
    xr, xp = x[..., :rotary_dim], x[..., rotary_dim:]
    x1, x2 = xr[..., ::2], xr[..., 1::2]
    rot = torch.stack((-x2, x1), dim=-1).flatten(-2)
    return torch.cat((xr * cos + rot * sin, xp), dim=-1)


ADAPTER_ACTS = {"relu": F.relu, "gelu": F.gelu, "gelu_tanh": lambda v: F.gelu(v, approximate="tanh")}


def adapter_branch(p: Params, prefix: str, x: torch.Tensor, act: str = "relu") -> torch.Tensor:
    """The adapter Sequential itself (adapters.py:14-26): [LayerNorm ->] Linear -> activation -> Linear.  The LayerNorm variant
    is recognised by its keys (a fourth module: ``3.weight`` is the up-projection then)."""
    if prefix + "3.weight" in p:
        x = F.layer_norm(x, (x.shape[-1],), p[prefix + "0.weight"], p[prefix + "0.bias"], 1e-5)
        dn, up = "1", "3"
    else:
        dn, up = "0", "2"
    h = ADAPTER_ACTS[act](F.linear(x, p[prefix + dn + ".weight"], p[prefix + dn + ".bias"]))
    return F.linear(h, p[prefix + up + ".weight"], p[prefix + up + ".bias"])


def adapter_fwd(p: Params, prefix: str, x: torch.Tensor, act: str = "relu") -> torch.Tensor:
    """reference magma/adapters.py:38-39: adapter(x) + x."""
    return adapter_branch(p, prefix, x, act) + x


def parallel_adapter_fwd(p: Params, prefix: str, scale_key: Optional[str], x: torch.Tensor, y: torch.Tensor,
                         act: str = "relu") -> torch.Tensor:
    """reference magma/adapters.py:62-65 / 88-92: y = module(x); return y + adapter(x) * adapter_scale
    (adapter_scale = 1 for "parallel", the trainable scalar for "scaled_parallel")."""
    z = adapter_branch(p, prefix, x, act)
    return y + z * (p[scale_key] if scale_key is not None and scale_key in p else 1)


def attention_fwd(p: Params, cfg: OracleConfig, i: int, x: torch.Tensor,
                  past: Optional[Tuple[torch.Tensor, torch.Tensor]], pos0: int):
    """HF modeling_gptj.py:136-226: q/k/v no bias, rotary on q,k, fp32
    QK^T / sqrt(dh), causal mask, softmax fp32, cast to v dtype, @v, out_proj."""
    B, S, d = x.shape
    H, Dh = cfg.n_head, cfg.head_dim
    ap = attn_prefix(cfg, i)
    q = F.linear(x, p[ap + "q_proj.weight"]).view(B, S, H, Dh)
    k = F.linear(x, p[ap + "k_proj.weight"]).view(B, S, H, Dh)
    v = F.linear(x, p[ap + "v_proj.weight"]).view(B, S, H, Dh)
    pos = torch.arange(pos0, pos0 + S)
    q = apply_rotary(q, pos, cfg.rotary_dim).permute(0, 2, 1, 3)
    k = apply_rotary(k, pos, cfg.rotary_dim).permute(0, 2, 1, 3)
    v = v.permute(0, 2, 1, 3)
    if past is not None:
        k = torch.cat((past[0], k), dim=2)
        v = torch.cat((past[1], v), dim=2)
    present = (k, v)
    T = k.shape[2]
    scores = torch.matmul(q.float(), k.float().transpose(-1, -2)) / math.sqrt(Dh)
    qpos = torch.arange(T - S, T)[:, None]
    mask = torch.arange(T)[None, :] <= qpos
    scores = torch.where(mask, scores, torch.finfo(scores.dtype).min)
    w = torch.softmax(scores, dim=-1).to(v.dtype)
    o = torch.matmul(w, v).permute(0, 2, 1, 3).reshape(B, S, d)
    o = F.linear(o, p[ap + "out_proj.weight"])
    if cfg.attn_adapter_hidden and cfg.attn_adapter_type == "normal":
        # reference magma/adapters.py:109-116 AdapterWrapper
        o = adapter_fwd(p, f"lm.transformer.h.{i}.attn.adapter.", o, cfg.adapter_act)
    elif cfg.attn_adapter_hidden:
        # reference magma/adapters.py:82-92 ParallelAdapterWrapper: the adapter reads the attention block's input x
        o = parallel_adapter_fwd(p, f"lm.transformer.h.{i}.attn.adapter.", f"lm.transformer.h.{i}.attn.adapter_scale", x, o, cfg.adapter_act)
    return o, present


def mlp_fwd(p: Params, cfg: OracleConfig, i: int, x: torch.Tensor) -> torch.Tensor:
    mp = mlp_prefix(cfg, i)
    h = gelu_new(F.linear(x, p[mp + "c_fc.weight"], p[mp + "c_fc.bias"]))
    m = F.linear(h, p[mp + "c_proj.weight"], p[mp + "c_proj.bias"])
    if cfg.mlp_adapter_hidden and cfg.mlp_adapter_type == "normal":
        # reference magma/magma.py:143-149: Sequential(mlp, Adapter)
        m = adapter_fwd(p, mlp_adapter_prefix(cfg, i), m, cfg.adapter_act)
    elif cfg.mlp_adapter_hidden:
        # reference magma/magma.py:129-136 + adapters.py:62-65: ParallelAdapter(module=mlp)
        m = parallel_adapter_fwd(p, mlp_adapter_prefix(cfg, i), f"lm.transformer.h.{i}.mlp.adapter_scale", x, m, cfg.adapter_act)
    return m


def block_fwd(p: Params, cfg: OracleConfig, i: int, x: torch.Tensor, past, pos0: int):
    """GPT-J parallel-residual block: attn(ln(x)) + mlp(ln(x)) + x
    (HF modeling_gptj.py:382-414)."""
    h = f"lm.transformer.h.{i}."
    ln = F.layer_norm(x, (cfg.d_model,), p[h + "ln_1.weight"], p[h + "ln_1.bias"], cfg.ln_eps)
    a, present = attention_fwd(p, cfg, i, ln, past, pos0)
    m = mlp_fwd(p, cfg, i, ln)
    return a + m + x, present


def shifted_ce(logits: torch.Tensor, labels: torch.Tensor) -> torch.Tensor:
    """HF loss_utils.ForCausalLMLoss: fp32 logits, shift by one, ignore -100,
    mean over valid targets."""
    lg = logits.float()[:, :-1, :].contiguous()
    lb = labels[:, 1:].contiguous()
    return F.cross_entropy(lg.view(-1, lg.shape[-1]), lb.view(-1), ignore_index=-100)


def lm_forward(p: Params, cfg: OracleConfig, inputs_embeds: Optional[torch.Tensor] = None,
               input_ids: Optional[torch.Tensor] = None, past=None,
               labels: Optional[torch.Tensor] = None, n_layer: Optional[int] = None):
    """The call the reference makes at magma/magma.py:270-274 and
    magma/sampling.py:81-90.  Returns dict(logits, loss, past_key_values, hidden)."""
    if inputs_embeds is None:
        inputs_embeds = F.embedding(input_ids, p["lm.transformer.wte.weight"])
    x = inputs_embeds
    pos0 = 0 if past is None else past[0][0].shape[2]
    presents = []
    L = cfg.n_layer if n_layer is None else n_layer
    for i in range(L):
        x, pr = block_fwd(p, cfg, i, x, None if past is None else past[i], pos0)
        presents.append(pr)
    x = F.layer_norm(x, (cfg.d_model,), p["lm.transformer.ln_f.weight"], p["lm.transformer.ln_f.bias"], cfg.ln_eps)
    logits = F.linear(x, p["lm.lm_head.weight"], p["lm.lm_head.bias"])
    loss = shifted_ce(logits, labels) if labels is not None else None
    return {"logits": logits, "loss": loss, "past_key_values": presents, "hidden": x}


# ----------------------------------------------------------------------------
# CLIP ModifiedResNet trunk minus attnpool  [UNVENDORED openai/CLIP model.py]
# reference call site: magma/image_encoders.py:65-74
# ----------------------------------------------------------------------------

def _conv_bn_impl(p: Params, cfg: OracleConfig, name: str, x: torch.Tensor, relu: bool,
                  stride: int = 1, bn_train: bool = False) -> torch.Tensor:
    e = "image_prefix.enc."
    w = p[e + name + ".weight"]
    x = F.conv2d(x, w, None, stride=stride, padding=w.shape[-1] // 2)
    bn = e + bn_name_for(name)
    x = F.batch_norm(x, p[bn + ".running_mean"], p[bn + ".running_var"], p[bn + ".weight"],
                     p[bn + ".bias"], training=bn_train, eps=cfg.bn_eps)
    return F.relu(x) if relu else x


def encoder_fwd(p: Params, cfg: OracleConfig, x: torch.Tensor, bn_train: bool = False) -> torch.Tensor:
    """(B,3,H,W) -> (B, (H/32)*(W/32), 32*width).  BatchNorm in eval mode by default
    (clip.load returns .eval() and Magma.__init__ never flips it); ``bn_train=True`` = batch statistics,
    the reference's behaviour after its first eval phase (train.py:164,182) -- the running statistics in ``p`` are
    updated in place, as nn.BatchNorm2d does."""
    from functools import partial
    cb = partial(_conv_bn_impl, bn_train=bn_train)
    return _encoder_body(p, cfg, x, cb)


def _encoder_body(p, cfg, x, _conv_bn):
    x = _conv_bn(p, cfg, "conv1", x, True, stride=2)
    x = _conv_bn(p, cfg, "conv2", x, True)
    x = _conv_bn(p, cfg, "conv3", x, True)
    x = F.avg_pool2d(x, 2)
    w = cfg.enc_width
    inplanes = w
    for li, (mult, blocks) in enumerate(zip((1, 2, 4, 8), cfg.enc_layers), start=1):
        planes = w * mult
        for j in range(blocks):
            stride = 2 if (li > 1 and j == 0) else 1
            pre = f"layer{li}.{j}."
            identity = x
            out = _conv_bn(p, cfg, pre + "conv1", x, True)
            out = _conv_bn(p, cfg, pre + "conv2", out, True)
            if stride > 1:
                out = F.avg_pool2d(out, stride)
            out = _conv_bn(p, cfg, pre + "conv3", out, False)
            if stride > 1 or inplanes != planes * 4:
                if stride > 1:
                    identity = F.avg_pool2d(identity, stride)
                identity = _conv_bn(p, cfg, pre + "downsample.0", identity, False)
            x = F.relu(out + identity)
            inplanes = planes * 4
    B, C, H, W = x.shape
    # reference magma/image_encoders.py:72-74: rearrange "b d h w -> b (h w) d"
    return x.permute(0, 2, 3, 1).reshape(B, H * W, C)


# ----------------------------------------------------------------------------
# CLIP VisionTransformer (encoder_name "clip" = ViT-B/32)  [UNVENDORED openai/CLIP model.py VisionTransformer]
# reference call site: magma/image_encoders.py:56-63 (clip.load(name)[0].visual, used whole: ln_post + proj included),
# consumed by the pooled branch of ImagePrefix (magma/image_prefix.py:60-72,85-101).
# Restated from the published CLIP architecture; pinned against the independent statement installed here,
# HF CLIPVisionModelWithProjection (tests/test_oracle_vs_hf.py).  Parameter names are openai/CLIP's.
# ----------------------------------------------------------------------------
@dataclass
class ViTConfig:
    width: int = 768
    layers: int = 12
    heads: int = 12
    patch: int = 32
    resolution: int = 224
    out_dim: int = 512
    ln_eps: float = 1e-5

    @property
    def tokens(self) -> int:
        return (self.resolution // self.patch) ** 2 + 1


def init_vit_params(v: ViTConfig, seed: int = 0, prefix: str = "image_prefix.enc.") -> Params:
    g = torch.Generator().manual_seed(seed)
    w, sc = v.width, v.width ** -0.5

    def rn(*shape, std):
        return torch.randn(*shape, generator=g) * std

    p: Params = {prefix + "conv1.weight": rn(w, 3, v.patch, v.patch, std=(3 * v.patch * v.patch) ** -0.5),
                 prefix + "class_embedding": rn(w, std=sc), prefix + "positional_embedding": rn(v.tokens, w, std=sc),
                 prefix + "proj": rn(w, v.out_dim, std=sc)}
    for name in ("ln_pre", "ln_post"):
        p[prefix + name + ".weight"] = 1.0 + rn(w, std=0.05)
        p[prefix + name + ".bias"] = rn(w, std=0.02)
    for i in range(v.layers):
        b = f"{prefix}transformer.resblocks.{i}."
        for name in ("ln_1", "ln_2"):
            p[b + name + ".weight"] = 1.0 + rn(w, std=0.05)
            p[b + name + ".bias"] = rn(w, std=0.02)
        p[b + "attn.in_proj_weight"] = rn(3 * w, w, std=sc)
        p[b + "attn.in_proj_bias"] = rn(3 * w, std=0.02)
        p[b + "attn.out_proj.weight"] = rn(w, w, std=sc)
        p[b + "attn.out_proj.bias"] = rn(w, std=0.02)
        p[b + "mlp.c_fc.weight"] = rn(4 * w, w, std=sc)
        p[b + "mlp.c_fc.bias"] = rn(4 * w, std=0.02)
        p[b + "mlp.c_proj.weight"] = rn(w, 4 * w, std=(4 * w) ** -0.5)
        p[b + "mlp.c_proj.bias"] = rn(w, std=0.02)
    return p


def vit_encoder_fwd(p: Params, v: ViTConfig, x: torch.Tensor, prefix: str = "image_prefix.enc.") -> torch.Tensor:
    """(B,3,R,R) -> (B, out_dim): patch conv (stride = kernel), [class | patches] + positional embedding, ln_pre, `layers` x
    {x += MHA(ln_1 x); x += c_proj(QuickGELU(c_fc(ln_2 x)))}, ln_post on the class token, @ proj."""
    B = x.shape[0]
    w, H = v.width, v.heads
    t = F.conv2d(x, p[prefix + "conv1.weight"], None, stride=v.patch)            # (B, w, g, g)
    t = t.reshape(B, w, -1).permute(0, 2, 1)
    cls = p[prefix + "class_embedding"].to(t.dtype).expand(B, 1, w)
    t = torch.cat((cls, t), dim=1) + p[prefix + "positional_embedding"].to(t.dtype)
    t = F.layer_norm(t, (w,), p[prefix + "ln_pre.weight"], p[prefix + "ln_pre.bias"], v.ln_eps)
    S, dh = t.shape[1], w // H
    for i in range(v.layers):
        b = f"{prefix}transformer.resblocks.{i}."
        h = F.layer_norm(t, (w,), p[b + "ln_1.weight"], p[b + "ln_1.bias"], v.ln_eps)
        qkv = F.linear(h, p[b + "attn.in_proj_weight"], p[b + "attn.in_proj_bias"])
        q, k, vv = (u.reshape(B, S, H, dh).permute(0, 2, 1, 3) for u in qkv.chunk(3, dim=-1))
        att = torch.softmax(torch.matmul(q.float(), k.float().transpose(-1, -2)) / math.sqrt(dh), dim=-1).to(vv.dtype)
        o = torch.matmul(att, vv).permute(0, 2, 1, 3).reshape(B, S, w)
        t = t + F.linear(o, p[b + "attn.out_proj.weight"], p[b + "attn.out_proj.bias"])
        h = F.layer_norm(t, (w,), p[b + "ln_2.weight"], p[b + "ln_2.bias"], v.ln_eps)
        h = F.linear(h, p[b + "mlp.c_fc.weight"], p[b + "mlp.c_fc.bias"])
        h = h * torch.sigmoid(1.702 * h)                                         # QuickGELU
        t = t + F.linear(h, p[b + "mlp.c_proj.weight"], p[b + "mlp.c_proj.bias"])
    c = F.layer_norm(t[:, 0, :], (w,), p[prefix + "ln_post.weight"], p[prefix + "ln_post.bias"], v.ln_eps)
    return c @ p[prefix + "proj"].to(c.dtype)


def pooled_prefix_fwd(p: Params, d_model: int, seq_len: int, feats: torch.Tensor, ln_eps: float = 1e-5,
                      dropout_mask: Optional[torch.Tensor] = None) -> torch.Tensor:
    """reference magma/image_prefix.py:85-109 for encoders without a token grid (feats (B, enc_dim)):
    Linear(enc_dim -> seq_len * d) -> "b (s d) -> b s d" -> dropout -> LayerNorm(d)."""
    x = F.linear(feats, p["image_prefix.proj.weight"], p["image_prefix.proj.bias"])
    x = x.reshape(feats.shape[0], seq_len, d_model)
    if dropout_mask is not None:
        x = x * dropout_mask
    if "image_prefix.ln.weight" in p:
        x = F.layer_norm(x, (d_model,), p["image_prefix.ln.weight"], p["image_prefix.ln.bias"], ln_eps)
    return x


def image_prefix_fwd(p: Params, cfg: OracleConfig, images: torch.Tensor,
                     dropout_mask: Optional[torch.Tensor] = None, bn_train: bool = False) -> torch.Tensor:
    """reference magma/image_prefix.py:78-109: enc -> proj -> dropout -> ln.
    ``dropout_mask`` (already scaled by 1/(1-p)) makes train-mode deterministic."""
    feats = encoder_fwd(p, cfg, images, bn_train=bn_train)
    x = F.linear(feats, p["image_prefix.proj.weight"], p["image_prefix.proj.bias"])
    if dropout_mask is not None:
        x = x * dropout_mask
    if cfg.use_prefix_ln:
        x = F.layer_norm(x, (cfg.d_model,), p["image_prefix.ln.weight"], p["image_prefix.ln.bias"], cfg.ln_eps)
    return x


# ----------------------------------------------------------------------------
# labels, full forward, generate
# ----------------------------------------------------------------------------

def build_labels(prefix_len: int, captions: torch.Tensor, eos_token: int) -> torch.Tensor:
    """reference magma/utils.py:334-364, literal loop form (integer, exact)."""
    B = captions.shape[0]
    assert captions.shape[1] >= prefix_len
    labels = torch.cat((torch.full((B, prefix_len), -100, dtype=torch.int64),
                        captions[:, : captions.shape[1] - prefix_len].to(torch.int64)), dim=1)
    for row in labels:
        for k in range(row.shape[0]):
            if int(row[k]) == eos_token:
                row[k + 1:] = -100
                break
    return labels


def magma_forward(p: Params, cfg: OracleConfig, images: torch.Tensor, captions: torch.Tensor,
                  dropout_mask: Optional[torch.Tensor] = None, n_layer: Optional[int] = None, bn_train: bool = False):
    """reference magma/magma.py:238-276."""
    prefix = image_prefix_fwd(p, cfg, images, dropout_mask, bn_train=bn_train)
    P = prefix.shape[1]
    labels = build_labels(P, captions, cfg.eos_token)
    words = F.embedding(captions, p["lm.transformer.wte.weight"]).to(prefix.dtype)
    emb = torch.cat((prefix, words[:, : captions.shape[1] - P, :]), dim=1)
    out = lm_forward(p, cfg, inputs_embeds=emb, labels=labels, n_layer=n_layer)
    out["labels"] = labels
    out["prefix"] = prefix
    return out


def embed(p: Params, cfg: OracleConfig, inputs: List[torch.Tensor]) -> torch.Tensor:
    """reference magma/magma.py:195-212."""
    outs = []
    for x in inputs:
        if x.ndim == 2:
            outs.append(F.embedding(x, p["lm.transformer.wte.weight"]))
        elif x.ndim == 4:
            outs.append(image_prefix_fwd(p, cfg, x.to(p["image_prefix.proj.weight"].dtype)))
        else:
            raise ValueError(f"Expected 2d or 4d tensor, got {x.ndim}d")
    return torch.cat(outs, dim=1)


@torch.no_grad()
def generate_greedy(p: Params, cfg: OracleConfig, embeddings: torch.Tensor, max_steps: int,
                    stop_on_eos: bool = True):
    """reference magma/sampling.py:43-121 with temperature == 0.0 (argmax).
    Returns (tokens (B, S0+steps) int64, per-step fp32 last-row logits list)."""
    B, S0, _ = embeddings.shape
    out = torch.full((B, S0), cfg.image_token, dtype=torch.int64)
    past = None
    step_logits = []
    for i in range(max_steps):
        if i == 0:
            r = lm_forward(p, cfg, inputs_embeds=embeddings, past=None)
        else:
            r = lm_forward(p, cfg, input_ids=out[:, -1:], past=past)
        logits = r["logits"][:, -1, :].float()
        past = r["past_key_values"]
        step_logits.append(logits)
        nxt = torch.argmax(logits, dim=-1, keepdim=True)
        out = torch.cat((out, nxt), dim=-1)
        if stop_on_eos and bool((nxt == cfg.eos_token).all()):
            break
    return out, step_logits


# sampling filters, restated literally (reference magma/sampling.py:7-30; Q6)
def top_p_filter(logits: torch.Tensor, threshold: float = 0.9) -> torch.Tensor:
    sorted_logits, sorted_indices = torch.sort(logits, descending=True)
    cum_probs = torch.cumsum(F.softmax(sorted_logits, dim=-1), dim=-1)
    rm = cum_probs < (1 - threshold)
    rm[..., 1:] = rm[..., :-1].clone()
    rm[..., 0] = 0
    sorted_logits[rm] = float("-inf")
    return sorted_logits.scatter(1, sorted_indices, sorted_logits)


def top_k_filter(logits: torch.Tensor, k: int) -> torch.Tensor:
    assert k > 0
    val, ind = torch.topk(logits, k)
    probs = torch.full_like(logits, float("-inf"))
    probs.scatter_(1, ind, val)
    return probs
