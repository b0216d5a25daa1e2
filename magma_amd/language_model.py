"""GPT-J-6B for the MI355X MAGMA path.

The reference builds GPT-J from a fork of HF transformers (reference
magma/language_model.py:12-45: 28 layers, 16 heads x 256, rotary_dim 64, the
"jax" parallel-residual block, vocab 50400).  Here the module tree only holds
the parameters under the fork's names:
    transformer.wte, transformer.h.{i}.ln_1, .attn.attention.{q,k,v,out}_proj,
    .mlp.{c_fc,c_proj}, transformer.ln_f, lm_head (untied, with bias)
and ``forward`` hands the call to the HIP engine (magma_amd/engine.py).  It
accepts the two call forms the reference uses:
    lm(inputs_embeds=..., labels=..., output_hidden_states=...)     magma.py:270-274
    lm(inputs_embeds=...|input_ids=..., use_cache=True, past_key_values=...)
                                                                   sampling.py:81-90
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Any, Optional

import torch
import torch.nn as nn


@dataclass
class GPTJConfig:
    vocab_size: int = 50400            # rows of wte (the INPUT vocabulary)
    vocab_out: Optional[int] = None    # rows of lm_head; None = vocab_size.  SURVEY Q1: the reference resizes the token
                                       # embeddings to len(tokenizer) = 50258 (magma.py:50) while whether the untied 50400-row
                                       # head follows is decided inside the un-vendored fork -- both sizes are independent here
    hidden_size: int = 4096
    num_layers: int = 28
    num_heads: int = 16
    rotary_dim: int = 64
    intermediate_size: int = 16384
    max_position_embeddings: int = 2048
    layer_norm_epsilon: float = 1e-5
    rotary: bool = True
    jax: bool = True
    gradient_checkpointing: bool = False  # 288 GB HBM: activations are kept, not recomputed
    use_cache: bool = True
    pad_token_id: Optional[int] = None
    init_std: float = 0.02

    @property
    def head_dim(self):
        return self.hidden_size // self.num_heads

    @property
    def head_rows(self) -> int:
        return self.vocab_size if self.vocab_out is None else self.vocab_out


def gptj_config(**overrides) -> GPTJConfig:
    return GPTJConfig(**overrides)


class LMOutput(dict):
    """Minimal ModelOutput: attribute + key access (.loss, .logits, .past_key_values, .hidden_states).

    A value may be LAZY (``LMOutput.lazy(fn)``): it is computed on first access and then stored.  The training-form forward
    uses it for ``.logits``: the reference materialises (B, 2048, V) logits on every call (magma.py:270-276); here the loss
    head runs on the rows that carry a target and the full tensor is produced only if a caller actually reads it."""

    class lazy:
        def __init__(self, fn):
            self.fn = fn

    def __getitem__(self, k):
        v = dict.__getitem__(self, k)
        if isinstance(v, LMOutput.lazy):
            v = v.fn()
            dict.__setitem__(self, k, v)
        return v

    def get(self, k, default=None):
        return self[k] if k in self else default

    def values(self):
        return [self[k] for k in self.keys()]

    def items(self):
        return [(k, self[k]) for k in self.keys()]

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def is_lazy(self, k) -> bool:
        """True while ``k`` has not been computed: ``out.is_lazy("logits")`` does not launch the (B*S x V) head GEMM that
        ``out.logits is not None`` would."""
        return isinstance(dict.get(self, k), LMOutput.lazy)

    def resolve(self) -> "LMOutput":
        """Compute every lazy value (afterwards the object is a plain dict of tensors and releases what the closures held)."""
        for k in list(self.keys()):
            self[k]
        return self

    # conversions that take dict's C fast path (dict(out), {**out}, out.copy(), pickle / torch.save) would hand out the raw
    # lazy object -- an unpicklable closure: resolve first
    def copy(self):
        return LMOutput(self.resolve())

    def __iter__(self):
        return dict.__iter__(self)

    def keys(self):
        return dict.keys(self)

    def __reduce__(self):
        return (LMOutput, (dict(self.items()),))

    def to_dict(self) -> dict:
        return dict(self.items())


class _AttnParams(nn.Module):
    def __init__(self, d, **kw):
        super().__init__()
        self.q_proj = nn.Linear(d, d, bias=False, **kw)
        self.k_proj = nn.Linear(d, d, bias=False, **kw)
        self.v_proj = nn.Linear(d, d, bias=False, **kw)
        self.out_proj = nn.Linear(d, d, bias=False, **kw)


class SelfAttention(nn.Module):
    """``attn.attention.*`` as in GPT-Neo/the fork."""

    def __init__(self, d, **kw):
        super().__init__()
        self.attention = _AttnParams(d, **kw)


class MLP(nn.Module):
    def __init__(self, d, ff, **kw):
        super().__init__()
        self.c_fc = nn.Linear(d, ff, **kw)
        self.c_proj = nn.Linear(ff, d, **kw)


class Block(nn.Module):
    def __init__(self, cfg: GPTJConfig, **kw):
        super().__init__()
        self.ln_1 = nn.LayerNorm(cfg.hidden_size, eps=cfg.layer_norm_epsilon, **kw)
        self.attn = SelfAttention(cfg.hidden_size, **kw)
        self.mlp = MLP(cfg.hidden_size, cfg.intermediate_size, **kw)


class Transformer(nn.Module):
    def __init__(self, cfg: GPTJConfig, **kw):
        super().__init__()
        self.wte = nn.Embedding(cfg.vocab_size, cfg.hidden_size, **kw)
        self.h = nn.ModuleList([Block(cfg, **kw) for _ in range(cfg.num_layers)])
        self.ln_f = nn.LayerNorm(cfg.hidden_size, eps=cfg.layer_norm_epsilon, **kw)


class GPTJForCausalLM(nn.Module):
    def __init__(self, config: GPTJConfig, device=None, dtype=None, init: bool = True):
        super().__init__()
        self.config = config
        # parameters are created uninitialised directly on the target device
        # (the reference used no_init_weights() on the CPU, language_model.py:43-44)
        kw = dict(device=device, dtype=dtype)
        with torch.no_grad():
            _skip = nn.init  # noqa: F841  (torch inits run on-device; cheap on a GPU)
            self.transformer = Transformer(config, **kw)
            self.lm_head = nn.Linear(config.hidden_size, config.head_rows, bias=True, **kw)
        if init:
            self.init_weights()
        self._engine = None

    @torch.no_grad()
    def init_weights(self, seed: Optional[int] = None):
        """Random init N(0, init_std) for matrices/embeddings, LayerNorm = identity."""
        g = None
        if seed is not None:
            g = torch.Generator(device=self.lm_head.weight.device).manual_seed(seed)
        std = self.config.init_std
        for m in self.modules():
            if isinstance(m, (nn.Linear, nn.Embedding)) and not getattr(m, "_is_adapter", False):
                m.weight.normal_(0.0, std, generator=g)
                if getattr(m, "bias", None) is not None:
                    m.bias.normal_(0.0, std, generator=g)
            elif isinstance(m, nn.LayerNorm):
                m.weight.fill_(1.0)
                m.bias.zero_()

    def resize_token_embeddings(self, new_num_tokens: Optional[int] = None, resize_head: bool = True,
                                new_head_rows: Optional[int] = None):
        """the reference resizes to len(tokenizer) = 50258 (magma.py:50).  ``resize_head=True`` (HF's rule for an
        untied output embedding) gives lm_head the same row count; ``resize_head=False`` keeps the head (a checkpoint whose
        fork left the 50400-row head alone); ``new_head_rows`` sets the head on its own.  Leading rows are kept."""
        wte, head = self.transformer.wte, self.lm_head
        kw = dict(device=wte.weight.device, dtype=wte.weight.dtype)
        old_in, old_out = wte.weight.shape[0], head.weight.shape[0]
        new_in = old_in if new_num_tokens is None else int(new_num_tokens)
        new_out = int(new_head_rows) if new_head_rows is not None else (new_in if resize_head else old_out)
        if new_in != old_in:
            n = min(old_in, new_in)
            new_wte = nn.Embedding(new_in, self.config.hidden_size, **kw)
            with torch.no_grad():
                new_wte.weight.normal_(0.0, self.config.init_std)
                new_wte.weight[:n] = wte.weight[:n]
            self.transformer.wte = new_wte
        if new_out != old_out:
            n = min(old_out, new_out)
            new_head = nn.Linear(self.config.hidden_size, new_out, bias=True, **kw)
            with torch.no_grad():
                new_head.weight.normal_(0.0, self.config.init_std)
                new_head.bias.zero_()
                new_head.weight[:n] = head.weight[:n]
                new_head.bias[:n] = head.bias[:n]
            self.lm_head = new_head
        self.config.vocab_size = new_in
        self.config.vocab_out = None if new_out == new_in else new_out
        if new_in != old_in or new_out != old_out:
            self.invalidate_packed()
        return self.transformer.wte

    device_token_selection = True      # forward(..., sampling=, eos_token=, seed=) picks the next token in the HIP engine

    # ---- engine plumbing ----
    def invalidate_packed(self):
        """Drop device-layout copies of the weights (after load_state_dict / optimizer steps)."""
        self._engine = None

    @property
    def engine(self):
        if self._engine is None:
            from .engine import LMEngine
            self._engine = LMEngine(self)
        return self._engine

    def load_state_dict(self, *a, **k):
        r = super().load_state_dict(*a, **k)
        self.invalidate_packed()
        return r

    def forward(self, input_ids: Optional[torch.Tensor] = None, inputs_embeds: Optional[torch.Tensor] = None,
                labels: Optional[torch.Tensor] = None, use_cache: bool = False, past_key_values: Any = None,
                output_hidden_states: bool = False, cache_hint: Optional[int] = None, reuse_cache: bool = False,
                return_logits: bool = False, sampling=None, eos_token: Optional[int] = None,
                seed: Optional[int] = None, feed_back: bool = False, **unused) -> LMOutput:
        return self.engine.forward(input_ids=input_ids, inputs_embeds=inputs_embeds, labels=labels,
                                   use_cache=use_cache, past_key_values=past_key_values,
                                   output_hidden_states=output_hidden_states, cache_hint=cache_hint,
                                   reuse_cache=reuse_cache, return_logits=return_logits, sampling=sampling,
                                   eos_token=eos_token, seed=seed, feed_back=feed_back)


def get_gptj(gradient_checkpointing: bool = False, from_pretrained: bool = False, device=None,
             dtype=torch.bfloat16, config: Optional[GPTJConfig] = None, init: bool = True) -> GPTJForCausalLM:
    """Random-init GPT-J-6B (reference magma/language_model.py:27-45 signature + device/dtype)."""
    if from_pretrained:
        raise NotImplementedError("GPTJ pretrained not implemented")  # as in the reference
    cfg = config or gptj_config()
    cfg.gradient_checkpointing = gradient_checkpointing
    return GPTJForCausalLM(cfg, device=device, dtype=dtype, init=init)
