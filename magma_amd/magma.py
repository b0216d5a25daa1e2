"""Magma -- drop-in for reference magma/magma.py on MI355X.

Same public surface: ``Magma(config, device)``, ``from_checkpoint``,
``preprocess_inputs``, ``embed``, ``forward``, ``generate``, ``add_adapters``
and the attributes callers read (tokenizer, config, transforms, seq_len,
image_prefix, lm, word_embedding, transformer, image_token, eos_token, device).
The module tree keeps the reference's parameter names; the arithmetic runs in
libmagma_hip.so.  Differences: weights are created directly on the
GPU in bf16 (the reference builds fp32 on the CPU, then .half()); ``device``
must be a GPU -- there is no CPU execution path; SURVEY Q2 ("freeze_lm" as the
paper intends: only adapters, the image encoder and the prefix train)."""
from __future__ import annotations

from copy import deepcopy
from os.path import exists
from pathlib import Path
import os
from typing import List, Optional

import torch
import torch.nn as nn

from .adapters import Adapter, AdapterWrapper, ParallelAdapter, ParallelAdapterWrapper
from .config import MultimodalConfig
from .image_input import ImageInput
from .image_prefix import ImagePrefix
from .language_model import GPTJConfig, LMOutput, get_gptj
from .sampling import generate
from .transforms import get_transforms
from .utils import build_labels, get_tokenizer, print_main


def _load_checkpoint_file(path):
    """torch.load to host memory: the safe unpickler first, the reference's full unpickle (magma.py:292) only on opt-in."""
    import argparse
    import pickle
    safe = [argparse.Namespace]
    try:      # numpy scalars (DeepSpeed's step counters): the reconstructor, numpy.dtype and the dtype classes it is called with
        import numpy as np
        safe += [np.dtype, np.ndarray]
        safe += [c for c in vars(getattr(np, "dtypes", None) or object()).values() if isinstance(c, type)]
        for mod in ("numpy._core.multiarray", "numpy.core.multiarray"):
            try:
                m = __import__(mod, fromlist=["scalar"])
                safe += [m.scalar, m._reconstruct]
                break
            except Exception:
                continue
    except Exception:
        pass
    try:
        with torch.serialization.safe_globals(safe):
            return torch.load(path, map_location=torch.device("cpu"), weights_only=True)
    except pickle.UnpicklingError as e:
        if os.environ.get("MAGMA_UNSAFE_UNPICKLE") != "1":
            raise RuntimeError(
                f"checkpoint {path} needs a full (code-executing) unpickle: {str(e).splitlines()[0]}  Set "
                "MAGMA_UNSAFE_UNPICKLE=1 to load it the way the reference does (torch.load without weights_only) -- only "
                "for files you trust.") from e
        import warnings
        warnings.warn(f"MAGMA_UNSAFE_UNPICKLE=1: loading {path} with a full unpickle", RuntimeWarning)
        return torch.load(path, map_location=torch.device("cpu"), weights_only=False)


class Magma(nn.Module):
    def __init__(self, config, device=None, lm_config: Optional[GPTJConfig] = None, enc: nn.Module = None,
                 dtype=torch.bfloat16, init_seed: Optional[int] = None):
        super().__init__()
        if isinstance(config, (str, Path)):
            config = MultimodalConfig.from_yml(config)
        else:
            assert isinstance(config, MultimodalConfig)
        self.device = torch.device(device) if device is not None else torch.device(
            "cuda" if torch.cuda.is_available() else "cpu")
        if self.device.type != "cuda":
            from .lib import MagmaHipError
            raise MagmaHipError("magma_amd.Magma runs on MI355X only (device=%s requested): pass device='cuda:0' (the "
                                "reference README's own call, Magma.from_checkpoint(..., device='cuda:0')); there is no CPU "
                                "execution path -- the CPU restatement lives in oracle/ and is test-only" % self.device)
        # every kernel launches on the CURRENT HIP device / stream (ops._need_gpu refuses operands that live elsewhere):
        # the model's device becomes current here and again in the entry points below
        torch.cuda.set_device(self.device)
        self.device = torch.device("cuda", torch.cuda.current_device())
        self.config = config
        self.dtype = dtype
        self.lm = get_gptj(device=self.device, dtype=dtype, config=lm_config, init=True)
        if init_seed is not None:
            self.lm.init_weights(init_seed)
        self.seq_len = self.lm.config.max_position_embeddings
        self.tokenizer = get_tokenizer("gpt2", sequence_length=self.seq_len)
        self.image_token = self.tokenizer.cls_token_id
        self.eos_token = self.tokenizer.eos_token_id
        if lm_config is None:   # full-size model: len(tokenizer) = 50258
            self.lm.resize_token_embeddings(len(self.tokenizer))
        elif min(self.lm.config.vocab_size, self.lm.config.head_rows) < len(self.tokenizer):
            # reduced test vocabularies: keep the special ids in range of BOTH tables (eos = V-2, image = V-1)
            v = min(self.lm.config.vocab_size, self.lm.config.head_rows)
            self.eos_token, self.image_token = v - 2, v - 1
        self.lm.config.pad_token_id = self.tokenizer.eos_token_id
        self.word_embedding = self.lm.transformer.wte
        self.transformer = self.lm.transformer.h
        self.mlp_adapter_added, self.attn_adapter_added = False, False
        self.image_prefix = ImagePrefix(config=config, out_dim=self.lm.config.hidden_size, device=self.device,
                                        dtype=dtype, enc=enc)
        self.image_prefix_seq_len = self.image_prefix.out_seq_len
        # RGB images are resized / cropped / normalised on the GPU (bit-identical to the PIL host path of
        # reference transforms.py:121-134; MAGMA_HOST_PREPROCESS=1 keeps everything on the host)
        self.transforms = get_transforms(config.image_size, config.encoder_name,
                                         input_resolution=self.image_prefix.enc.input_resolution,
                                         device=None if os.environ.get("MAGMA_HOST_PREPROCESS") == "1" else self.device)
        if config.adapter_config:
            mlp_config = deepcopy(config.adapter_config.get("mlp", None))
            if mlp_config:
                assert mlp_config.get("adapter_type") is not None
                self.add_adapters(location="mlp", adapter_type=mlp_config.pop("adapter_type"),
                                  downsample_factor=mlp_config.pop("downsample_factor", 4), **mlp_config)
            attn_config = deepcopy(config.adapter_config.get("attention", None))
            if attn_config:
                assert attn_config.get("adapter_type") is not None
                self.add_adapters(location="attention", adapter_type=attn_config.pop("adapter_type"), **attn_config)
        # trainable set: adapters + image encoder + proj/ln; LM frozen
        if config.freeze_lm:
            for name, p in self.lm.named_parameters():
                p.requires_grad = bool(config.adapter_config) and "adapter" in name
        if config.freeze_img_encoder:
            for p in self.image_prefix.enc.parameters():
                p.requires_grad = False

    # ------------------------------------------------------------ adapters
    def add_adapters(self, downsample_factor: int = 4, adapter_type: str = "normal", location: str = "mlp",
                     ff_attr: str = "mlp", attn_attr: str = "attn", **adapter_kwargs):
        assert adapter_type in ["normal", "parallel", "scaled_parallel"], \
            "adapter_type must be one of 'normal', 'parallel', or 'scaled_parallel'"
        assert location in ["mlp", "attention"], "location must be one of 'mlp' or 'attention'"
        parallel = adapter_type in ("parallel", "scaled_parallel")
        if (location == "mlp" and self.mlp_adapter_added) or (location == "attention" and self.attn_adapter_added):
            raise ValueError("Adapter layer already added")
        dim = self.lm.config.hidden_size
        kw = dict(device=self.device, dtype=self.dtype)
        for blk in self.transformer:
            if location == "mlp" and parallel:        # reference magma.py:129-136
                setattr(blk, ff_attr, ParallelAdapter(module=getattr(blk, ff_attr), dim=dim, downsample_factor=downsample_factor,
                                                      scaled=adapter_type == "scaled_parallel", **adapter_kwargs, **kw))
            elif location == "mlp":
                adpt = Adapter(dim=dim, downsample_factor=downsample_factor, **adapter_kwargs, **kw)
                setattr(blk, ff_attr, nn.Sequential(getattr(blk, ff_attr), adpt))
            elif parallel:                            # reference magma.py:154-161
                setattr(blk, attn_attr, ParallelAdapterWrapper(module=getattr(blk, attn_attr), dim=dim,
                                                               downsample_factor=downsample_factor,
                                                               scaled="scaled" in adapter_type, **adapter_kwargs, **kw))
            else:
                setattr(blk, attn_attr, AdapterWrapper(attn_block=getattr(blk, attn_attr), dim=dim,
                                                       downsample_factor=downsample_factor, **adapter_kwargs, **kw))
        if location == "mlp":
            self.mlp_adapter_added = True
        else:
            self.attn_adapter_added = True
        self.lm.invalidate_packed()

    def invalidate_packed(self):
        from .adapters import bump_weights_epoch
        bump_weights_epoch()
        self.lm.invalidate_packed()
        self.image_prefix.invalidate_packed()

    def load_state_dict(self, state_dict, strict: bool = True):
        r = super().load_state_dict(state_dict, strict=strict)
        self.invalidate_packed()
        return r

    # -------------------------------------------------------------- inputs
    def preprocess_inputs(self, input_list: list, embed=True) -> List[torch.Tensor]:
        """Strings -> token ids, ImageInput -> normalised image tensor; mutates the
        caller's list in place exactly like the reference (magma.py:181-186)."""
        for i in range(len(input_list)):
            inp = input_list[i]
            if isinstance(inp, str):
                input_list[i] = self.tokenizer.encode(inp, return_tensors="pt")
            elif isinstance(inp, ImageInput):
                input_list[i] = inp.get_transformed_image(transform_fn=self.transforms)
            else:
                raise Exception(f"Invalid input type:{type(inp)}")
        return self.embed(input_list) if embed else input_list

    @torch.no_grad()
    def embed(self, inputs: List[torch.Tensor]) -> torch.Tensor:
        """2-D -> word embeddings, 4-D -> image prefix; written straight into one
        (b, s, d) buffer (replaces the torch.cat at reference magma.py:212)."""
        from . import ops
        torch.cuda.set_device(self.device)
        parts, total, B = [], 0, None
        for x in inputs:
            if x.ndim == 2:
                n = x.shape[1]
            elif x.ndim == 4:
                n = None
            else:
                raise ValueError(f"Expected 2d or 4d tensor, got {x.ndim}d")
            parts.append(n)
            B = x.shape[0] if B is None else B
            assert x.shape[0] == B, "all inputs must share the batch size"
        d = self.lm.config.hidden_size
        embs = []
        for x, n in zip(inputs, parts):
            if n is None:
                embs.append(self.image_prefix(x.to(self.device)))
                total += embs[-1].shape[1]
            else:
                embs.append(None)
                total += n
        out = torch.empty(B, total, d, dtype=self.dtype, device=self.device)
        off = 0
        for x, n, e in zip(inputs, parts, embs):
            if e is None:
                ops.embedding(x.to(self.device).contiguous(), self.lm.engine.wte, out, row_off=off)
                off += n
            else:
                out[:, off:off + e.shape[1]] = e
                off += e.shape[1]
        return out

    @torch.no_grad()
    def generate(self, embeddings, max_steps: int = 100, temperature: float = 0.7, top_k: int = 0,
                 top_p: float = 0.9, decode: bool = True, stop_on_eos: bool = True, seed: int = None,
                 eos_check_every: int = None):
        """reference magma.py:214-236 (+ stop_on_eos / seed / eos_check_every, see sampling.generate)."""
        torch.cuda.set_device(self.device)
        return generate(self, embeddings=embeddings, max_steps=max_steps, temperature=temperature, top_k=top_k,
                        top_p=top_p, decode=decode, stop_on_eos=stop_on_eos, seed=seed, eos_check_every=eos_check_every)

    # ------------------------------------------------------------- forward
    def forward(self, images=None, captions=None, output_hidden_states: bool = False, input_embeddings=None,
                dropout_mask=None, return_logits: bool = False) -> LMOutput:
        """reference magma.py:238-276.  ``.loss`` and ``.logits`` (B, seq_len, V) bf16 as in the reference; the loss head runs
        on the rows that carry a target and the full logits tensor (2048 x 50258 per sample) is computed when a caller first
        reads ``.logits`` (or at once with ``return_logits=True``)."""
        torch.cuda.set_device(self.device)
        assert captions is not None, "Must provide captions in training"
        assert (images is None) != (input_embeddings is None), "Pass in either images, or input embeddings, not both."
        assert captions.shape[1] == self.seq_len, \
            f"in training, captions should be padded to sequence length ({self.seq_len}), but are length {captions.shape[1]}"
        from . import ops
        captions = captions.to(self.device).contiguous()
        if input_embeddings is None:
            input_embeddings = self.image_prefix(images.to(self.device), dropout_mask=dropout_mask)
        P = input_embeddings.shape[1]
        labels = build_labels(input_embeddings, captions, self.eos_token, self.device)
        B, S = captions.shape
        emb = torch.empty(B, S, self.lm.config.hidden_size, dtype=self.dtype, device=self.device)
        emb[:, :P] = input_embeddings
        ops.embedding(captions[:, : S - P].contiguous(), self.lm.engine.wte, emb, row_off=P)
        out = self.lm(inputs_embeds=emb, labels=labels, output_hidden_states=output_hidden_states,
                      return_logits=return_logits)
        out["labels"] = labels
        return out

    # ---------------------------------------------------------- checkpoints
    @classmethod
    def from_checkpoint(cls, config_path, checkpoint_path, device="cpu", **model_kwargs):
        """Load a (DeepSpeed-layout) MAGMA checkpoint: a torch-saved dict, optionally
        wrapped in "module" (reference magma.py:278-301).  The download fallback of
        the reference needs the network and is not reproduced.

        The signature is the reference's, default ``device='cpu'`` included (magma.py:279).  This build has no CPU
        execution path, so the default RAISES ``MagmaHipError`` naming the fix: pass ``device='cuda:0'`` as the
        reference's README does.  (The checkpoint itself is always read to host memory first, as in the reference.)

        The file is read with ``weights_only=True`` first (argparse.Namespace allow-listed: DeepSpeed's
        mp_rank_00_model_states.pt carries one next to "module").  A file that still needs a full unpickle -- which can
        execute arbitrary code -- is only loaded that way with ``MAGMA_UNSAFE_UNPICKLE=1`` (what the reference's call,
        written for torch < 2.6, always did)."""
        if not exists(checkpoint_path):
            raise FileNotFoundError(f"checkpoint {checkpoint_path} does not exist (no network download in this build)")
        model = cls(config=config_path, device=device, **model_kwargs)     # model_kwargs: reduced lm_config / enc (tests)
        from .tokenizer import ByteTokenizer
        if isinstance(model.tokenizer, ByteTokenizer) and os.environ.get("MAGMA_ALLOW_BYTE_TOKENIZER") != "1":
            raise RuntimeError("from_checkpoint needs the real GPT-2 tokenizer (set MAGMA_TOKENIZER_DIR to its files): the "
                               "byte-level stand-in would feed the wrong token ids to trained weights.  "
                               "MAGMA_ALLOW_BYTE_TOKENIZER=1 overrides (synthetic checkpoints / tests).")
        sd = _load_checkpoint_file(checkpoint_path)
        if "module" in sd.keys():
            sd = sd["module"]
        print_main(f"loading magma checkpoint from: {checkpoint_path}")
        missing, unexpected = model.load_checkpoint_state(sd)
        if missing:
            print_main(f"checkpoint has no value for {len(missing)} tensors (kept at their initial values): {missing[:8]}...")
        if unexpected:
            print_main(f"checkpoint keys without a destination: {unexpected[:8]}...")
        print_main("magma successfully loaded")
        model.eval()
        return model

    def load_checkpoint_state(self, sd: dict):
        """strict=False load that tolerates the reference's aliases (SURVEY Q8:
        ``transformer.N...`` / ``word_embedding.weight`` duplicate ``lm.transformer...``),
        skips buffers of the fork (attention.bias / masked_bias) and sniffs the
        vocabulary rows of wte / lm_head (Q1)."""
        own = self.state_dict()
        fixed = {}
        for k, v in sd.items():
            if k.endswith("attention.bias") or k.endswith("masked_bias"):
                continue
            if k.startswith("transformer."):       # alias of lm.transformer.h (reference magma.py:53)
                k = "lm.transformer.h." + k[len("transformer."):]
            elif k == "word_embedding.weight":
                k = "lm.transformer.wte.weight"
            fixed[k] = v
        # Q1: the input and the output vocabulary are sniffed independently (a checkpoint may carry a 50258-row wte next
        # to a 50400-row lm_head, or the reverse of neither)
        v_in = fixed["lm.transformer.wte.weight"].shape[0] if "lm.transformer.wte.weight" in fixed else None
        v_out = fixed["lm.lm_head.weight"].shape[0] if "lm.lm_head.weight" in fixed else None
        cur_in, cur_out = own["lm.transformer.wte.weight"].shape[0], own["lm.lm_head.weight"].shape[0]
        if (v_in is not None and v_in != cur_in) or (v_out is not None and v_out != cur_out):
            self.lm.resize_token_embeddings(v_in if v_in is not None else cur_in, new_head_rows=v_out if v_out is not None else cur_out)
            self.word_embedding = self.lm.transformer.wte
            own = self.state_dict()
        missing, unexpected = [], []
        bad = [f"{k}: checkpoint {tuple(v.shape)} vs model {tuple(own[k].shape)}" for k, v in fixed.items()
               if k in own and own[k].shape != v.shape]
        if bad:     # load_state_dict(strict=False) raises on a size mismatch too (reference magma.py:298)
            raise RuntimeError("size mismatch while loading the checkpoint: " + "; ".join(bad[:6]))
        with torch.no_grad():
            for k, v in fixed.items():
                if k in own:
                    # cast on the destination device (a 28-block fp32 checkpoint converts in ~1 s on the GPU, ~40 s on the host)
                    own[k].copy_(v.to(own[k].device).to(own[k].dtype))
                else:
                    unexpected.append(k)
        # the aliases of Q8 and BatchNorm's step counters are not "missing"
        missing = [k for k in own if k not in fixed and not k.endswith("num_batches_tracked")
                   and not k.startswith(("transformer.", "word_embedding."))]
        self.invalidate_packed()
        return missing, unexpected
