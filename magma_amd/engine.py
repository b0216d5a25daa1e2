"""LMEngine -- executes the GPT-J(+adapters) graph on the HIP kernels.

One engine per GPTJForCausalLM.  It owns the device-layout copies of the
(frozen) weights -- fragment-tiled so that the same bytes feed both the
128x128 MFMA tile GEMM (prefill / training shapes) and the weight-streaming
decode GEMM -- the rotary tables, the KV cache objects and, for decode, a
HIP graph of the whole token step (28 x ~10 launches + head) that is replayed
per token with the position held in device memory.

Per block (parallel residual, adapters after attention/MLP):
    ln   = LayerNorm(x)
    qkv  = ln Wqkv^T                      -> rotary(q,k), K/V scattered into the cache
    ctx  = causal softmax(q k^T / 16) v   (flash kernel / decode kernel)
    a    = ctx Wout^T                     [v2: a += Wup relu(Wdn a + b) + b]
    h    = gelu_new(ln Wfc^T + b)
    m    = h Wproj^T + b
    x'   = m + Wup relu(Wdn m + b) + b + a + x     (one GEMM epilogue: bias + 3 residuals)
"""
from __future__ import annotations

import os
from typing import List, Optional

import torch

from . import ops
from .language_model import LMOutput

BF16 = torch.bfloat16


def rotary_tables(rotary_dim: int, n_pos: int, device):
    """sin/cos of pos * 10000^(-2i/rotary_dim) in fp32, computed with the same
    torch expression as HF/GPT-J's create_sinusoidal_positions so the table is
    bit-identical to what the reference graph multiplies by."""
    inv_freq = 1.0 / (10000 ** (torch.arange(0, rotary_dim, 2, dtype=torch.int64).float() / rotary_dim))
    ang = torch.einsum("i,j->ij", torch.arange(n_pos, dtype=torch.int64).float(), inv_freq)
    return torch.sin(ang).to(device).contiguous(), torch.cos(ang).to(device).contiguous()


class KVCache:
    """Opaque ``past_key_values`` (reference sampling.py:81-93 only hands it back)."""

    def __init__(self, n_layer: int, B: int, H: int, Smax: int, device):
        self.k = torch.empty(n_layer, B, H, Smax, 256, dtype=BF16, device=device)
        self.v = torch.empty(n_layer, B, H, Smax, 256, dtype=BF16, device=device)
        self.d_pos = torch.zeros(1, dtype=torch.int32, device=device)   # next write position
        self.pos = 0                                                     # host mirror
        # token-selection state of the generate() loop (device side): {step, first step at which every row emitted eos},
        # the seed of the sampling stream, the eos id the bookkeeping launch compares against
        self.sample_state = torch.tensor([0, -1], dtype=torch.int32, device=device)
        self.seed = torch.zeros(1, dtype=torch.int64, device=device)
        self.eos = -1
        self.history = torch.zeros(B, Smax, dtype=torch.int64, device=device)   # token selected at step s of the current loop
        self.B, self.Smax = B, Smax
        self.decode_state = None

    def __len__(self):
        return self.k.shape[0]


class _Layer:
    pass


class LMEngine:
    def __init__(self, lm):
        cfg = lm.config
        self.cfg = cfg
        dev = lm.lm_head.weight.device
        if dev.type != "cuda":
            raise ops.L.MagmaHipError("the MAGMA LM runs on MI355X only: move the model to a GPU (no CPU fallback)")
        self.device = dev
        self.d, self.H, self.L = cfg.hidden_size, cfg.num_heads, cfg.num_layers
        if self.d != self.H * 256:
            raise ValueError("the attention kernels are specialised for head_dim = 256 (GPT-J)")
        self.eps = cfg.layer_norm_epsilon
        self.wte = lm.transformer.wte.weight.detach()
        if self.wte.dtype != BF16:
            self.wte = self.wte.to(BF16)
        self.wte = self.wte.contiguous()
        self.V = lm.lm_head.weight.shape[0]
        self.layers: List[_Layer] = []
        f32 = lambda t: t.detach().float().contiguous()  # noqa: E731
        from .adapters import ParallelAdapter
        for blk in lm.transformer.h:
            ly = _Layer()
            attn = blk.attn
            ly.attn_adapter = None
            # parallel adapters (reference adapters.py:42-92) read the block INPUT (ln_1 output) instead of the wrapped
            # module's output and are scaled by adapter_scale: *_par = fp32 [d] vector holding the scale, else None
            ly.attn_par = ly.mlp_par = None
            # adapter options (reference adapters.py:11-24): *_act = epilogue code of the bottleneck activation, *_ad_ln =
            # (gamma, beta, eps) of the LayerNorm in front of the down-projection or None
            ly.attn_act = ly.mlp_act = ops.MG_ACT_RELU
            ly.attn_ad_ln = ly.mlp_ad_ln = None
            if isinstance(attn, ParallelAdapter):    # ParallelAdapterWrapper
                ly.attn_adapter, ly.attn_act, ly.attn_ad_ln = self._pack_adapter(attn)
                ly.attn_par = torch.full((self.d,), attn.scale_value(), dtype=torch.float32, device=dev)
                attn = attn.module
            elif hasattr(attn, "attn_block"):        # AdapterWrapper (v2)
                ly.attn_adapter, ly.attn_act, ly.attn_ad_ln = self._pack_adapter(attn)
                attn = attn.attn_block
            a = attn.attention
            ly.out = ops.PackedLinear(a.out_proj.weight)
            mlp = blk.mlp
            ly.mlp_adapter = None
            if isinstance(mlp, ParallelAdapter):
                ly.mlp_adapter, ly.mlp_act, ly.mlp_ad_ln = self._pack_adapter(mlp)
                ly.mlp_par = torch.full((self.d,), mlp.scale_value(), dtype=torch.float32, device=dev)
                mlp = mlp.module
            elif isinstance(mlp, torch.nn.Sequential):  # Sequential(mlp, Adapter)  (reference magma.py:143-149)
                ly.mlp_adapter, ly.mlp_act, ly.mlp_ad_ln = self._pack_adapter(mlp[1])
                mlp = mlp[0]
            # qkv and fc_in read the same LayerNorm output: ONE operand [q | k | v | fc_in] (bias: zeros | b_fc) for the fused
            # prefill GEMM; ly.qkv / ly.fc_in are row ranges of it that share its storage
            d3 = 3 * self.d
            fcb = mlp.c_fc.bias.detach().float()
            ly.in_cat = ops.PackedLinear(torch.cat([a.q_proj.weight, a.k_proj.weight, a.v_proj.weight, mlp.c_fc.weight], dim=0),
                                         bias=torch.cat([torch.zeros(d3, dtype=torch.float32, device=dev), fcb]))
            ly.qkv = ly.in_cat.rows(0, d3)
            ly.fc_in = ly.in_cat.rows(d3, ly.in_cat.N, bias=fcb.contiguous())
            ly.fc_out = ops.PackedLinear(mlp.c_proj.weight, mlp.c_proj.bias)
            ly.ln_g, ly.ln_b = f32(blk.ln_1.weight), f32(blk.ln_1.bias)
            ly.dec_in = None      # decode-only fused [qkv | fc_in] operand with ln_1 folded in (built lazily)
            ly._src = (a, mlp)
            self.layers.append(ly)
        self.lnf_g, self.lnf_b = f32(lm.transformer.ln_f.weight), f32(lm.transformer.ln_f.bias)
        self.head = ops.PackedLinear(lm.lm_head.weight, lm.lm_head.bias)
        self.Vp = ops.ceil_to(self.V, 8)
        self.sin_t, self.cos_t = rotary_tables(cfg.rotary_dim, cfg.max_position_embeddings, dev)
        self.rot = cfg.rotary_dim
        self.head_dec = None
        self._lm_head = lm.lm_head
        self._cache_pool = {}                        # (B, Smax) -> KVCache + captured decode graph, LRU-bounded
        self._cache_pool_max = int(os.environ.get("MAGMA_CACHE_POOL", "4"))
        # fp8 operands for the prefill / forward GEMMs (BASELINE config 5): None | "attn" (QKV, out_proj, adapters)
        # | "all" (+ fc_in, fc_out).  bf16 stays the default: it is what the parity tests and the headline use.
        self.fp8_mode = os.environ.get("MAGMA_FP8") or None
        self.fp8_attn = os.environ.get("MAGMA_FP8_ATTN", "1") == "1"     # with fp8_mode: QK^T / PV of the cache-less forward in e4m3 too
        # scaling of the fp8 operands: "row" = one fp32 scale per activation row / weight output channel (any tile kernel), "mx" =
        # OCP MX, one E8M0 scale per 32 K-elements of both operands, applied by the MFMA itself (both tile kernels)
        self.fp8_scaling = os.environ.get("MAGMA_FP8_SCALING", "row")
        if self.fp8_scaling not in ("row", "mx"):
            raise ValueError("MAGMA_FP8_SCALING must be 'row' or 'mx'")
        # W8A16 decode: e4m3 weights (per-output-channel scales) widened to bf16 in registers by the weight-streaming
        # GEMVs -> half the bytes per token step.  Changes the numerics (weight quantisation), so it is opt-in.
        self.decode_w8 = os.environ.get("MAGMA_DECODE_W8", "0") == "1"
        self._dec_in_variant = int(os.environ.get("MAGMA_DEC_IN_VARIANT", "0"))   # tuning knob: nt | waves<<4 | kc<<8
        self._dec_dn_variant = int(os.environ.get("MAGMA_DEC_DN_VARIANT", "0"))   # the same for the adapter-down and the
        self._dec_cat_variant = int(os.environ.get("MAGMA_DEC_CAT_VARIANT", "0"))  # [W_out | W_up] launches of the v1 block
        self._side_stream = torch.cuda.Stream(device=dev)
        self.group_launches = os.environ.get("MAGMA_DECODE_GROUPED", "1") == "1"
        # MAGMA_DECODE_FOLD -- how the adapter of a MAGMA_v1 block is laid over the block's launches (round 4):
        #   0  four launches as in rounds 1-3: [ln_1+qkv+fc_in] -> [attention || fc_out] -> [out_proj || adapter-down] -> [adapter-up]
        #   2  (default) the up-projection shares a launch with out_proj as ONE GEMV over the concatenated input [ctx | t] against
        #      [W_out | W_up]:  ... -> [attention || fc_out] -> [adapter-down] -> [[W_out | W_up]]: x' = that + b_up + m + x.  Same
        #      bytes, same launch count, 2.500 against 2.522-2.528 ms per token (the 8 MB up-projection no longer pays a
        #      launch of its own; adapter-down alone costs what the co-launch with out_proj hid).
        #   1  THREE launches: the down-projection multiplied through fc_out offline -- t = relu(W_dn m + b_dn), m = W_fc h + b_fc,
        #      so t = relu((W_dn W_fc) h + (W_dn b_fc + b_dn)), a second output segment of the fc_out launch (+25 MB per block).
        #      Parity-green (re-association only: tests/test_fullwidth_gpu.py) and SLOWER, 2.70 ms per token: the co-launch grows
        #      from 256 to 320 weight tiles of 512 KB on 256 CUs and takes 41.4 us instead of 29.6 (a quarter of the CUs stream two
        #      tiles), more than the merged [W_out | W_up] launch saves (9.7 us for 13.2 + 5.0).  profiles/r04_decode_block_variants.txt
        # The reference forms the same sum (reference adapters.py:38-39) in another association order.
        self.fold_dn = int(os.environ.get("MAGMA_DECODE_FOLD", "2"))
        self.fuse_in = os.environ.get("MAGMA_PREFILL_FUSE_IN", "1") == "1"      # [qkv | fc_in] as one prefill GEMM
        self.cat_up = os.environ.get("MAGMA_PREFILL_CAT", "1") == "1"           # out_proj + adapter-up as one GEMM over [ctx | t]
        self.two_streams = os.environ.get("MAGMA_DECODE_STREAMS", "1") == "2"   # measured slower (3.07 vs 2.94 ms/step): off

    @staticmethod
    def _pack_adapter(mod):
        """((down, up) packed, activation code, (gamma, beta, eps) | None) of an Adapter-like module, whatever its options."""
        from .adapters import activation_codes
        code, _, _ = activation_codes(mod.act)
        ln = None
        if mod.ln is not None:
            ln = (mod.ln.weight.detach().float().contiguous(), mod.ln.bias.detach().float().contiguous(), float(mod.ln.eps))
        return (ops.PackedLinear(mod.down.weight, mod.down.bias), ops.PackedLinear(mod.up.weight, mod.up.bias)), code, ln

    @staticmethod
    def _epi_act(code):
        """Epilogue code of a bottleneck activation: torch.nn.GELU() (erf) is not an epilogue -- the GEMM runs without activation
        and _act_fix applies it as its own pass."""
        return ops.MG_ACT_NONE if code == ops.MG_ACT_GELU_ERF else code

    @staticmethod
    def _act_fix(t, code):
        return ops.gelu_erf(t, out=t) if code == ops.MG_ACT_GELU_ERF else t

    @staticmethod
    def _ad_in(ln, x, out=None):
        """Input of an adapter's down-projection: x, or LayerNorm(x) for an adapter built with add_layernorm."""
        return x if ln is None else ops.layernorm(x, ln[0], ln[1], ln[2], out=out)

    def _ensure_decode_packs(self):
        """Decode operands with the LayerNorms folded in (frozen gamma/beta): per layer one
        fused [3d+ff, d] matrix W' = [Wqkv;Wfc]*gamma whose single weight-streaming launch
        produces qkv and gelu(fc_in) from the raw residual stream (no LayerNorm launch, one
        GEMV instead of two); ln_f is folded into lm_head the same way."""
        if self.head_dec is not None:
            return
        d3 = 3 * self.d
        for ly in self.layers:
            a, mlp = ly._src
            w = torch.cat([a.q_proj.weight, a.k_proj.weight, a.v_proj.weight, mlp.c_fc.weight], dim=0)
            b = torch.cat([torch.zeros(d3, device=self.device), mlp.c_fc.bias.detach().float()])
            w2, b2, cs = ops.fold_layernorm(w, b, ly.ln_g, ly.ln_b)
            lin = ops.PackedLinear(w2, bias=b2[:d3])
            lin.bias_b, lin.colsum = b2[d3:].contiguous(), cs
            ly.dec_in = lin
            del w, w2
            self._fold_adapter_down(ly)
            if self.fold_dn in (1, 2):
                self._ensure_out_up(ly)          # built HERE, never inside a token step (no allocation under hipGraph capture)
        w2, b2, cs = ops.fold_layernorm(self._lm_head.weight, self._lm_head.bias, self.lnf_g, self.lnf_b)
        self.head_dec = ops.PackedLinear(w2, bias=b2)
        self.head_dec.colsum = cs

    def _v1_block(self, ly) -> bool:
        """MAGMA_v1 block shape: mlp adapter of the 'normal' type only, every K a multiple of 128."""
        if ly.mlp_adapter is None or ly.attn_adapter is not None or ly.mlp_par is not None:
            return False
        if ly.mlp_ad_ln is not None or ly.mlp_act != ops.MG_ACT_RELU:      # the folds are written for the plain ReLU bottleneck
            return False
        dn, up = ly.mlp_adapter
        return not (dn.N % 16 or (self.d + dn.N) % 128 or ly.fc_out.Kp % 128 or dn.K != self.d or up.K != dn.N or up.Kp != up.K)

    def _ensure_out_up(self, ly):
        """[W_out | W_up] (d rows over K = d + r, bias b_up) of a MAGMA_v1 block -- the operand of the ONE GEMM / GEMV that
        replaces out_proj and the adapter's up-projection (prefill / forward blocks and the decode step) -- or None where a
        block does not have that shape.  Built with the decode operands (_ensure_decode_packs / repack_adapters) or on the first
        prefill that wants it; dropped by repack_adapters (it contains W_up).  It is a SECOND copy of W_out (42 MB per block,
        1.2 GB at 28 blocks): ly.out stays for the paths that still read it (fp8 modes, MAGMA_PREFILL_CAT=0 / MAGMA_DECODE_FOLD=0,
        attention adapters)."""
        if "out_up" not in ly.__dict__:
            ly.out_up = None
            if self._v1_block(ly):
                a, _ = ly._src
                up = ly.mlp_adapter[1]
                w_up = ops.PackedLinear.untile(up.ft)[: up.N, : up.K]
                ly.out_up = ops.PackedLinear(torch.cat([a.out_proj.weight.detach().to(BF16), w_up], dim=1), bias=up.bias)
        return ly.out_up

    def _fold_adapter_down(self, ly):
        """Decode-only operand of the three-launch block (MAGMA_DECODE_FOLD=1), or None:
          ly.fc_dn = [W_fc_out ; W_dn W_fc_out]   (d + r rows over K = ff;  bias b_fc | W_dn b_fc + b_dn)"""
        ly.fc_dn = None
        if self.fold_dn != 1 or not self._v1_block(ly):
            return
        dn, _ = ly.mlp_adapter
        _, mlp = ly._src
        w_fc, b_fc = mlp.c_proj.weight.detach().float(), mlp.c_proj.bias.detach().float()
        w_dn = ops.PackedLinear.untile(dn.ft)[: dn.N, : dn.K].float()
        ly.fc_dn = ops.PackedLinear(torch.cat([mlp.c_proj.weight.detach().to(BF16), (w_dn @ w_fc).to(BF16)], dim=0), bias=b_fc)
        ly.fc_dn.bias_b = (w_dn @ b_fc + dn.bias).contiguous()

    def _ensure_decode_packs_w8(self):
        """e4m3 copies of every decode operand (same LayerNorm folds; the fold's column sums are taken from the
        DEQUANTISED weights so that  rstd*(acc*scale - mean*colsum)  stays exact for what the kernel multiplies)."""
        if getattr(self, "head_w8", None) is not None:
            return
        d3 = 3 * self.d
        for ly in self.layers:
            a, mlp = ly._src
            w = torch.cat([a.q_proj.weight, a.k_proj.weight, a.v_proj.weight, mlp.c_fc.weight], dim=0)
            b = torch.cat([torch.zeros(d3, device=self.device), mlp.c_fc.bias.detach().float()])
            w2, b2, _ = ops.fold_layernorm(w, b, ly.ln_g, ly.ln_b)
            lin = ops.PackedLinearW8(w2, bias=b2[:d3])
            lin.bias_b, lin.colsum = b2[d3:].contiguous(), lin.dequant().sum(1).contiguous()
            w8 = _Layer()
            w8.dec_in = lin
            w8.out = ops.PackedLinearW8(a.out_proj.weight)
            w8.fc_out = ops.PackedLinearW8(mlp.c_proj.weight, mlp.c_proj.bias)
            w8.mlp_adapter = None
            if ly.mlp_adapter is not None:
                # (the up-projection alone is only used by the MAGMA_v1 step; in MAGMA_v2 -- K = 512 -- it lives in up_cat below)
                w8.mlp_adapter = tuple(ops.PackedLinearW8(ops.PackedLinear.untile(p.ft)[: p.N, : p.K], p.bias) if p.K % 1024 == 0 else None
                                       for p in ly.mlp_adapter)
            # MAGMA_v2 (attention AND mlp adapters): the attention adapter's down-projection and the concatenated up-projection
            w8.attn_adapter = w8.up_cat = None
            cat = self._adapter_up_cat(ly)
            if cat is not None and ly.attn_par is None and ly.mlp_par is None and cat.K % 1024 == 0:
                w8.attn_adapter = (ops.PackedLinearW8(ops.PackedLinear.untile(ly.attn_adapter[0].ft)[: ly.attn_adapter[0].N, : ly.attn_adapter[0].K],
                                                      ly.attn_adapter[0].bias),)
                w8.up_cat = ops.PackedLinearW8(ops.PackedLinear.untile(cat.ft)[: cat.N, : cat.K], cat.bias)
            ly.w8 = w8
            del w, w2
        w2, b2, _ = ops.fold_layernorm(self._lm_head.weight, self._lm_head.bias, self.lnf_g, self.lnf_b)
        self.head_w8 = ops.PackedLinearW8(w2, bias=b2)
        self.head_w8.colsum = self.head_w8.dequant().sum(1).contiguous()

    def repack_adapters(self, lm):
        """Refresh only the (trainable) adapter operands after optimizer steps; the
        12 GB of frozen weights keep their packed copies."""
        for ly, blk in zip(self.layers, lm.transformer.h):
            if ly.attn_adapter is not None:
                ly.attn_adapter, ly.attn_act, ly.attn_ad_ln = self._pack_adapter(blk.attn)
                if ly.attn_par is not None:           # scaled_parallel: the scale is a trained parameter too
                    ly.attn_par = torch.full((self.d,), blk.attn.scale_value(), dtype=torch.float32, device=ly.attn_par.device)
            if ly.mlp_adapter is not None:
                par = ly.mlp_par is not None
                ly.mlp_adapter, ly.mlp_act, ly.mlp_ad_ln = self._pack_adapter(blk.mlp if par else blk.mlp[1])
                if par:
                    ly.mlp_par = torch.full((self.d,), blk.mlp.scale_value(), dtype=torch.float32, device=ly.mlp_par.device)
            ly.fp8 = {}
            ly.__dict__.pop("up_cat", None)
            ly.__dict__.pop("out_up", None)        # [W_out | W_up] contains the adapter weights: rebuilt on next use
            ly.__dict__.pop("fc_dn", None)
            if self.head_dec is not None:          # decode operands exist: rebuild the folded ones now, not inside the next token step
                self._fold_adapter_down(ly)
                if self.fold_dn in (1, 2):
                    self._ensure_out_up(ly)

    @staticmethod
    def _par_up(up, par):
        """(scale vector, packed up-projection whose bias is pre-multiplied by the scale) of a parallel adapter:
        the epilogue computes acc*scale[n] + bias[n], the reference (acc + b_up) * adapter_scale."""
        key = "_par_scaled"
        sc = up.__dict__.get(key)
        if sc is None:
            sc = up.__dict__[key] = ops.PackedLinear.__new__(ops.PackedLinear)
            sc.__dict__.update(up.__dict__)
            sc.bias = None if up.bias is None else (up.bias * par[: up.N]).contiguous()
        return par, sc

    def _adapter_up_cat(self, ly):
        """[W_up_mlp | W_up_attn] along K (bias = sum) for blocks that carry both adapters, or None."""
        if ly.mlp_adapter is None or ly.attn_adapter is None or ly.mlp_ad_ln is not None or ly.attn_ad_ln is not None:
            return None
        if ops.MG_ACT_GELU_ERF in (ly.mlp_act, ly.attn_act):       # its own pass after the down-projection: generic block
            return None
        cat = ly.__dict__.get("up_cat")
        if cat is None:
            (dn_m, up_m), (dn_a, up_a) = ly.mlp_adapter, ly.attn_adapter
            ok = (dn_m.N == up_m.K == up_m.Kp and dn_a.N == up_a.K == up_a.Kp and (up_m.K + up_a.K) % 128 == 0
                  and ly.fc_out.Kp % 128 == 0 and ly.out.Kp % 128 == 0 and dn_m.Kp % 128 == 0 and dn_a.Kp % 128 == 0)
            if not ok:
                ly.up_cat = False
                return None
            w = torch.cat([ops.PackedLinear.untile(up_m.ft)[: up_m.N, : up_m.K], ops.PackedLinear.untile(up_a.ft)[: up_a.N, : up_a.K]], dim=1)
            cat = ly.up_cat = ops.PackedLinear(w, bias=up_m.bias + up_a.bias)
        return cat or None

    # ---- fp8 operand path (config 5) ----------------------------------------------------------------
    def _fp8_weight(self, ly, name: str, lin):
        """e4m3 copy (per-output-channel scales) of a packed bf16 weight, made on first use."""
        packs = ly.__dict__.setdefault("fp8", {})
        mx = self.fp8_scaling == "mx"
        w8 = packs.get(name)
        if w8 is None or isinstance(w8, ops.PackedLinearMX) != mx:
            w = ops.PackedLinear.untile(lin.ft)[: lin.N, : lin.K] if lin.ft is not None else lin.rm[: lin.N, : lin.K]
            w8 = packs[name] = (ops.PackedLinearMX if mx else ops.PackedLinearFP8)(w, lin.bias)
        return w8

    def _quantize(self, x):
        return ops.quantize_mx_fp8(x) if self.fp8_scaling == "mx" else ops.quantize_rows_fp8(x)

    def _linear(self, ly, name, lin, x, xq=None, **kw):
        """x @ lin^T through the bf16 tile GEMM, or -- when this projection is in the active fp8 set -- through the
        fp8 MFMA on a freshly quantised (per-row scale) copy of x.  ``xq`` passes an already quantised x."""
        on = self.fp8_mode == "all" or (self.fp8_mode == "attn" and name not in ("fc_in", "fc_out"))
        if not on or lin.K % 16:
            return ops.gemm(x, lin, **kw)
        q, sc = xq if xq is not None else self._quantize(x)
        if self.fp8_scaling == "mx":
            return ops.gemm_mx_fp8(q, sc, self._fp8_weight(ly, name, lin), **kw)
        return ops.gemm_fp8(q, sc, self._fp8_weight(ly, name, lin), **kw)

    # ------------------------------------------------------------------ API
    def forward(self, input_ids=None, inputs_embeds=None, labels=None, use_cache=False, past_key_values=None,
                output_hidden_states=False, cache_hint: Optional[int] = None, reuse_cache: bool = False,
                return_logits: bool = False, sampling=None, eos_token: Optional[int] = None,
                seed: Optional[int] = None, feed_back: bool = False) -> LMOutput:
        if labels is not None:
            if inputs_embeds is None:
                inputs_embeds = self.embed_ids(input_ids)
            return self.forward_loss(inputs_embeds, labels, output_hidden_states, return_logits)
        if past_key_values is not None:
            if not feed_back and input_ids is None:
                raise ValueError("cached decoding takes input_ids (reference sampling.py:88-90)")
            if not feed_back and input_ids.shape[1] != 1:
                # several new tokens against the cache (the reference LM takes any input_ids length with past_key_values; its
                # generate() only ever sends one, sampling.py:86-90): the causal mask makes this exactly T single-token steps in
                # order, each appending its K / V -- run as such, logits (B, T, V) stacked
                T = input_ids.shape[1]
                if T == 0:
                    raise ValueError("cached decoding needs at least one new token")
                rows = []
                for i in range(T):      # only the LAST position selects a token (history / RNG step / eos latch untouched before)
                    lg, tok = self.decode(input_ids[:, i:i + 1], past_key_values, sampling=sampling, select=i == T - 1)
                    rows.append(lg.clone())
                return LMOutput(logits=torch.stack(rows, 1), past_key_values=past_key_values, next_token=tok, loss=None,
                                eos_state=past_key_values.sample_state)
            logits, tok = self.decode(None if feed_back else input_ids, past_key_values, sampling=sampling)
            return LMOutput(logits=logits.unsqueeze(1), past_key_values=past_key_values, next_token=tok, loss=None,
                            eos_state=past_key_values.sample_state)
        if inputs_embeds is None:
            inputs_embeds = self.embed_ids(input_ids)
        if use_cache:
            logits, cache, hs = self.prefill(inputs_embeds, cache_hint, output_hidden_states, reuse_cache)
            # generate() only reads the last position, so only that row is computed
            out = LMOutput(logits=logits.unsqueeze(1), past_key_values=cache, hidden_states=hs, loss=None)
            if eos_token is not None:     # generate(): first token of the loop selected here, device-side bookkeeping armed
                if cache.eos != int(eos_token):        # the eos id is a launch argument of the captured bookkeeping kernel
                    cache.eos = int(eos_token)
                    if cache.decode_state is not None:
                        cache.decode_state.graphs.clear()
                cache.sample_state.copy_(torch.tensor([0, -1], dtype=torch.int32), non_blocking=True)
                if seed is not None:
                    cache.seed.fill_(int(seed) & 0x7fffffffffffffff)
                st = self._ensure_decode_state(cache)      # the first token lands where the decode steps read it back
                out["next_token"] = self.select_token(logits, cache, sampling, out=st.token)
                out["eos_state"] = cache.sample_state
            return out
        x, hs = self._blocks_prefill(inputs_embeds, None, output_hidden_states)
        B, S, _ = inputs_embeds.shape
        logits = self._full_logits(x, B * S).view(B, S, self.V)
        return LMOutput(logits=logits, past_key_values=None, hidden_states=hs, loss=None)

    def embed_ids(self, ids: torch.Tensor) -> torch.Tensor:
        ids = ids.to(self.device).contiguous()
        out = torch.empty(ids.shape[0], ids.shape[1], self.d, dtype=BF16, device=self.device)
        return ops.embedding(ids, self.wte, out)

    # -------------------------------------------------------------- prefill
    def _blocks_prefill(self, embeds: torch.Tensor, cache: Optional[KVCache], want_hidden=False, lse_out=None):
        B, S, d = embeds.shape
        assert d == self.d
        if S > self.cfg.max_position_embeddings:
            raise ValueError(f"sequence length {S} exceeds max_position_embeddings")
        dev = self.device
        x = embeds.to(BF16).contiguous().view(B * S, d)
        M = B * S
        vt_ld = ops.ceil_to(S, 32)
        q = torch.empty(B, self.H, S, 256, dtype=BF16, device=dev)
        vt = torch.empty(B, self.H, vt_ld // 32, 256, 32, dtype=BF16, device=dev)   # V^T in 32-key tiles
        if cache is None:   # no cache requested: one scratch K/V shared by all layers
            kscr = torch.empty(B, self.H, S, 256, dtype=BF16, device=dev)
            vscr = torch.empty(B, self.H, S, 256, dtype=BF16, device=dev)
        # MAGMA_v1 blocks (mlp adapter of the 'normal' type): out_proj and the adapter's up-projection are ONE GEMM over the
        # concatenated input [ctx | t] against [W_out | W_up] -- x' = that + b_up + m + x, the sum the reference forms (reference
        # adapters.py:38-39 + the block's residual) in another association order.  One launch, one epilogue and one (M, d)
        # round trip (the attention output `a`) less per block; at M = 456 also one split-K fix-up less.  MAGMA_PREFILL_CAT=0: off.
        r_cat = 0
        if self.cat_up and not self.fp8_mode:
            r_cat = max([ly.mlp_adapter[0].N for ly in self.layers if ly.mlp_adapter is not None and ly.attn_adapter is None
                         and ly.mlp_par is None and self._ensure_out_up(ly) is not None] + [0])
        ctx_t = torch.empty(M, d + r_cat, dtype=BF16, device=dev)
        ctx = ctx_t[:, :d]
        hs = [x.view(B, S, d)] if want_hidden else None
        for li, ly in enumerate(self.layers):
            ln = ops.layernorm(x, ly.ln_g, ly.ln_b, self.eps)
            lnq = self._quantize(ln) if self.fp8_mode else None             # shared by qkv (and fc_in in "all" mode)
            h_fused = None
            if self.fuse_in and not self.fp8_mode:
                # qkv and gelu(fc_in) in ONE launch over [q | k | v | fc_in] (gelu_new on the fc_in columns only).  At the
                # BASELINE prefill (M = 8 x 57 = 456 rows) the 256x256 kernel then covers the 28 672 columns with
                # 2 x 112 = 224 tiles -- one round of the 256 CUs, no split-K fix-up -- instead of 384 + 512 tiles of 128^2
                # in two under-filled launches
                d3 = 3 * self.d
                qh = ops.gemm(ln, ly.in_cat, act=ops.MG_ACT_GELU_NEW, act_n0=d3, tile=256 if 256 < M <= 512 else 0)
                qkv, h_fused = qh[:, :d3], qh[:, d3:]
            else:
                qkv = self._linear(ly, "qkv", ly.qkv, ln, lnq)
            kc, vc = (cache.k[li], cache.v[li]) if cache is not None else (kscr, vscr)
            if self.fp8_mode and self.fp8_attn and cache is None and qkv.is_contiguous():
                # BASELINE config[4]: the attention core on the fp8 MFMA as well (no KV cache to fill: cached decoding reads bf16)
                a8 = ops.rotary_split_fp8(qkv, B, S, self.H, self.rot, self.sin_t, self.cos_t)
                ops.attn_prefill_fp8(a8, ctx, lse=None if lse_out is None else lse_out[li])
            else:
                ops.rotary_split(qkv, B, S, self.H, self.rot, self.sin_t, self.cos_t, q, kc, vc, pos0=0, vt=vt)
                ops.attn_prefill(q, kc, vt, ctx, B, self.H, S, lse=None if lse_out is None else lse_out[li])
            if r_cat and ly.__dict__.get("out_up") is not None and ly.mlp_adapter[0].N == r_cat:
                h = h_fused if h_fused is not None else self._linear(ly, "fc_in", ly.fc_in, ln, lnq, act=ops.MG_ACT_GELU_NEW)
                m = ops.gemm(h, ly.fc_out)
                ops.gemm(m, ly.mlp_adapter[0], out=ctx_t[:, d:], act=ops.MG_ACT_RELU)
                x = ops.gemm(ctx_t, ly.out_up, residuals=(m, x))
                if want_hidden:
                    hs.append(x.view(B, S, d))
                continue
            a = self._linear(ly, "out", ly.out, ctx)
            if ly.attn_adapter is not None and ly.attn_par is not None:      # parallel: adapter reads the attention INPUT
                sc, up = self._par_up(ly.attn_adapter[1], ly.attn_par)
                t = self._act_fix(self._linear(ly, "attn_dn", ly.attn_adapter[0], self._ad_in(ly.attn_ad_ln, ln), act=self._epi_act(ly.attn_act)), ly.attn_act)
                a = ops.gemm(t, up, scale=sc, residuals=(a,))
            elif ly.attn_adapter is not None:
                t = self._act_fix(self._linear(ly, "attn_dn", ly.attn_adapter[0], self._ad_in(ly.attn_ad_ln, a), act=self._epi_act(ly.attn_act)), ly.attn_act)
                a = self._linear(ly, "attn_up", ly.attn_adapter[1], t, residuals=(a,))
            h = h_fused if h_fused is not None else self._linear(ly, "fc_in", ly.fc_in, ln, lnq, act=ops.MG_ACT_GELU_NEW)
            if ly.mlp_adapter is not None and ly.mlp_par is not None:        # parallel: adapter reads the MLP INPUT
                sc, up = self._par_up(ly.mlp_adapter[1], ly.mlp_par)
                m = self._linear(ly, "fc_out", ly.fc_out, h)
                t = self._act_fix(self._linear(ly, "mlp_dn", ly.mlp_adapter[0], self._ad_in(ly.mlp_ad_ln, ln), act=self._epi_act(ly.mlp_act)), ly.mlp_act)
                x = ops.gemm(t, up, scale=sc, residuals=(m, a, x))
            elif ly.mlp_adapter is not None:
                m = self._linear(ly, "fc_out", ly.fc_out, h)
                t = self._act_fix(self._linear(ly, "mlp_dn", ly.mlp_adapter[0], self._ad_in(ly.mlp_ad_ln, m), act=self._epi_act(ly.mlp_act)), ly.mlp_act)
                x = self._linear(ly, "mlp_up", ly.mlp_adapter[1], t, residuals=(m, a, x))
            else:
                x = self._linear(ly, "fc_out", ly.fc_out, h, residuals=(a, x))
            if want_hidden:
                hs.append(x.view(B, S, d))
        return x, hs

    def prefill(self, embeds: torch.Tensor, cache_hint: Optional[int] = None, want_hidden=False,
                reuse_cache: bool = False):
        B, S, _ = embeds.shape
        n_pos = self.cfg.max_position_embeddings
        Smax = min(n_pos, ops.ceil_to(S + (cache_hint if cache_hint else 256), 64))
        if reuse_cache:
            # generate() owns the cache for the duration of one call: keep one KV cache (and the
            # hipGraph of the token step captured on it) per shape instead of re-allocating and
            # re-capturing for every call
            cache = self._cache_pool.pop((B, Smax), None)
            if cache is None:
                while len(self._cache_pool) >= self._cache_pool_max:      # least recently used shape goes first
                    self._cache_pool.pop(next(iter(self._cache_pool)))
                cache = KVCache(self.L, B, self.H, Smax, self.device)
            self._cache_pool[(B, Smax)] = cache                          # (re-)insert as most recently used
        else:
            cache = KVCache(self.L, B, self.H, Smax, self.device)
        x, hs = self._blocks_prefill(embeds, cache, want_hidden)
        cache.pos = S
        cache.d_pos.fill_(S)
        last = x.view(B, S, self.d)[:, S - 1, :]                 # strided rows, no copy
        xl = ops.layernorm(last, self.lnf_g, self.lnf_b, self.eps)
        logits = self._head(xl)
        return logits, cache, hs

    def _head(self, xl: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        M = xl.shape[0]
        if out is None:
            out = torch.empty(M, self.Vp, dtype=torch.float32, device=self.device)
        if M <= 16:
            ops.gemm_skinny(xl, self.head, out=out)
        else:
            ops.gemm(xl, self.head, out=out)
        return out[:, : self.V]

    def _full_logits(self, x: torch.Tensor, M: int) -> torch.Tensor:
        xl = ops.layernorm(x, self.lnf_g, self.lnf_b, self.eps)
        out = torch.empty(M, self.Vp, dtype=BF16, device=self.device)
        ops.gemm(xl, self.head, out=out)
        return out[:, : self.V]

    # --------------------------------------------------------------- decode
    def _alloc_decode_state(self, cache: KVCache):
        B, d, dev = cache.B, self.d, self.device
        ff = self.layers[0].fc_in.N
        st = _Layer()
        e = lambda *s, dt=BF16: torch.empty(*s, dtype=dt, device=dev)  # noqa: E731
        st.ids = torch.zeros(B, 1, dtype=torch.int64, device=dev)
        st.xa, st.xb = e(B, d), e(B, d)
        st.ln, st.qkv, st.q = e(B, d), e(B, 3 * d), e(B, self.H, 1, 256)
        st.ctx, st.a, st.a2, st.h, st.m = e(B, d), e(B, d), e(B, d), e(B, ff), e(B, d)
        r_mlp = max([ly.mlp_adapter[0].N for ly in self.layers if ly.mlp_adapter] + [8])
        r_att = max([ly.attn_adapter[0].N for ly in self.layers if ly.attn_adapter] + [8])
        st.t, st.ta = e(B, r_mlp), e(B, r_att)
        st.tcat = e(B, r_mlp + r_att)      # [mlp bottleneck | attention bottleneck] side by side (fused up-projection)
        st.ctx_t = e(B, d + r_mlp)         # [attention context | mlp bottleneck]: input of the [W_out | W_up] GEMV (fold_dn)
        st.lnf = e(B, d)
        st.ad_ln = e(B, d)                 # LayerNorm output in front of an adapter built with add_layernorm
        st.logits = e(B, self.Vp, dt=torch.float32)
        st.token = torch.zeros(B, dtype=torch.int64, device=dev)
        st.graphs = {}             # token-selection mode (None = greedy | (temperature, top_k, top_p)) -> captured hipGraph
        st.steps = 0
        return st

    def select_token(self, logits: torch.Tensor, cache: KVCache, mode, out: Optional[torch.Tensor] = None,
                     advance: bool = False, clear: Optional[torch.Tensor] = None) -> torch.Tensor:
        """next token of every row from fp32 logits (B, V): greedy argmax (mode None; reference sampling.py:96-97) or the
        sampled branch (mode = (temperature, top_k, top_p); :99-107), then the loop bookkeeping in one small launch
        (all-eos step, step counter, token history, and -- inside a decode step -- the KV write position).  Enqueue-only:
        used inside the captured token step and, eagerly, on the prefill logits."""
        if mode is None:
            tok = ops.argmax(logits, out=out)
        else:
            tok = ops.sample(logits, mode[0], mode[1], mode[2], cache.seed, cache.sample_state, out=out)
        ops.sample_finish(tok, cache.eos, cache.sample_state, d_pos=cache.d_pos if advance else None, history=cache.history,
                          clear=clear, clear_stride=16 if clear is not None else 1)
        return tok

    def check_decode(self, cache: KVCache):
        """Kept for callers of earlier revisions (the in-launch hand-off experiments had a time-out flag to read here)."""
        return None

    def _decode_step(self, cache: KVCache, st, mode=None, feed_back: bool = False):
        """Enqueue one token step for all B sequences (graph-capturable: no
        allocation, no sync, position read from cache.d_pos on the device).
        ``feed_back``: the input ids are the tokens the previous step selected (st.token, still on the device) -- the
        reference's loop feeds exactly those back (sampling.py:88-90) -- instead of ids copied in from the caller."""
        B = cache.B
        ops.embedding(st.token.view(B, 1) if feed_back else st.ids, self.wte, st.xa.view(B, 1, self.d))
        x, xn = st.xa, st.xb
        d3 = 3 * self.d
        main = torch.cuda.current_stream()
        side = self._side_stream if self.two_streams else None
        w8_on = self.decode_w8
        # B > 16 (the weight-streaming GEMV kernels take M <= 16): the same block, every projection through the tile GEMM
        # on the prefill operands -- weights are still read once per step for the whole batch (reference sampling.py:43-121
        # has no batch limit).  LayerNorm is a launch of its own there (the fold lives in the GEMV kernel).
        wide = B > 16
        G = ops.gemm if wide else ops.gemm_skinny
        if wide and w8_on:
            raise NotImplementedError("W8A16 decode covers batches of at most 16 sequences")
        for li, ly in enumerate(self.layers):
            src = ly.w8 if w8_on else ly                 # e4m3 or bf16 operands (same launches)
            if wide:
                ops.layernorm(x, ly.ln_g, ly.ln_b, self.eps, out=st.ln)
                ops.gemm(st.ln, ly.qkv, out=st.qkv)
                ops.gemm(st.ln, ly.fc_in, out=st.h, act=ops.MG_ACT_GELU_NEW)
            else:
                # ln_1 + qkv + fc_in(+gelu) in ONE weight-streaming launch
                ops.gemm_skinny(x, src.dec_in, out=st.qkv, ln_fold=(src.dec_in.colsum, self.d, self.eps),
                                split=(d3, st.h, ops.MG_ACT_GELU_NEW, src.dec_in.bias_b), variant=self._dec_in_variant)
            # attention branch (latency-bound, 128 workgroups) runs on a second HIP stream
            # underneath the MLP branch's weight streaming; both join at the adapter-up GEMV
            par = ly.mlp_par is not None or ly.attn_par is not None
            grouped = (not wide and self.group_launches and not par and ly.mlp_adapter is not None and ly.attn_adapter is None
                       and ly.mlp_ad_ln is None and ly.mlp_act != ops.MG_ACT_GELU_ERF and ly.fc_out.Kp % 128 == 0 and ly.out.Kp % 128 == 0 and ly.mlp_adapter[0].Kp % 128 == 0)
            out_up = self._ensure_out_up(ly) if (grouped and not w8_on and self.fold_dn in (1, 2)) else None
            if out_up is not None and self.fold_dn == 2:
                # MAGMA_DECODE_FOLD=2: attention || fc_out (context row lands in st.ctx_t), adapter-down alone, [W_out | W_up] GEMV
                r = ly.mlp_adapter[0].N
                ctx, t = st.ctx_t[:, : self.d], st.ctx_t[:, self.d: self.d + r]
                ops.decode_attn_gemv(st.qkv, cache.k[li], cache.v[li], ctx, B, self.H, cache.d_pos, self.rot, self.sin_t, self.cos_t,
                                     (st.h, ly.fc_out, st.m, {}))
                ops.gemm_skinny(st.m, ly.mlp_adapter[0], out=t, act=ops.MG_ACT_RELU, variant=self._dec_dn_variant)
                ops.gemm_skinny(st.ctx_t[:, : self.d + r], ly.out_up, out=xn, residuals=(st.m, x), variant=self._dec_cat_variant)
                x, xn = xn, x
                continue
            if out_up is not None and self.fold_dn == 1 and getattr(ly, "fc_dn", None) is not None:
                # three launches (fold_dn).  launch 2: attention || [fc_out ; W_dn W_fc_out]: m and the adapter bottleneck t
                # from ONE pass over h; the context row lands beside t in st.ctx_t
                r = ly.mlp_adapter[0].N
                ctx, t = st.ctx_t[:, : self.d], st.ctx_t[:, self.d: self.d + r]
                ops.decode_attn_gemv(st.qkv, cache.k[li], cache.v[li], ctx, B, self.H, cache.d_pos, self.rot, self.sin_t, self.cos_t,
                                     (st.h, ly.fc_dn, st.m, {"split": (self.d, t, ops.MG_ACT_RELU, ly.fc_dn.bias_b)}))
                # launch 3: x' = [W_out | W_up] [ctx ; t] + b_up + m + x
                ops.gemm_skinny(st.ctx_t[:, : self.d + r], ly.out_up, out=xn, residuals=(st.m, x))
                x, xn = xn, x
                continue
            if grouped and w8_on and (src.mlp_adapter is None or src.mlp_adapter[0] is None or src.mlp_adapter[1] is None):
                # an adapter projection whose K is not a multiple of 1024 has no e4m3 operand (_ensure_decode_packs_w8)
                raise NotImplementedError("W8A16 decode needs adapter projections with K % 1024 == 0 (downsample_factor 4 at d = 4096)")
            if grouped:
                # launch 2: attention workgroups + fc_out GEMV workgroups in one grid (they are independent
                # branches of the parallel block; the latency-bound attention hides under the weight stream)
                ops.decode_attn_gemv(st.qkv, cache.k[li], cache.v[li], st.ctx, B, self.H, cache.d_pos, self.rot,
                                     self.sin_t, self.cos_t, (st.h, src.fc_out, st.m, {}))
                # launch 3: out_proj || adapter-down
                t = st.t[:, : ly.mlp_adapter[0].N]
                ops.gemm_skinny2((st.ctx, src.out, st.a, {}), (st.m, src.mlp_adapter[0], t, {"act": ly.mlp_act}))
                # launch 4: adapter-up + the block's three residuals
                ops.gemm_skinny(t, src.mlp_adapter[1], out=xn, residuals=(st.m, st.a, x))
                x, xn = xn, x
                continue
            up_cat = self._adapter_up_cat(ly) if self.group_launches and not par and not wide else None
            if w8_on and (up_cat is None or src.up_cat is None or src.mlp_adapter[0] is None):
                raise NotImplementedError("W8A16 decode covers the grouped MAGMA_v1 and MAGMA_v2 steps ('normal' adapters, K % 1024 == 0) only")
            if up_cat is not None:
                # MAGMA_v2 (attention AND mlp adapters): 5 launches.  x' = up_m(t) + up_a(ta) + m + a + x is ONE GEMV over
                # the concatenated bottlenecks [t | ta] against [W_up_m | W_up_a] (the adapter outputs only ever appear summed).
                r1 = ly.mlp_adapter[0].N
                t, ta = st.tcat[:, :r1], st.tcat[:, r1: r1 + ly.attn_adapter[0].N]
                if w8_on:
                    up_cat = src.up_cat
                ops.decode_attn_gemv(st.qkv, cache.k[li], cache.v[li], st.ctx, B, self.H, cache.d_pos, self.rot,
                                     self.sin_t, self.cos_t, (st.h, src.fc_out, st.m, {}))
                ops.gemm_skinny2((st.ctx, src.out, st.a, {}), (st.m, src.mlp_adapter[0], t, {"act": ly.mlp_act}))
                ops.gemm_skinny(st.a, src.attn_adapter[0], out=ta, act=ly.attn_act)
                ops.gemm_skinny(st.tcat[:, : up_cat.Kp], up_cat, out=xn, residuals=(st.m, st.a, x))
                x, xn = xn, x
                continue
            if side is not None:
                side.wait_stream(main)
                torch.cuda.set_stream(side)
            ops.attn_decode_fused(st.qkv, cache.k[li], cache.v[li], st.ctx, B, self.H, cache.d_pos, self.rot,
                                  self.sin_t, self.cos_t)
            a = G(st.ctx, ly.out, out=st.a)
            if par and not wide:      # parallel adapters read ln_1(x): the one decode configuration that needs the LayerNorm as a tensor
                ops.layernorm(x, ly.ln_g, ly.ln_b, self.eps, out=st.ln)
            if ly.attn_adapter is not None:
                ta = st.ta[:, : ly.attn_adapter[0].N]
                if ly.attn_par is not None:
                    sc, up = self._par_up(ly.attn_adapter[1], ly.attn_par)
                    self._act_fix(G(self._ad_in(ly.attn_ad_ln, st.ln, out=st.ad_ln), ly.attn_adapter[0], out=ta, act=self._epi_act(ly.attn_act)), ly.attn_act)
                    a = G(ta, up, out=st.a2, scale=sc, residuals=(a,))
                else:
                    self._act_fix(G(self._ad_in(ly.attn_ad_ln, a, out=st.ad_ln), ly.attn_adapter[0], out=ta, act=self._epi_act(ly.attn_act)), ly.attn_act)
                    a = G(ta, ly.attn_adapter[1], out=st.a2, residuals=(a,))
            if side is not None:
                torch.cuda.set_stream(main)
            if ly.mlp_adapter is not None:
                G(st.h, ly.fc_out, out=st.m)
                t = st.t[:, : ly.mlp_adapter[0].N]
                if side is not None:
                    main.wait_stream(side)
                if ly.mlp_par is not None:
                    sc, up = self._par_up(ly.mlp_adapter[1], ly.mlp_par)
                    self._act_fix(G(self._ad_in(ly.mlp_ad_ln, st.ln, out=st.ad_ln), ly.mlp_adapter[0], out=t, act=self._epi_act(ly.mlp_act)), ly.mlp_act)
                    G(t, up, out=xn, scale=sc, residuals=(st.m, a, x))
                else:
                    self._act_fix(G(self._ad_in(ly.mlp_ad_ln, st.m, out=st.ad_ln), ly.mlp_adapter[0], out=t, act=self._epi_act(ly.mlp_act)), ly.mlp_act)
                    G(t, ly.mlp_adapter[1], out=xn, residuals=(st.m, a, x))
            else:
                if side is not None:
                    main.wait_stream(side)
                G(st.h, ly.fc_out, out=xn, residuals=(a, x))
            x, xn = xn, x
        if wide:
            ops.layernorm(x, self.lnf_g, self.lnf_b, self.eps, out=st.lnf)
            ops.gemm(st.lnf, self.head, out=st.logits)
        else:
            head = self.head_w8 if w8_on else self.head_dec
            ops.gemm_skinny(x, head, out=st.logits, ln_fold=(head.colsum, self.d, self.eps))
        if mode == "noselect":
            ops.advance_pos(cache.d_pos)          # teacher-forced position: nothing selected, nothing recorded
        else:
            self.select_token(st.logits[:, : self.V], cache, mode, out=st.token, advance=True)

    def _ensure_decode_state(self, cache: KVCache):
        st = cache.decode_state
        if st is None:
            if cache.B <= 16:           # larger batches run the tile GEMM on the prefill operands (no LayerNorm-folded packs)
                self._ensure_decode_packs()
            if self.decode_w8:
                self._ensure_decode_packs_w8()
            st = cache.decode_state = self._alloc_decode_state(cache)
        return st

    def decode(self, input_ids: Optional[torch.Tensor], cache: KVCache, use_graph: bool = True, sampling=None, select: bool = True):
        """One cached step.  Returns (fp32 logits (B,V) view, selected token (B,) view: greedy, or sampled when
        ``sampling = (temperature, top_k, top_p)``); both are overwritten by the next step.  ``input_ids=None`` feeds the
        previously selected tokens back without leaving the device.  ``select=False`` (teacher-forced positions of a
        multi-token call): no token is selected -- the history, the RNG step counter and the all-eos latch are left alone,
        only the KV write position advances; the returned token view is stale."""
        if cache.pos >= cache.Smax:
            raise ValueError(f"KV cache full (Smax={cache.Smax}); pass a larger cache_hint / max_steps")
        if cache.B > 16:
            use_graph = False       # the tile-GEMM step of large batches is launched eagerly (split-K scratch is per stream)
            if not getattr(self, "_warned_wide", False):
                self._warned_wide = True
                import warnings
                warnings.warn(f"decode batch {cache.B} > 16: the token step runs every projection through the tile GEMM, launched "
                              "eagerly with a separate LayerNorm (correct, tested) -- the weight-streaming GEMVs, the fused "
                              "launches and the captured hipGraph of the B <= 16 step do not apply, and W8A16 decode is refused",
                              RuntimeWarning, stacklevel=2)
        st = self._ensure_decode_state(cache)
        feed_back = input_ids is None
        if not feed_back:
            st.ids.copy_(input_ids.reshape(cache.B, 1))
        mode = None if sampling is None else (float(sampling[0]), int(sampling[1]), float(sampling[2]))
        if not select:
            if feed_back:
                raise ValueError("decode(select=False) needs input_ids: there is no selected token to feed back")
            mode = "noselect"
        key = (mode, feed_back)
        if not use_graph:
            self._decode_step(cache, st, mode, feed_back)
        elif key in st.graphs:
            st.graphs[key].replay()
        elif st.steps == 0:
            self._decode_step(cache, st, mode, feed_back)    # first step eager (loads code objects)
        else:
            g = torch.cuda.CUDAGraph()            # hipGraph on ROCm
            with torch.cuda.graph(g):
                self._decode_step(cache, st, mode, feed_back)
            st.graphs[key] = g
            g.replay()
        st.steps += 1
        cache.pos += 1
        return st.logits[:, : self.V], st.token

    # ------------------------------------------------- loss (eval) forward
    def forward_loss(self, embeds: torch.Tensor, labels: torch.Tensor, want_hidden=False, want_logits=False) -> LMOutput:
        """Shifted cross-entropy of the full sequence (reference magma.py:270-274).
        lm_head + CE are evaluated only on rows that carry a target (the others
        contribute nothing to the loss); inference-mode forward, see train engine
        for the autograd version."""
        B, S, _ = embeds.shape
        x, hs = self._blocks_prefill(embeds, None, want_hidden)
        labels = labels.to(self.device)
        tgt = labels[:, 1:].reshape(-1)
        rows = (torch.arange(B, device=self.device)[:, None] * S + torch.arange(S - 1, device=self.device)[None, :]).reshape(-1)
        keep = (tgt != -100).nonzero().squeeze(1)           # host sync: index plumbing only
        if keep.numel() == 0:
            return LMOutput(loss=torch.full((), float("nan"), device=self.device), logits=None, hidden_states=hs,
                            past_key_values=None)
        xr = x.index_select(0, rows[keep])
        xl = ops.layernorm(xr, self.lnf_g, self.lnf_b, self.eps)
        logits = torch.empty(xl.shape[0], self.Vp, dtype=torch.float32, device=self.device)
        ops.gemm(xl, self.head, out=logits)
        loss, _ = ops.cross_entropy(logits[:, : self.V], tgt[keep].contiguous())
        # reference magma.py:270-276 .logits: (B, S, V) over every position -- on request now, otherwise on first access
        full = (self._full_logits(x, B * S).view(B, S, self.V) if want_logits
                else LMOutput.lazy(lambda: self._full_logits(x, B * S).view(B, S, self.V)))
        return LMOutput(loss=loss, logits=full, hidden_states=hs, past_key_values=None,
                        target_rows=rows[keep], target_logits=logits[:, : self.V])
