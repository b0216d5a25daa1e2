"""Tensor-level wrappers over the C ABI.  PyTorch is used for device memory,
streams and one-off weight re-layout only; every hot-path computation below is
a call into libmagma_hip.so.  No CPU path exists: tensors must be on a GPU."""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence

import torch

from . import lib as L
from .lib import (MG_A_CONV3X3, MG_A_DENSE, MG_ACT_GELU_NEW, MG_ACT_NONE, MG_ACT_RELU,
                  MG_W_FRAGTILED, MG_W_ROWMAJOR, Epilogue, GemmDesc, SkinnyDesc, check)

BF16 = torch.bfloat16
_zero_pages = {}


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _need_gpu(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise L.MagmaHipError(
                "magma_amd ops run on MI355X only (tensor on %s); there is no CPU fallback" % t.device)


def _p(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def zero_page(device) -> torch.Tensor:
    key = str(device)
    if key not in _zero_pages:
        _zero_pages[key] = torch.zeros(256, dtype=BF16, device=device)
    return _zero_pages[key]


def ceil_to(x: int, m: int) -> int:
    return (x + m - 1) // m * m


class PackedLinear:
    """A [N, K] weight in the layouts the GEMM kernels consume.

    ``ft``: fragment-tiled [ceil(N/16)][Kp/32][64 lanes][8] (MG_W_FRAGTILED): one
            wave instruction reads 1 KiB contiguous == one MFMA operand fragment.
            Used by the decode (weight-streaming) kernel and by the tile GEMM.
    ``rm``: row-major [N, Kp], K zero-padded to a multiple of 64 (trainable
            weights that change every step keep this cheap layout).
    """

    def __init__(self, weight: torch.Tensor, bias: Optional[torch.Tensor] = None,
                 tiled: bool = True, rowmajor: bool = False):
        _need_gpu(weight)
        assert weight.ndim == 2
        self.N, self.K = weight.shape
        if self.K % 8:
            raise ValueError("K must be a multiple of 8")
        self.Kp = ceil_to(self.K, 64)
        self.device = weight.device
        self.ft = None
        self.rm = None
        self.bias = None if bias is None else bias.detach().to(torch.float32).contiguous()
        w = weight.detach().to(BF16)
        if rowmajor:
            self.rm = self._pad(w, self.N, self.Kp)
        if tiled:
            self.ft = self.tile(self._pad(w, ceil_to(self.N, 16), self.Kp))

    @staticmethod
    def _pad(w, n, k):
        if w.shape == (n, k):
            return w.contiguous()
        out = torch.zeros(n, k, dtype=BF16, device=w.device)
        out[: w.shape[0], : w.shape[1]] = w
        return out

    @staticmethod
    def tile(w: torch.Tensor) -> torch.Tensor:
        n, k = w.shape
        return w.view(n // 16, 16, k // 32, 4, 8).permute(0, 2, 3, 1, 4).contiguous()

    @staticmethod
    def untile(ft: torch.Tensor) -> torch.Tensor:
        nt, ks = ft.shape[0], ft.shape[1]
        return ft.view(nt, ks, 4, 16, 8).permute(0, 3, 1, 2, 4).reshape(nt * 16, ks * 32)


def _epilogue(out: torch.Tensor, N: int, bias, scale, act, residuals, act_after) -> Epilogue:
    ep = Epilogue()
    ep.scale = _p(scale)
    ep.bias = _p(bias)
    ep.act = act
    ep.act_after = act_after
    res = list(residuals) + [None] * (3 - len(residuals))
    ldr = 0
    for r in residuals:
        _need_gpu(r)
        assert r.dtype == BF16 and r.ndim == 2 and r.stride(1) == 1 and r.shape[1] >= N
        ldr = ldr or r.stride(0)
        assert r.stride(0) == ldr, "all residuals must share a row stride"
    ep.res0, ep.res1, ep.res2 = _p(res[0]), _p(res[1]), _p(res[2])
    ep.ldr = ldr
    ep.C = out.data_ptr()
    ep.ldc = out.stride(0)
    ep.out_f32 = 1 if out.dtype == torch.float32 else 0
    return ep


def gemm(a: torch.Tensor, w: PackedLinear, out: Optional[torch.Tensor] = None, *, act: int = MG_ACT_NONE,
         residuals: Sequence[torch.Tensor] = (), act_after: int = MG_ACT_NONE, scale=None,
         use_bias: bool = True, out_dtype=BF16, layout: Optional[str] = None,
         conv: Optional[tuple] = None) -> torch.Tensor:
    """out[M,N] = epilogue(a[M,K] @ w^T).  ``conv=(H, W, Cin)`` switches the A
    loader to implicit-im2col 3x3 over an NHWC image (a = [B*H*W, Cin])."""
    _need_gpu(a)
    assert a.dtype == BF16 and a.ndim == 2 and a.stride(1) == 1
    M = a.shape[0]
    if out is None:  # row stride padded to 8 elements (the C ABI wants ldc % 4 == 0)
        out = torch.empty(M, ceil_to(w.N, 8), dtype=out_dtype, device=a.device)[:, : w.N]
    assert out.ndim == 2 and out.shape[0] == M and out.shape[1] >= w.N and out.stride(1) == 1
    d = GemmDesc()
    d.A, d.lda = a.data_ptr(), a.stride(0)
    if layout is None:
        layout = "ft" if w.ft is not None else "rm"
    if layout == "ft":
        d.W, d.ldw, d.w_layout = w.ft.data_ptr(), w.Kp, MG_W_FRAGTILED
    else:
        d.W, d.ldw, d.w_layout = w.rm.data_ptr(), w.Kp, MG_W_ROWMAJOR
    d.M, d.N, d.K = M, w.N, w.K
    if conv is None:
        assert a.shape[1] == w.K, f"A has K={a.shape[1]}, weight has K={w.K}"
        d.a_mode = MG_A_DENSE
    else:
        d.a_mode = MG_A_CONV3X3
        d.H, d.Wd, d.Cin = conv
        assert a.shape[1] == conv[2] and a.is_contiguous() and w.K == 9 * conv[2]
    d.zero_page = zero_page(a.device).data_ptr()
    d.ep = _epilogue(out, w.N, w.bias if use_bias else None, scale, act, residuals, act_after)
    check(L.load().mg_gemm_bf16(C.byref(d), _stream()), "mg_gemm_bf16")
    return out


def gemm_skinny(x: torch.Tensor, w: PackedLinear, out: Optional[torch.Tensor] = None, *, act: int = MG_ACT_NONE,
                residuals: Sequence[torch.Tensor] = (), act_after: int = MG_ACT_NONE, scale=None,
                use_bias: bool = True, out_dtype=BF16, variant: int = 0) -> torch.Tensor:
    """Decode-shape (M <= 16) weight-streaming GEMM; needs the fragment-tiled layout."""
    _need_gpu(x)
    assert x.dtype == BF16 and x.ndim == 2 and x.stride(1) == 1 and w.ft is not None
    assert x.shape[1] == w.Kp, "decode activations must span the padded K"
    M = x.shape[0]
    if out is None:
        out = torch.empty(M, ceil_to(w.N, 8), dtype=out_dtype, device=x.device)[:, : w.N]
    d = SkinnyDesc()
    d.X, d.ldx, d.W = x.data_ptr(), x.stride(0), w.ft.data_ptr()
    d.M, d.N, d.Kp, d.nt_hint = M, w.N, w.Kp, variant
    d.ep = _epilogue(out, w.N, w.bias if use_bias else None, scale, act, residuals, act_after)
    check(L.load().mg_gemm_skinny_bf16(C.byref(d), _stream()), "mg_gemm_skinny_bf16")
    return out


def layernorm(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, eps: float = 1e-5,
              out: Optional[torch.Tensor] = None) -> torch.Tensor:
    _need_gpu(x)
    assert x.dtype == BF16 and x.ndim == 2 and x.stride(1) == 1
    assert gamma.dtype == torch.float32 and beta.dtype == torch.float32
    if out is None:
        out = torch.empty(x.shape, dtype=BF16, device=x.device)
    check(L.load().mg_layernorm_bf16(x.data_ptr(), x.stride(0), gamma.data_ptr(), beta.data_ptr(),
                                     out.data_ptr(), out.stride(0), x.shape[0], x.shape[1], eps, _stream()),
          "mg_layernorm_bf16")
    return out


def embedding(ids: torch.Tensor, wte: torch.Tensor, out: torch.Tensor, row_off: int = 0) -> torch.Tensor:
    """out[b, row_off + t, :] = wte[ids[b, t]]; out is (B, S_total, d) contiguous."""
    _need_gpu(ids, wte, out)
    assert ids.dtype == torch.int64 and ids.ndim == 2 and ids.is_contiguous()
    assert wte.dtype == BF16 and wte.is_contiguous() and out.dtype == BF16 and out.ndim == 3
    B, T = ids.shape
    assert out.stride(2) == 1 and out.stride(1) == out.shape[2] and row_off + T <= out.shape[1]
    check(L.load().mg_embedding_bf16(ids.data_ptr(), B, T, wte.data_ptr(), wte.shape[0], wte.shape[1],
                                     out.data_ptr(), out.stride(0), row_off, _stream()), "mg_embedding_bf16")
    return out


def rotary_split(qkv, B, S, H, rot_dim, sin_t, cos_t, q_out, kcache, vcache, *, pos0: int = 0,
                 d_pos: Optional[torch.Tensor] = None, vt: Optional[torch.Tensor] = None):
    _need_gpu(qkv)
    Smax = kcache.shape[2]
    check(L.load().mg_rotary_split_bf16(qkv.data_ptr(), B, S, H, rot_dim, sin_t.data_ptr(), cos_t.data_ptr(), pos0,
                                        _p(d_pos), q_out.data_ptr(), kcache.data_ptr(), vcache.data_ptr(), Smax,
                                        _p(vt), 0 if vt is None else vt.shape[3], _stream()),
          "mg_rotary_split_bf16")


def attn_prefill(q, kcache, vt, out, B, H, S, lse: Optional[torch.Tensor] = None):
    _need_gpu(q)
    check(L.load().mg_attn_prefill_bf16(q.data_ptr(), kcache.data_ptr(), vt.data_ptr(), out.data_ptr(), _p(lse),
                                        B, H, S, kcache.shape[2], vt.shape[3], _stream()), "mg_attn_prefill_bf16")
    return out


def attn_decode(q, kcache, vcache, out, B, H, d_pos):
    _need_gpu(q)
    check(L.load().mg_attn_decode_bf16(q.data_ptr(), kcache.data_ptr(), vcache.data_ptr(), out.data_ptr(), B, H,
                                       kcache.shape[2], d_pos.data_ptr(), _stream()), "mg_attn_decode_bf16")
    return out


def argmax(logits: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    _need_gpu(logits)
    assert logits.dtype == torch.float32 and logits.ndim == 2 and logits.stride(1) == 1
    if out is None:
        out = torch.empty(logits.shape[0], dtype=torch.int64, device=logits.device)
    check(L.load().mg_argmax_f32(logits.data_ptr(), logits.stride(0), logits.shape[0], logits.shape[1],
                                 out.data_ptr(), _stream()), "mg_argmax_f32")
    return out


def advance_pos(d_pos: torch.Tensor, delta: int = 1):
    check(L.load().mg_advance_pos(d_pos.data_ptr(), delta, _stream()), "mg_advance_pos")


def avgpool2(x: torch.Tensor) -> torch.Tensor:
    """x: [B,H,W,C] NHWC bf16 contiguous."""
    _need_gpu(x)
    assert x.dtype == BF16 and x.is_contiguous() and x.ndim == 4
    B, H, W, Cc = x.shape
    y = torch.empty(B, H // 2, W // 2, Cc, dtype=BF16, device=x.device)
    check(L.load().mg_avgpool2_nhwc_bf16(x.data_ptr(), y.data_ptr(), B, H, W, Cc, _stream()), "mg_avgpool2_nhwc_bf16")
    return y


def stem_im2col(img: torch.Tensor) -> torch.Tensor:
    """img: [B,3,H,W] bf16 NCHW -> [B*(H/2)*(W/2), 32]."""
    _need_gpu(img)
    assert img.dtype == BF16 and img.is_contiguous() and img.shape[1] == 3
    B, _, H, W = img.shape
    out = torch.empty(B * (H // 2) * (W // 2), 32, dtype=BF16, device=img.device)
    check(L.load().mg_stem_im2col_bf16(img.data_ptr(), out.data_ptr(), B, H, W, _stream()), "mg_stem_im2col_bf16")
    return out


def build_labels(captions: torch.Tensor, prefix_len: int, eos: int) -> torch.Tensor:
    _need_gpu(captions)
    assert captions.dtype == torch.int64 and captions.ndim == 2 and captions.is_contiguous()
    B, S = captions.shape
    if S < prefix_len:
        raise AssertionError("captions.shape[1] must be >= prefix length")  # reference utils.py:349
    labels = torch.empty_like(captions)
    check(L.load().mg_build_labels_i64(captions.data_ptr(), labels.data_ptr(), B, S, prefix_len, eos, _stream()),
          "mg_build_labels_i64")
    return labels


def cross_entropy(logits: torch.Tensor, targets: torch.Tensor):
    """logits [R,V] fp32 (rows already shifted), targets [R] int64 (-100 = ignore).
    Returns (mean loss scalar tensor, per-row loss)."""
    _need_gpu(logits, targets)
    assert logits.dtype == torch.float32 and logits.stride(1) == 1 and targets.dtype == torch.int64
    R, V = logits.shape
    rows = torch.empty(R, dtype=torch.float32, device=logits.device)
    out = torch.empty(2, dtype=torch.float32, device=logits.device)
    check(L.load().mg_ce_rows_f32(logits.data_ptr(), logits.stride(0), targets.data_ptr(), rows.data_ptr(), R, V,
                                  _stream()), "mg_ce_rows_f32")
    check(L.load().mg_ce_reduce_f32(rows.data_ptr(), targets.data_ptr(), R, out.data_ptr(), _stream()),
          "mg_ce_reduce_f32")
    return out[0], rows
