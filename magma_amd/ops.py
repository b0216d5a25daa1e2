"""Tensor-level wrappers over the C ABI.  PyTorch is used for device memory,
streams and one-off weight re-layout only; every hot-path computation below is
a call into libmagma_hip.so.  No CPU path exists: tensors must be on a GPU."""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Sequence

import torch

from . import lib as L
from .lib import (MG_A_CONV3X3, MG_A_DENSE, MG_ACT_GELU_NEW, MG_ACT_NONE, MG_ACT_QUICK_GELU, MG_ACT_RELU, MG_AUX_GELU_GRAD,
                  MG_AUX_MUL, MG_AUX_NONE, MG_AUX_QUICK_GELU_GRAD, MG_AUX_RELU_GATE, MG_W_FRAGTILED, MG_W_ROWMAJOR, Epilogue, GemmDesc,
                  SkinnyDesc, check)

BF16 = torch.bfloat16
_zero_pages = {}


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _need_gpu(*ts):
    """Every op launches on the CURRENT device and its current stream (_stream): operands on a CPU, or on a GPU
    that is not the current one, are refused instead of faulting / silently peer-accessing (Magma.__init__ and the
    entry points of Magma / MagmaEngine make the model's device current)."""
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda:
            raise L.MagmaHipError(
                "magma_amd ops run on MI355X only (tensor on %s); there is no CPU fallback" % t.device)
        if t.device.index != torch.cuda.current_device():
            raise L.MagmaHipError("tensor on %s but the current HIP device is cuda:%d: wrap the call in "
                                  "torch.cuda.device(tensor.device)" % (t.device, torch.cuda.current_device()))


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

    def rows(self, n0: int, n1: int, bias: Optional[torch.Tensor] = None) -> "PackedLinear":
        """Output channels [n0, n1) of this weight as a PackedLinear that SHARES its storage (n0 % 16 == 0: whole row tiles of
        the fragment-tiled layout are contiguous)."""
        assert n0 % 16 == 0 and 0 <= n0 < n1 <= self.N and (n1 % 16 == 0 or n1 == self.N)
        v = PackedLinear.__new__(PackedLinear)
        v.N, v.K, v.Kp, v.device = n1 - n0, self.K, self.Kp, self.device
        v.ft = None if self.ft is None else self.ft[n0 // 16: (n1 + 15) // 16]
        v.rm = None if self.rm is None else self.rm[n0:n1]
        v.bias = bias
        return v

    @staticmethod
    def tile(w: torch.Tensor) -> torch.Tensor:
        n, k = w.shape
        return w.view(n // 16, 16, k // 32, 4, 8).permute(0, 2, 3, 1, 4).contiguous()

    @staticmethod
    def untile(ft: torch.Tensor) -> torch.Tensor:
        nt, ks = ft.shape[0], ft.shape[1]
        return ft.view(nt, ks, 4, 16, 8).permute(0, 3, 1, 2, 4).reshape(nt * 16, ks * 32)


def _epilogue(out: Optional[torch.Tensor], N: int, bias, scale, act, residuals, act_after, aux=None, aux_mode=MG_AUX_NONE,
              aux_after=False, out2=None, mx_out=None, M: int = 0) -> Epilogue:
    """``mx_out`` = (q [M, ceil(N / 128) * 128] uint8, scales uint8 [mg_mx_scale_bytes(M, N)]) from ``mx_empty``: the epilogue also
    writes the OCP MX e4m3 copy of its final value (mg_epilogue.C8); ``out`` may then be None (no bf16 output at all)."""
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
    if out is not None:
        ep.C = out.data_ptr()
        ep.ldc = out.stride(0)
        ep.out_f32 = 1 if out.dtype == torch.float32 else 0
    if mx_out is not None:
        q8, sc8 = mx_out
        _need_gpu(q8, sc8)
        assert q8.dtype == torch.uint8 and q8.ndim == 2 and q8.stride(1) == 1 and q8.shape[0] >= M and q8.stride(0) == ceil_to(N, 128)
        assert sc8.dtype == torch.uint8 and sc8.numel() == int(L.load().mg_mx_scale_bytes(M, N))
        ep.C8, ep.ldc8, ep.c8_scales, ep.c8_rgroups = q8.data_ptr(), q8.stride(0), sc8.data_ptr(), (M + 63) // 64
    if aux is not None and aux_mode != MG_AUX_NONE:
        _need_gpu(aux)
        assert aux.dtype == BF16 and aux.ndim == 2 and aux.stride(1) == 1 and aux.shape[1] >= N
        ep.aux, ep.ldaux, ep.aux_mode, ep.aux_after = aux.data_ptr(), aux.stride(0), aux_mode, int(bool(aux_after))
    if out2 is not None:
        assert out2.dtype == BF16 and out2.ndim == 2 and out2.stride(1) == 1 and out2.shape[1] >= N
        ep.C2, ep.ldc2 = out2.data_ptr(), out2.stride(0)
    return ep


_SPLITK_WS = {}


def splitk_workspace(device) -> torch.Tensor:
    """fp32 scratch of the split-K GEMMs, one per (device, stream) -- launches on one stream are ordered,
    so they can share it.  MAGMA_SPLITK_WS_MB sizes it (default 64)."""
    key = (torch.cuda.current_device(), _stream())        # same device / stream the kernels launch on
    ws = _SPLITK_WS.get(key)
    if ws is None:
        mb = int(os.environ.get("MAGMA_SPLITK_WS_MB", "64"))
        ws = _SPLITK_WS[key] = torch.empty(mb << 18, dtype=torch.float32, device=device)
    return ws


def gemm(a: torch.Tensor, w: PackedLinear, out: Optional[torch.Tensor] = None, *, act: int = MG_ACT_NONE,
         residuals: Sequence[torch.Tensor] = (), act_after: int = MG_ACT_NONE, scale=None,
         use_bias: bool = True, out_dtype=BF16, layout: Optional[str] = None,
         conv: Optional[tuple] = None, aux=None, aux_mode: int = MG_AUX_NONE, aux_after: bool = False,
         out2: Optional[torch.Tensor] = None, tile: int = 0, split_k: int = 0, act_n0: int = 0,
         accumulate: bool = False, row_scale: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out[M,N] = epilogue(a[M,K] @ w^T).  ``act_n0``: ``act`` applies to output columns >= act_n0 only.  ``conv=(H, W, Cin)`` switches the A
    loader to implicit-im2col 3x3 over an NHWC image (a = [B*H*W, Cin]).
    ``split_k``: 0 lets the library cut K when the grid would leave the chip idle, 1 never, n forces n.
    ``accumulate`` (fp32 ``out`` only): out += ... instead of out = ...; ``row_scale`` fp32 [M]: per-row factor of the accumulator
    (mg_epilogue.accumulate / .row_scale -- the weight-gradient GEMMs add into the gradient buffer in place)."""
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
        d.W, d.ldw, d.w_layout = w.rm.data_ptr(), w.rm.stride(0), MG_W_ROWMAJOR
    d.M, d.N, d.K = M, w.N, w.K
    if conv is None:
        assert a.shape[1] == w.K, f"A has K={a.shape[1]}, weight has K={w.K}"
        d.a_mode = MG_A_DENSE
    else:
        d.a_mode = MG_A_CONV3X3
        d.H, d.Wd, d.Cin = conv
        assert a.shape[1] == conv[2] and a.is_contiguous() and w.K == 9 * conv[2]
    d.zero_page = zero_page(a.device).data_ptr()
    d.tile_hint = tile
    d.split_k = split_k
    if split_k != 1:
        ws = splitk_workspace(a.device)
        d.workspace, d.workspace_bytes = ws.data_ptr(), ws.numel() * 4
    d.ep = _epilogue(out, w.N, w.bias if use_bias else None, scale, act, residuals, act_after, aux, aux_mode,
                     aux_after, out2)
    d.ep.act_n0 = act_n0
    if accumulate:
        assert out.dtype == torch.float32, "accumulate adds into an fp32 output"
        d.ep.accumulate = 1
    if row_scale is not None:
        _need_gpu(row_scale)
        assert row_scale.dtype == torch.float32 and row_scale.is_contiguous() and row_scale.numel() >= M
        d.ep.row_scale = row_scale.data_ptr()
    check(L.load().mg_gemm_bf16(C.byref(d), _stream()), "mg_gemm_bf16")
    return out


class RawWeight:
    """Row-major [N, K] bf16 tensor used directly as the GEMM B operand (no copy):
    trainable weights, and the transposed activations of the wgrad GEMMs."""

    def __init__(self, w: torch.Tensor, bias: Optional[torch.Tensor] = None, K: Optional[int] = None):
        _need_gpu(w)
        assert w.dtype == BF16 and w.ndim == 2 and w.stride(1) == 1 and w.stride(0) % 8 == 0
        self.N = w.shape[0]
        self.K = w.shape[1] if K is None else K
        self.Kp = w.stride(0)
        self.rm, self.ft = w, None
        self.bias = bias


def pad_k_rowmajor(w: torch.Tensor) -> torch.Tensor:
    """[N, K] bf16 -> contiguous row-major GEMM operand whose row stride is a multiple of 8 elements."""
    n, k = w.shape
    if k % 8 == 0:
        return w.contiguous()
    out = torch.zeros(n, ceil_to(k, 8), dtype=w.dtype, device=w.device)
    out[:, :k] = w
    return out


def skinny_desc(x: torch.Tensor, w: PackedLinear, out: Optional[torch.Tensor] = None, *, act: int = MG_ACT_NONE,
                residuals: Sequence[torch.Tensor] = (), act_after: int = MG_ACT_NONE, scale=None,
                use_bias: bool = True, out_dtype=BF16, variant: int = 0, ln_fold: Optional[tuple] = None,
                split: Optional[tuple] = None):
    """Build the C descriptor of one decode-shape (M <= 16) weight-streaming GEMM.  Returns
    (desc, out, keepalive)."""
    _need_gpu(x)
    assert x.dtype == BF16 and x.ndim == 2 and x.stride(1) == 1 and w.ft is not None
    assert x.shape[1] == w.Kp, "decode activations must span the padded K"
    w8 = isinstance(w, PackedLinearW8)
    M = x.shape[0]
    if out is None:
        out = torch.empty(M, ceil_to(w.N, 8), dtype=out_dtype, device=x.device)[:, : w.N]
    d = SkinnyDesc()
    d.X, d.ldx, d.W = x.data_ptr(), x.stride(0), w.ft.data_ptr()
    d.M, d.N, d.Kp, d.nt_hint = M, w.N, w.Kp, variant
    if w8:
        d.w_scale = w.scale.data_ptr()
    n_a = w.N if split is None else split[0]
    d.ep = _epilogue(out, n_a, w.bias if use_bias else None, scale, act, residuals, act_after)
    if ln_fold is not None:
        cs, dd, eps = ln_fold
        d.ln_colsum, d.ln_inv_d, d.ln_eps = cs.data_ptr(), 1.0 / dd, eps
    if split is not None:
        split_n, out_b, act_b, bias_b = split
        d.split_n = split_n
        d.ep_b = _epilogue(out_b, w.N - split_n, bias_b, None, act_b, (), MG_ACT_NONE)
    return d, out


class PackedLinearW8:
    """Decode weight as e4m3 bytes + one fp32 scale per output channel, for the W8A16 weight-streaming GEMV (bf16
    activations; the kernel widens the bytes in registers).  ``ft`` = [ceil(N/16)][Kp/64][64 lanes][16 B]: lane
    kq*16+n holds W[n][64j + 8kq .. +7] then W[n][64j + 32 + 8kq .. +7].  K must be a multiple of 1024."""

    def __init__(self, weight: torch.Tensor, bias: Optional[torch.Tensor] = None):
        _need_gpu(weight)
        self.N, self.K = weight.shape
        assert self.K % 1024 == 0, "W8A16 decode weights need K % 1024 == 0"
        self.Kp = self.K
        n16 = ceil_to(self.N, 16)
        w = torch.zeros(n16, self.K, dtype=BF16, device=weight.device)
        w[: self.N] = weight.detach().to(BF16)
        q, sc = quantize_rows_fp8(w, self.K)
        self.scale = sc.contiguous()                               # padded to n16 (rows beyond N are zero)
        self.bias = None if bias is None else bias.detach().to(torch.float32).contiguous()
        # [n16, K] -> [nt, n(16), pair j, step(2), kq(4), 8] -> [nt, j, kq, n, step, 8]
        self.ft = q.view(n16 // 16, 16, self.K // 64, 2, 4, 8).permute(0, 2, 4, 1, 3, 5).contiguous()

    def dequant(self) -> torch.Tensor:
        n16 = self.ft.shape[0] * 16
        q = self.ft.permute(0, 3, 1, 4, 2, 5).reshape(n16, self.K)
        return (q.view(torch.float8_e4m3fn).float() * self.scale[:, None])[: self.N]


def gemm_skinny(x: torch.Tensor, w: PackedLinear, out: Optional[torch.Tensor] = None, **kw) -> torch.Tensor:
    """Decode-shape (M <= 16) weight-streaming GEMM; needs the fragment-tiled layout.
    ``ln_fold=(colsum fp32 [N], d, eps)``: LayerNorm of x folded into the GEMV (weights and
    bias must be pre-folded, see fold_layernorm).  ``split=(split_n, out_b, act_b, bias_b)``:
    columns >= split_n are written to ``out_b`` with their own activation / bias vector."""
    d, out = skinny_desc(x, w, out, **kw)
    check(L.load().mg_gemm_skinny_bf16(C.byref(d), _stream()), "mg_gemm_skinny_bf16")
    return out


def gemm_skinny2(a: tuple, b: tuple):
    """Two independent decode GEMVs in one launch; a, b = (x, w, out, kwargs)."""
    da, oa = skinny_desc(a[0], a[1], a[2], **a[3])
    db, ob = skinny_desc(b[0], b[1], b[2], **b[3])
    check(L.load().mg_gemm_skinny2_bf16(C.byref(da), C.byref(db), _stream()), "mg_gemm_skinny2_bf16")
    return oa, ob


def decode_attn_gemv(qkv, kcache, vcache, attn_out, B, H, d_pos, rot_dim, sin_t, cos_t, gemv: tuple):
    """Decode attention (rotary + append + attend) co-launched with one independent GEMV
    gemv = (x, w, out, kwargs)."""
    _need_gpu(qkv)
    d, og = skinny_desc(gemv[0], gemv[1], gemv[2], **gemv[3])
    assert attn_out.ndim == 2 and attn_out.stride(1) == 1 and attn_out.shape[1] == H * 256   # a column range of a wider row is fine
    check(L.load().mg_decode_attn_gemv_bf16(qkv.data_ptr(), kcache.data_ptr(), vcache.data_ptr(), attn_out.data_ptr(),
                                            attn_out.stride(0), B, H, kcache.shape[2], d_pos.data_ptr(), rot_dim, sin_t.data_ptr(),
                                            cos_t.data_ptr(), C.byref(d), _stream()), "mg_decode_attn_gemv_bf16")
    return attn_out, og


def fold_layernorm(weight: torch.Tensor, bias: Optional[torch.Tensor], gamma: torch.Tensor, beta: torch.Tensor):
    """LN(x) W^T + b  ==  rstd*(x W'^T - mean*colsum) + b'  with  W' = W*gamma (bf16),
    colsum[n] = sum_k W'[n][k] (of the rounded W'), b' = b + W beta.  One-off weight prep."""
    wf = weight.detach().float()
    w2 = (wf * gamma.float()[None, :]).to(BF16)
    colsum = w2.float().sum(1).contiguous()
    b2 = wf @ beta.float()
    if bias is not None:
        b2 = b2 + bias.detach().float()
    return w2, b2.contiguous(), colsum


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
    assert qkv.ndim == 2 and qkv.stride(1) == 1          # [B*S, >= 3*H*256]: a column range of a wider GEMM output is fine
    Smax = kcache.shape[2]
    check(L.load().mg_rotary_split_bf16(qkv.data_ptr(), qkv.stride(0), B, S, H, rot_dim, sin_t.data_ptr(), cos_t.data_ptr(), pos0,
                                        _p(d_pos), q_out.data_ptr(), kcache.data_ptr(), vcache.data_ptr(), Smax,
                                        _p(vt), 0 if vt is None else vt.shape[2] * 32, _stream()),
          "mg_rotary_split_bf16")


def rotary_split_train(qkv, B, S, H, rot_dim, sin_t, cos_t, q, k, v, vt, qt, kt):
    """Training form of rotary_split: q, k, v [B,H,S,256] and the three transposes [B,H,ld/32,256,32] in one pass."""
    _need_gpu(qkv)
    check(L.load().mg_rotary_split_train_bf16(qkv.data_ptr(), B, S, H, rot_dim, sin_t.data_ptr(), cos_t.data_ptr(),
                                              q.data_ptr(), k.data_ptr(), v.data_ptr(), vt.data_ptr(), qt.data_ptr(),
                                              kt.data_ptr(), vt.shape[2] * 32, _stream()), "mg_rotary_split_train_bf16")


class AttnFP8Operands:
    """What mg_rotary_split_fp8 writes for the fp8 attention forward (include/magma_hip.h): OCP MX e4m3 copies of the rotated q, k
    and of v^T with their E8M0 scales."""

    def __init__(self, B, H, S, device):
        self.B, self.H, self.S = B, H, S
        self.Sp = int(L.load().mg_attn_fp8_scale_stride(S))
        nt = (S + 63) // 64
        u8 = dict(dtype=torch.uint8, device=device)
        self.q8, self.k8 = torch.empty(B, H, S, 256, **u8), torch.empty(B, H, S, 256, **u8)
        self.v8t = torch.empty(B, H, nt, 256, 64, **u8)
        self.eq, self.ek = torch.full((B, H, self.Sp), 127, **u8), torch.full((B, H, self.Sp), 127, **u8)
        self.sv8 = torch.empty(B, H, nt, 512, **u8)

    def dequant(self):
        """(q, k, v) [B,H,S,256] fp32: the values the fp8 attention kernel multiplies (tests / oracles)."""
        B, H, S = self.B, self.H, self.S
        f8 = torch.float8_e4m3fn
        sq = torch.exp2(self.eq[:, :, :S].float() - 127.0)[..., None]
        sk = torch.exp2(self.ek[:, :, :S].float() - 127.0)[..., None]
        q = self.q8.view(f8).float() * sq
        k = self.k8.view(f8).float() * sk
        nt = self.v8t.shape[2]
        vt = self.v8t.view(f8).float()                                     # [B,H,nt,256 d,64 bytes]
        sv = torch.exp2(self.sv8.view(B, H, nt, 2, 32, 8).float() - 127.0)  # [key block][d % 32][d / 32]
        sv = sv.permute(0, 1, 2, 3, 5, 4).reshape(B, H, nt, 2, 256)        # -> [key block][d]
        # byte 32 hi + 16 b + r  <->  key 32 b + (r & 3) + 8 (r >> 2) + 4 hi
        idx = torch.empty(64, dtype=torch.long)
        blk = torch.empty(64, dtype=torch.long)
        for hi in range(2):
            for b in range(2):
                for r in range(16):
                    idx[32 * b + (r & 3) + 8 * (r >> 2) + 4 * hi] = 32 * hi + 16 * b + r
                    blk[32 * b + (r & 3) + 8 * (r >> 2) + 4 * hi] = b
        idx, blk = idx.to(vt.device), blk.to(vt.device)
        v_keys = vt[..., idx]                                              # [B,H,nt,256,64 keys in natural order]
        v_keys = v_keys * sv[:, :, :, blk, :].permute(0, 1, 2, 4, 3)        # scale of (key block, d)
        v = v_keys.permute(0, 1, 2, 4, 3).reshape(B, H, nt * 64, 256)[:, :, :S]
        return q, k, v


def rotary_split_fp8(qkv, B, S, H, rot_dim, sin_t, cos_t, q=None, k=None, v=None, qt=None, kt=None, inplace: bool = False) -> AttnFP8Operands:
    """rotary_split_train without V^T + the OCP MX e4m3 operands of the fp8 attention forward (-> AttnFP8Operands).
    q / k / v (and qt / kt) None: forward only, just the e4m3 operands.  ``inplace``: the rotated q / k are also written back into
    ``qkv`` (rotary_qk_inplace's result from this pass)."""
    _need_gpu(qkv)
    assert qkv.is_contiguous() and qkv.shape == (B * S, 3 * H * 256)
    op = AttnFP8Operands(B, H, S, qkv.device)
    check(L.load().mg_rotary_split_fp8(qkv.data_ptr(), B, S, H, rot_dim, sin_t.data_ptr(), cos_t.data_ptr(), _p(q), _p(k),
                                       _p(v), _p(qt), _p(kt), (qt.shape[2] * 32) if qt is not None else 0,
                                       op.q8.data_ptr(), op.k8.data_ptr(), op.v8t.data_ptr(), op.eq.data_ptr(), op.ek.data_ptr(),
                                       op.sv8.data_ptr(), int(bool(inplace)), _stream()), "mg_rotary_split_fp8")
    return op


def attn_prefill_fp8(op: AttnFP8Operands, out, lse: Optional[torch.Tensor] = None, mx_out=None):
    """Causal flash attention on the fp8 MFMA (mg_attn_prefill_fp8): out [B*S, >= H*256] bf16 (a column range of a wider row is fine).
    ``mx_out`` = (q, scales) from mx_empty(B*S, H*256): the epilogue also writes the OCP MX e4m3 copy of out."""
    assert out.ndim == 2 and out.stride(1) == 1 and out.shape[1] == op.H * 256 and out.stride(0) % 8 == 0
    q8, sc8 = mx_out if mx_out is not None else (None, None)
    if q8 is not None:
        assert q8.dtype == torch.uint8 and q8.shape[0] >= op.B * op.S and q8.stride(0) == ceil_to(op.H * 256, 128)
        assert sc8.numel() == int(L.load().mg_mx_scale_bytes(op.B * op.S, op.H * 256))
    check(L.load().mg_attn_prefill_fp8(op.q8.data_ptr(), op.k8.data_ptr(), op.v8t.data_ptr(), op.eq.data_ptr(), op.ek.data_ptr(),
                                       op.sv8.data_ptr(), out.data_ptr(), out.stride(0), _p(lse), op.B, op.H, op.S,
                                       _p(q8), 0 if q8 is None else q8.stride(0), _p(sc8), _stream()),
          "mg_attn_prefill_fp8")
    return out


def attn_prefill(q, kcache, vt, out, B, H, S, lse: Optional[torch.Tensor] = None):
    _need_gpu(q)
    assert out.ndim == 2 and out.stride(1) == 1 and out.shape[1] == H * 256      # a column range of a wider row is fine
    check(L.load().mg_attn_prefill_bf16(q.data_ptr(), kcache.data_ptr(), vt.data_ptr(), out.data_ptr(), out.stride(0), _p(lse),
                                        B, H, S, kcache.shape[2], vt.shape[2] * 32, _stream()), "mg_attn_prefill_bf16")
    return out


def attn_decode(q, kcache, vcache, out, B, H, d_pos):
    _need_gpu(q)
    check(L.load().mg_attn_decode_bf16(q.data_ptr(), kcache.data_ptr(), vcache.data_ptr(), out.data_ptr(), B, H,
                                       kcache.shape[2], d_pos.data_ptr(), _stream()), "mg_attn_decode_bf16")
    return out


def attn_decode_fused(qkv, kcache, vcache, out, B, H, d_pos, rot_dim, sin_t, cos_t):
    """rotary(q,k) + KV append at *d_pos + attention over [0, *d_pos], one launch."""
    _need_gpu(qkv)
    check(L.load().mg_attn_decode_fused_bf16(qkv.data_ptr(), kcache.data_ptr(), vcache.data_ptr(), out.data_ptr(), B, H,
                                             kcache.shape[2], d_pos.data_ptr(), rot_dim, sin_t.data_ptr(),
                                             cos_t.data_ptr(), _stream()), "mg_attn_decode_fused_bf16")
    return out


def argmax(logits: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    _need_gpu(logits)
    assert logits.dtype == torch.float32 and logits.ndim == 2 and logits.stride(1) == 1
    if out is None:
        out = torch.empty(logits.shape[0], dtype=torch.int64, device=logits.device)
    check(L.load().mg_argmax_f32(logits.data_ptr(), logits.stride(0), logits.shape[0], logits.shape[1],
                                 out.data_ptr(), _stream()), "mg_argmax_f32")
    return out


def sample(logits: torch.Tensor, temperature: float, top_k: int, top_p: float, seed: Optional[torch.Tensor],
           state: Optional[torch.Tensor], out: Optional[torch.Tensor] = None, filtered: Optional[torch.Tensor] = None,
           want_token: bool = True):
    """reference sampling.py:99-107 on the device: top-k filter, the reference's top-p rule, softmax(logits / temperature)
    and one multinomial draw per row (Philox stream keyed by ``seed`` at counter (state[0], row)).  ``filtered`` receives
    the filtered logits (the tensor the reference's filters return) when given."""
    _need_gpu(logits, seed, state, out, filtered)
    assert logits.dtype == torch.float32 and logits.ndim == 2 and logits.stride(1) == 1
    B, V = logits.shape
    if want_token and out is None:
        out = torch.empty(B, dtype=torch.int64, device=logits.device)
    if filtered is not None:
        assert filtered.dtype == torch.float32 and filtered.shape == logits.shape and filtered.stride(1) == 1
    check(L.load().mg_sample_f32(logits.data_ptr(), logits.stride(0), B, V, float(temperature), int(top_k), float(top_p),
                                 _p(seed), _p(state), _p(out) if want_token else None, _p(filtered),
                                 0 if filtered is None else filtered.stride(0), _stream()), "mg_sample_f32")
    return out


def sample_finish(token: torch.Tensor, eos: int, state: torch.Tensor, d_pos: Optional[torch.Tensor] = None, delta: int = 1,
                  history: Optional[torch.Tensor] = None, clear: Optional[torch.Tensor] = None, clear_stride: int = 1):
    """Bookkeeping of one token step: first all-eos step, step counter, (optionally) KV write position += delta and the
    token history [B, n_steps] int64."""
    _need_gpu(token, state, d_pos, history)
    assert token.dtype == torch.int64 and state.dtype == torch.int32 and state.numel() >= 2
    if history is not None:
        assert history.dtype == torch.int64 and history.ndim == 2 and history.stride(1) == 1 and history.shape[0] == token.numel()
    if clear is not None:
        assert clear.dtype == torch.int32 and clear.is_contiguous()
    check(L.load().mg_sample_finish(token.data_ptr(), token.numel(), int(eos), state.data_ptr(), _p(d_pos), delta, _p(history),
                                    0 if history is None else history.stride(0), 0 if history is None else history.shape[1],
                                    _p(clear), 0 if clear is None else clear.numel() // clear_stride, clear_stride, _stream()),
          "mg_sample_finish")


def advance_pos(d_pos: torch.Tensor, delta: int = 1):
    check(L.load().mg_advance_pos(d_pos.data_ptr(), delta, _stream()), "mg_advance_pos")


def patchify(img: torch.Tensor, P: int) -> torch.Tensor:
    """img [B,3,H,W] bf16 -> [B*(H/P)*(W/P), 3*P*P] rows in (c, py, px) order (im2col of the stride-P patch conv)."""
    _need_gpu(img)
    assert img.dtype == BF16 and img.is_contiguous() and img.ndim == 4 and img.shape[1] == 3
    B, _, H, W = img.shape
    out = torch.empty(B * (H // P) * (W // P), 3 * P * P, dtype=BF16, device=img.device)
    check(L.load().mg_patchify_bf16(img.data_ptr(), out.data_ptr(), B, H, W, P, _stream()), "mg_patchify_bf16")
    return out


def vit_embed(patches: torch.Tensor, cls: torch.Tensor, pos: torch.Tensor, B: int) -> torch.Tensor:
    """[class | patches] + positional embedding -> [B, G+1, width] bf16."""
    _need_gpu(patches, cls, pos)
    G, width = patches.shape[0] // B, patches.shape[1]
    assert patches.dtype == BF16 and cls.dtype == BF16 and pos.dtype == BF16 and pos.shape == (G + 1, width)
    out = torch.empty(B, G + 1, width, dtype=BF16, device=patches.device)
    check(L.load().mg_vit_embed_bf16(patches.data_ptr(), cls.contiguous().data_ptr(), pos.contiguous().data_ptr(), out.data_ptr(),
                                     B, G, width, _stream()), "mg_vit_embed_bf16")
    return out


def attn_small(qkv: torch.Tensor, B: int, S: int, H: int) -> torch.Tensor:
    """Non-causal attention, head dim 64, S <= 256: qkv [B*S, 3*H*64] -> [B*S, H*64]."""
    _need_gpu(qkv)
    assert qkv.dtype == BF16 and qkv.is_contiguous() and qkv.shape == (B * S, 3 * H * 64)
    out = torch.empty(B * S, H * 64, dtype=BF16, device=qkv.device)
    check(L.load().mg_attn_small_bf16(qkv.data_ptr(), out.data_ptr(), B, S, H, _stream()), "mg_attn_small_bf16")
    return out


def attn_small_bwd(qkv: torch.Tensor, d_out: torch.Tensor, B: int, S: int, H: int) -> torch.Tensor:
    """Backward of attn_small: qkv [B*S, 3*H*64], d_out [B*S, H*64] -> d_qkv [B*S, 3*H*64] (S <= 64)."""
    _need_gpu(qkv, d_out)
    assert qkv.dtype == BF16 and qkv.is_contiguous() and qkv.shape == (B * S, 3 * H * 64)
    assert d_out.dtype == BF16 and d_out.is_contiguous() and d_out.shape == (B * S, H * 64)
    dqkv = torch.empty_like(qkv)
    check(L.load().mg_attn_small_bwd_bf16(qkv.data_ptr(), d_out.data_ptr(), dqkv.data_ptr(), B, S, H, _stream()), "mg_attn_small_bwd_bf16")
    return dqkv


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


def weight_standardize(w: torch.Tensor, gain: torch.Tensor, scale: float, eps: float, to_khwc: bool = False,
                       ldo: Optional[int] = None) -> torch.Tensor:
    """timm ScaledStdConv2d weight transform: w [cout,cin,kh,kw] bf16, gain [cout(,1,1,1)] -> [cout, ldo] bf16 GEMM operand."""
    _need_gpu(w, gain)
    assert w.dtype == BF16 and gain.dtype == BF16 and w.ndim == 4 and w.is_contiguous() and gain.numel() == w.shape[0]
    cout, cin, kh, kw = w.shape
    ldo = ceil_to(cin * kh * kw, 8) if ldo is None else ldo
    out = torch.empty(cout, ldo, dtype=BF16, device=w.device)
    check(L.load().mg_weight_standardize_bf16(w.data_ptr(), gain.contiguous().data_ptr(), out.data_ptr(), cout, cin, kh, kw, ldo,
                                              int(bool(to_khwc)), float(scale), float(eps), _stream()), "mg_weight_standardize_bf16")
    return out


def im2col_nchw(img: torch.Tensor, k: int, stride: int, pad: int, ldo: int) -> torch.Tensor:
    """img [B,C,H,W] bf16 NCHW -> [B*Ho*Wo, ldo], column (c*k + ky)*k + kx (small-Cin strided stem convs)."""
    _need_gpu(img)
    assert img.dtype == BF16 and img.is_contiguous() and img.ndim == 4
    B, Cc, H, W = img.shape
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    out = torch.empty(B * Ho * Wo, ldo, dtype=BF16, device=img.device)
    check(L.load().mg_im2col_nchw_bf16(img.data_ptr(), out.data_ptr(), B, Cc, H, W, k, stride, pad, ldo, _stream()), "mg_im2col_nchw_bf16")
    return out


def _pool_s2(fn_name: str, x: torch.Tensor) -> torch.Tensor:
    _need_gpu(x)
    assert x.dtype == BF16 and x.is_contiguous() and x.ndim == 4
    B, H, W, Cc = x.shape
    y = torch.empty(B, (H - 1) // 2 + 1, (W - 1) // 2 + 1, Cc, dtype=BF16, device=x.device)
    check(getattr(L.load(), fn_name)(x.data_ptr(), y.data_ptr(), B, H, W, Cc, _stream()), fn_name)
    return y


def maxpool3x3s2(x: torch.Tensor) -> torch.Tensor:
    """MaxPool2d(3, stride 2, padding 1) on [B,H,W,C] NHWC bf16."""
    return _pool_s2("mg_maxpool3x3s2_nhwc_bf16", x)


def subsample2(x: torch.Tensor) -> torch.Tensor:
    """x[:, ::2, ::2, :] of an NHWC bf16 map as a contiguous tensor."""
    return _pool_s2("mg_subsample2_nhwc_bf16", x)


def relu_mean_rows(x: torch.Tensor) -> torch.Tensor:
    """[B, HW, C] bf16 -> [B, C]: mean over the positions of relu(x)."""
    _need_gpu(x)
    assert x.dtype == BF16 and x.is_contiguous() and x.ndim == 3
    B, HW, Cc = x.shape
    y = torch.empty(B, Cc, dtype=BF16, device=x.device)
    check(L.load().mg_relu_mean_rows_bf16(x.data_ptr(), y.data_ptr(), B, HW, Cc, _stream()), "mg_relu_mean_rows_bf16")
    return y


def weight_standardize_bwd(w: torch.Tensor, gain: torch.Tensor, dwhat: torch.Tensor, dw: torch.Tensor, dgain: torch.Tensor,
                           scale: float, eps: float, dmult: float = 1.0):
    """Backward of weight_standardize: dwhat [cout, >= fan_in] fp32 (weight's own column order) -> dw [cout, fan_in] +=, dgain [cout] +=."""
    _need_gpu(w, gain, dwhat, dw, dgain)
    cout = w.shape[0]
    fan_in = w.numel() // cout
    assert w.dtype == BF16 and gain.dtype == BF16 and w.is_contiguous() and gain.numel() == cout
    assert dwhat.dtype == torch.float32 and dwhat.ndim == 2 and dwhat.stride(1) == 1 and dwhat.shape[1] >= fan_in
    assert dw.dtype == torch.float32 and dw.is_contiguous() and dw.numel() == w.numel() and dgain.dtype == torch.float32 and dgain.numel() == cout
    check(L.load().mg_weight_standardize_bwd_f32(w.data_ptr(), gain.contiguous().data_ptr(), dwhat.data_ptr(), dwhat.stride(0), dw.data_ptr(),
                                                 dgain.data_ptr(), cout, fan_in, float(scale), float(eps), float(dmult), _stream()), "mg_weight_standardize_bwd_f32")


def maxpool3x3s2_bwd(x: torch.Tensor, dy: torch.Tensor) -> torch.Tensor:
    _need_gpu(x, dy)
    assert x.dtype == BF16 and dy.dtype == BF16 and x.is_contiguous() and dy.is_contiguous() and x.ndim == 4
    B, H, W, Cc = x.shape
    dx = torch.empty_like(x)
    check(L.load().mg_maxpool3x3s2_bwd_nhwc_bf16(x.data_ptr(), dy.data_ptr(), dx.data_ptr(), B, H, W, Cc, _stream()), "mg_maxpool3x3s2_bwd_nhwc_bf16")
    return dx


def subsample2_bwd(dy: torch.Tensor, H: int, W: int) -> torch.Tensor:
    _need_gpu(dy)
    assert dy.dtype == BF16 and dy.is_contiguous() and dy.ndim == 4
    B, _, _, Cc = dy.shape
    dx = torch.empty(B, H, W, Cc, dtype=BF16, device=dy.device)
    check(L.load().mg_subsample2_bwd_nhwc_bf16(dy.data_ptr(), dx.data_ptr(), B, H, W, Cc, _stream()), "mg_subsample2_bwd_nhwc_bf16")
    return dx


def relu_mean_rows_bwd(x: torch.Tensor, g: torch.Tensor) -> torch.Tensor:
    _need_gpu(x, g)
    assert x.dtype == BF16 and g.dtype == BF16 and x.is_contiguous() and g.is_contiguous() and x.ndim == 3
    B, HW, Cc = x.shape
    dx = torch.empty_like(x)
    check(L.load().mg_relu_mean_rows_bwd_bf16(x.data_ptr(), g.data_ptr(), dx.data_ptr(), B, HW, Cc, _stream()), "mg_relu_mean_rows_bwd_bf16")
    return dx


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


# ======================= training path (backward + optimizer) =======================

def transpose(x: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """[R, C] -> [C, round_up(R,8)]; the padding columns are zero (they are the K
    padding of the wgrad GEMM that consumes the result)."""
    _need_gpu(x)
    assert x.dtype == BF16 and x.ndim == 2 and x.stride(1) == 1
    R, Cc = x.shape
    if out is None:
        out = torch.empty(Cc, ceil_to(R, 8), dtype=BF16, device=x.device)
    check(L.load().mg_transpose_bf16(x.data_ptr(), x.stride(0), 0, out.data_ptr(), out.stride(0), 0, R, Cc, 1, _stream()),
          "mg_transpose_bf16")
    return out


def transpose_colsum(x: torch.Tensor, colsum_out: torch.Tensor) -> torch.Tensor:
    """transpose(x) that also accumulates the column sums of x into colsum_out (fp32 [C]): one pass instead of two."""
    _need_gpu(x, colsum_out)
    assert x.dtype == BF16 and x.ndim == 2 and x.stride(1) == 1 and colsum_out.dtype == torch.float32 and colsum_out.numel() >= x.shape[1]
    R, Cc = x.shape
    out = torch.empty(Cc, ceil_to(R, 8), dtype=BF16, device=x.device)
    check(L.load().mg_transpose_colsum_bf16(x.data_ptr(), x.stride(0), out.data_ptr(), out.stride(0), R, Cc, colsum_out.data_ptr(), _stream()),
          "mg_transpose_colsum_bf16")
    return out


def head_transpose(src: torch.Tensor, B: int, H: int, S: int, sb: int, ss: int, sh: int,
                   out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """-> column-tiled transposed layout [B, H, ceil(S/32), 256, 32]: out[b,h,t,d,i] = src[b, 32t+i, h, d], zero padded."""
    _need_gpu(src)
    ld = ceil_to(S, 32)
    if out is None:
        out = torch.empty(B, H, ld // 32, 256, 32, dtype=BF16, device=src.device)
    check(L.load().mg_head_transpose_bf16(src.data_ptr(), sb, ss, sh, out.data_ptr(), ld, B, H, S, _stream()),
          "mg_head_transpose_bf16")
    return out


def colsum(x: torch.Tensor, out: torch.Tensor, y: Optional[torch.Tensor] = None):
    """out[n] += sum_m x[m,n] * (y[m,n] if y else 1); out fp32 [N]."""
    _need_gpu(x, out)
    assert x.dtype == BF16 and x.ndim == 2 and x.stride(1) == 1 and out.dtype == torch.float32
    check(L.load().mg_colsum_f32(x.data_ptr(), x.stride(0), _p(y), 0 if y is None else y.stride(0), out.data_ptr(),
                                 x.shape[0], x.shape[1], _stream()), "mg_colsum_f32")
    return out


def layernorm_bwd(dy, x, gamma, eps=1e-5, res=None, want_xhat=False):
    _need_gpu(dy, x)
    assert dy.dtype == BF16 and x.dtype == BF16 and dy.shape == x.shape and dy.stride(1) == 1 and x.stride(1) == 1
    dx = torch.empty(x.shape, dtype=BF16, device=x.device)
    xhat = torch.empty(x.shape, dtype=BF16, device=x.device) if want_xhat else None
    check(L.load().mg_layernorm_bwd_bf16(dy.data_ptr(), dy.stride(0), x.data_ptr(), x.stride(0), gamma.data_ptr(),
                                         _p(res), 0 if res is None else res.stride(0), dx.data_ptr(), dx.stride(0),
                                         _p(xhat), 0 if xhat is None else xhat.stride(0), x.shape[0], x.shape[1], eps,
                                         _stream()), "mg_layernorm_bwd_bf16")
    return (dx, xhat) if want_xhat else dx


def cross_entropy_fwd_bwd(logits: torch.Tensor, targets: torch.Tensor, ld_out: int):
    """loss + dlogits (bf16 [R, ld_out], zero padded) for rows already shifted."""
    _need_gpu(logits, targets)
    R, V = logits.shape
    rows = torch.empty(R, dtype=torch.float32, device=logits.device)
    stats = torch.empty(2, dtype=torch.float32, device=logits.device)
    lib = L.load()
    check(lib.mg_ce_rows_f32(logits.data_ptr(), logits.stride(0), targets.data_ptr(), rows.data_ptr(), R, V, _stream()), "mg_ce_rows_f32")
    check(lib.mg_ce_reduce_f32(rows.data_ptr(), targets.data_ptr(), R, stats.data_ptr(), _stream()), "mg_ce_reduce_f32")
    dl = torch.empty(R, ld_out, dtype=BF16, device=logits.device)
    check(lib.mg_ce_bwd_bf16(logits.data_ptr(), logits.stride(0), targets.data_ptr(), stats.data_ptr(), dl.data_ptr(),
                             ld_out, R, V, _stream()), "mg_ce_bwd_bf16")
    return stats[0], dl


MG_ACT_GELU_ERF = 100      # NOT an epilogue code: torch.nn.GELU() runs as its own pass (gelu_erf / gelu_erf_grad_mul below)
MG_AUX_GELU_ERF_GRAD = 100


def gelu_erf(x: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """torch.nn.GELU() (erf) over a [rows, cols] bf16 matrix (row stride allowed); ``out`` may be x."""
    _need_gpu(x)
    assert x.dtype == BF16 and x.ndim == 2 and x.stride(1) == 1 and x.shape[1] % 8 == 0
    if out is None:
        out = torch.empty_like(x)
    check(L.load().mg_gelu_erf_bf16(x.data_ptr(), x.stride(0), None, 0, out.data_ptr(), out.stride(0), x.shape[0], x.shape[1], _stream()),
          "mg_gelu_erf_bf16")
    return out


def gelu_erf_grad_mul(g: torch.Tensor, pre: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """g * gelu'(pre) (erf form), same shapes; ``out`` may be g."""
    _need_gpu(g, pre)
    assert g.dtype == BF16 and pre.dtype == BF16 and g.shape == pre.shape and g.stride(1) == 1 and pre.stride(1) == 1 and g.shape[1] % 8 == 0
    if out is None:
        out = torch.empty_like(g)
    check(L.load().mg_gelu_erf_bf16(pre.data_ptr(), pre.stride(0), g.data_ptr(), g.stride(0), out.data_ptr(), out.stride(0),
                                    g.shape[0], g.shape[1], _stream()), "mg_gelu_erf_bf16")
    return out


def rotary_merge_bwd(dq, dk, dv, B, S, H, rot_dim, sin_t, cos_t):
    _need_gpu(dq)
    out = torch.empty(B * S, 3 * H * 256, dtype=BF16, device=dq.device)
    check(L.load().mg_rotary_merge_bwd_bf16(dq.data_ptr(), dk.data_ptr(), dv.data_ptr(), B, S, H, rot_dim,
                                            sin_t.data_ptr(), cos_t.data_ptr(), out.data_ptr(), _stream()),
          "mg_rotary_merge_bwd_bf16")
    return out


def attn_bwd(q, k, v, qt, kt, dO, dOt, O, lse, B, H, S):
    """q,k,v [B,H,S,256]; qt,kt,dOt [B,H,256,ld]; dO [B*S,H*256]; O [B*S, >= H*256] (any row stride); lse [B,H,S]
    -> dq,dk,dv [B,H,S,256]."""
    _need_gpu(q)
    assert O.ndim == 2 and O.stride(1) == 1 and dO.is_contiguous()
    dev = q.device
    D = torch.empty(B, H, S, 2, dtype=torch.float32, device=dev)   # {-16 lse, -rowsum(dO o O)} per query (workspace)
    dq, dk, dv = (torch.empty(B, H, S, 256, dtype=BF16, device=dev) for _ in range(3))
    check(L.load().mg_attn_bwd_bf16(q.data_ptr(), k.data_ptr(), v.data_ptr(), qt.data_ptr(), kt.data_ptr(),
                                    dO.data_ptr(), dOt.data_ptr(), O.data_ptr(), lse.data_ptr(), D.data_ptr(),
                                    dq.data_ptr(), dk.data_ptr(), dv.data_ptr(), B, H, S, qt.shape[2] * 32, O.stride(0), _stream()),
          "mg_attn_bwd_bf16")
    return dq, dk, dv


def attn_bwd_merged(q, k, v, qt, kt, dO, O, lse, B, H, S, rot_dim, sin_t, cos_t):
    """dO transpose + attn_bwd + rotary_merge_bwd in one call: -> dqkv [B*S, 3*H*256] (gradient of the fused qkv
    projection)."""
    _need_gpu(q)
    assert O.ndim == 2 and O.stride(1) == 1 and dO.is_contiguous()       # O may be the [:, :d] view of a [ctx | t] buffer
    dev = q.device
    D = torch.empty(B, H, S, 2, dtype=torch.float32, device=dev)
    dOt = torch.empty_like(qt)                                       # workspace, filled by the first launch
    dqkv = torch.empty(B * S, 3 * H * 256, dtype=BF16, device=dev)
    check(L.load().mg_attn_bwd_merged_bf16(q.data_ptr(), k.data_ptr(), v.data_ptr(), qt.data_ptr(), kt.data_ptr(),
                                           dO.data_ptr(), dOt.data_ptr(), O.data_ptr(), lse.data_ptr(), D.data_ptr(),
                                           dqkv.data_ptr(), rot_dim, sin_t.data_ptr(), cos_t.data_ptr(), B, H, S,
                                           qt.shape[2] * 32, O.stride(0), _stream()), "mg_attn_bwd_merged_bf16")
    return dqkv


class AttnRows:
    """q / k / v of a batch of heads as strided rows of 256 (include/magma_hip.h, attention without transposed images): either three
    [B,H,S,256] tensors or -- ``AttnRows.of_qkv`` -- column ranges of the fused qkv activation [B*S, 3 H 256] itself."""

    def __init__(self, q, k, v, ld_row, stride_b, stride_h, B, H, S, keep=()):
        self.q, self.k, self.v, self.ld_row, self.stride_b, self.stride_h = q, k, v, ld_row, stride_b, stride_h
        self.B, self.H, self.S, self.keep = B, H, S, keep          # keep: the tensors the pointers live in

    @classmethod
    def of_bhsd(cls, q, k, v):
        B, H, S, dh = q.shape
        assert dh == 256 and q.is_contiguous() and k.is_contiguous() and v.is_contiguous() and k.shape == q.shape == v.shape
        return cls(q.data_ptr(), k.data_ptr(), v.data_ptr(), 256, H * S * 256, S * 256, B, H, S, keep=(q, k, v))

    @classmethod
    def of_qkv(cls, qkv, B, S, H):
        assert qkv.ndim == 2 and qkv.stride(1) == 1 and qkv.shape[0] == B * S and qkv.shape[1] >= 3 * H * 256
        ld, p = qkv.stride(0), qkv.data_ptr()
        return cls(p, p + H * 256 * 2, p + 2 * H * 256 * 2, ld, S * ld, 256, B, H, S, keep=(qkv,))


def rotary_qk_inplace(qkv, B, S, H, rot_dim, sin_t, cos_t):
    """GPT-J rotary on the q and k sections of qkv [B*S, >= 3 H 256], in place (v untouched)."""
    _need_gpu(qkv)
    assert qkv.ndim == 2 and qkv.stride(1) == 1 and qkv.shape[0] == B * S
    check(L.load().mg_rotary_qk_inplace_bf16(qkv.data_ptr(), qkv.stride(0), B, S, H, rot_dim, sin_t.data_ptr(), cos_t.data_ptr(),
                                             _stream()), "mg_rotary_qk_inplace_bf16")
    return qkv


def attn_fwd_rows(x: AttnRows, out, lse: Optional[torch.Tensor] = None):
    """Causal flash attention reading q / k / v as strided rows (no V^T): out [B*S, >= H*256] (a column range of a wider row is fine)."""
    _need_gpu(out)
    assert out.ndim == 2 and out.stride(1) == 1 and out.shape[1] == x.H * 256
    check(L.load().mg_attn_fwd_rows_bf16(x.q, x.k, x.v, x.ld_row, x.stride_b, x.stride_h, out.data_ptr(), out.stride(0), _p(lse),
                                         x.B, x.H, x.S, _stream()), "mg_attn_fwd_rows_bf16")
    return out


def attn_bwd_rows(x: AttnRows, dO, O, lse, merged_rot=None, mx_out=None, no_out: bool = False, first_rows: int = 0):
    """Attention backward without transposed operands.  merged_rot None -> (dq, dk, dv) [B,H,S,256]; merged_rot = (rot_dim, sin_t,
    cos_t) -> dqkv [B*S, 3 H 256], the gradient of the fused qkv projection (inverse rotary applied).  Merged form only:
    ``mx_out`` = (q, scales) from mx_empty(B*S, 3 H 256) -> the epilogue also writes the OCP MX e4m3 copy of dqkv; with ``no_out``
    only that copy (returns None).  ``first_rows`` > 0: only the gradients of the positions < first_rows of every sequence are
    wanted (rounded up to whole 128-position blocks; the other rows of the outputs stay unwritten)."""
    _need_gpu(dO)
    assert O.ndim == 2 and O.stride(1) == 1 and dO.is_contiguous()
    B, H, S, dev = x.B, x.H, x.S, dO.device
    D = torch.empty(B, H, S, 2, dtype=torch.float32, device=dev)
    if merged_rot is None:
        dq, dk, dv = (torch.empty(B, H, S, 256, dtype=BF16, device=dev) for _ in range(3))
        check(L.load().mg_attn_bwd_rows_bf16(x.q, x.k, x.v, x.ld_row, x.stride_b, x.stride_h, dO.data_ptr(), O.data_ptr(), O.stride(0),
                                             lse.data_ptr(), D.data_ptr(), dq.data_ptr(), dk.data_ptr(), dv.data_ptr(), None, 0, None, None,
                                             B, H, S, None, None, int(first_rows), _stream()), "mg_attn_bwd_rows_bf16")
        return dq, dk, dv
    rot_dim, sin_t, cos_t = merged_rot
    q8, sc8 = mx_out if mx_out is not None else (None, None)
    assert not no_out or q8 is not None
    if q8 is not None:
        assert q8.dtype == torch.uint8 and q8.shape[0] >= B * S and q8.stride(0) == 3 * H * 256
        assert sc8.numel() == int(L.load().mg_mx_scale_bytes(B * S, 3 * H * 256))
    dqkv = None if no_out else torch.empty(B * S, 3 * H * 256, dtype=BF16, device=dev)
    check(L.load().mg_attn_bwd_rows_bf16(x.q, x.k, x.v, x.ld_row, x.stride_b, x.stride_h, dO.data_ptr(), O.data_ptr(), O.stride(0),
                                         lse.data_ptr(), D.data_ptr(), None, None, None, _p(dqkv), rot_dim, sin_t.data_ptr(),
                                         cos_t.data_ptr(), B, H, S, _p(q8), _p(sc8), int(first_rows), _stream()), "mg_attn_bwd_rows_bf16")
    return dqkv


def avgpool2_bwd(dy: torch.Tensor, B, H, W, Cc, gate: Optional[torch.Tensor] = None) -> torch.Tensor:
    """dy [B,H/2,W/2,C] -> dx [B,H,W,C] (zeroed where gate <= 0 when a gate is given)."""
    _need_gpu(dy)
    dx = torch.empty(B, H, W, Cc, dtype=BF16, device=dy.device)
    check(L.load().mg_avgpool2_bwd_nhwc_bf16(dy.data_ptr(), _p(gate), dx.data_ptr(), B, H, W, Cc, _stream()), "mg_avgpool2_bwd_nhwc_bf16")
    return dx


def mul(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    _need_gpu(a, b)
    out = torch.empty_like(a)
    check(L.load().mg_mul_bf16(a.data_ptr(), b.data_ptr(), out.data_ptr(), a.numel(), _stream()), "mg_mul_bf16")
    return out


def scale_rows_acc(dst: torch.Tensor, src: torch.Tensor, row_scale: Optional[torch.Tensor] = None):
    """dst[r,c] += src[r,c] * row_scale[r]; dst fp32 contiguous view of a flat gradient buffer."""
    _need_gpu(dst, src)
    assert dst.dtype == torch.float32 and src.dtype == torch.float32 and src.stride(1) == 1 and dst.is_contiguous()
    rows, cols = dst.shape
    check(L.load().mg_scale_rows_acc_f32(dst.data_ptr(), src.data_ptr(), src.stride(0), _p(row_scale), rows, cols, _stream()),
          "mg_scale_rows_acc_f32")


def add_gate(a, b=None, gate=None, out=None):
    """out = (a + b) * (gate > 0); b, gate optional; flat bf16 tensors of equal numel (multiple of 8)."""
    _need_gpu(a)
    if out is None:
        out = torch.empty_like(a)
    check(L.load().mg_add_gate_bf16(a.data_ptr(), _p(b), _p(gate), out.data_ptr(), a.numel(), _stream()), "mg_add_gate_bf16")
    return out


def bn_param_grad(g, y, sub, gamma, beta, dgamma, dbeta):
    _need_gpu(g)
    M, Cc = g.shape
    check(L.load().mg_bn_param_grad_f32(g.data_ptr(), y.data_ptr(), _p(sub), gamma.data_ptr(), beta.data_ptr(),
                                        dgamma.data_ptr(), dbeta.data_ptr(), M, Cc, _stream()), "mg_bn_param_grad_f32")


def transpose_bn_param_grad(g, y, sub, gamma, beta, dgamma, dbeta) -> torch.Tensor:
    """g^T [C, round_up(M, 8)] (as ``transpose``) and, from the same pass over g, the frozen-statistics BatchNorm parameter gradients
    that ``bn_param_grad`` accumulates."""
    _need_gpu(g, y, sub, gamma, beta, dgamma, dbeta)
    M, Cc = g.shape
    assert g.dtype == BF16 and g.stride(1) == 1 and y.shape == g.shape and y.stride() == g.stride() and (sub is None or (sub.shape == g.shape and sub.stride() == g.stride()))
    out = torch.empty(Cc, ceil_to(M, 8), dtype=BF16, device=g.device)
    check(L.load().mg_transpose_bn_param_grad_bf16(g.data_ptr(), g.stride(0), out.data_ptr(), out.stride(0), M, Cc, y.data_ptr(), _p(sub),
                                                   gamma.data_ptr(), beta.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(), _stream()),
          "mg_transpose_bn_param_grad_bf16")
    return out


def im2col_t(x_nhwc: torch.Tensor, B, H, W, Cin) -> torch.Tensor:
    """-> [Cin*9, round_up(M,8)] (row = ci*9 + tap, zero padded columns) for the 3x3 wgrad GEMM."""
    _need_gpu(x_nhwc)
    M = B * H * W
    ldo = ceil_to(M, 8)
    out = torch.empty(9 * Cin, ldo, dtype=BF16, device=x_nhwc.device)
    check(L.load().mg_im2col_t_bf16(x_nhwc.data_ptr(), out.data_ptr(), ldo, B, H, W, Cin, _stream()), "mg_im2col_t_bf16")
    return out


def sumsq(g: torch.Tensor, out: torch.Tensor):
    """out[0] += sum(g^2); g fp32 or bf16 (the exchanged buckets of the data-parallel step)."""
    _need_gpu(g, out)
    fn = L.load().mg_sumsq_bf16 if g.dtype == BF16 else L.load().mg_sumsq_f32
    check(fn(g.data_ptr(), g.numel(), out.data_ptr(), _stream()), "mg_sumsq")


def cast_f32_bf16(src: torch.Tensor, dst: torch.Tensor):
    _need_gpu(src, dst)
    assert src.dtype == torch.float32 and dst.dtype == BF16 and src.numel() == dst.numel() and src.is_contiguous()
    check(L.load().mg_cast_f32_bf16(src.data_ptr(), dst.data_ptr(), src.numel(), _stream()), "mg_cast_f32_bf16")
    return dst


def adamw(p, m, v, g, p_bf16, lr, beta1, beta2, eps, wd, step, max_norm=0.0, norm_sq=None, grad_scale=1.0):
    """Fused clip + AdamW on fp32 master / m / v; g = gradient sum in fp32 or bf16."""
    _need_gpu(p, g)
    fn = L.load().mg_adamw_gbf16_f32 if g.dtype == BF16 else L.load().mg_adamw_f32
    check(fn(p.data_ptr(), m.data_ptr(), v.data_ptr(), g.data_ptr(), _p(p_bf16), p.numel(), lr,
             beta1, beta2, eps, wd, step, max_norm, _p(norm_sq), grad_scale, _stream()), "mg_adamw")


def bn_fold(gamma, beta, mean, var, eps: float):
    """(scale, shift) fp32 [C] of a frozen-statistics BatchNorm: scale = gamma / sqrt(var + eps), shift = beta - mean*scale."""
    _need_gpu(gamma)
    for t in (gamma, beta, mean, var):
        assert t.dtype == torch.float32 and t.is_contiguous()
    scale, shift = torch.empty_like(gamma), torch.empty_like(gamma)
    check(L.load().mg_bn_fold_f32(gamma.data_ptr(), beta.data_ptr(), mean.data_ptr(), var.data_ptr(), eps, scale.data_ptr(),
                                  shift.data_ptr(), gamma.numel(), _stream()), "mg_bn_fold_f32")
    return scale, shift


def bn_batch_fold(s1, s2, gamma, beta, M: int, eps: float, momentum: float, running_mean=None, running_var=None):
    """Per-channel batch statistics -> (scale, shift, mean, rstd) fp32 [C]; running statistics updated in place (fp32 tensors)."""
    _need_gpu(s1, s2, gamma, beta)
    C_ = s1.numel()
    scale, shift, mean, rstd = (torch.empty(C_, dtype=torch.float32, device=s1.device) for _ in range(4))
    check(L.load().mg_bn_batch_fold_f32(s1.data_ptr(), s2.data_ptr(), gamma.data_ptr(), beta.data_ptr(), M, eps, momentum,
                                        _p(running_mean), _p(running_var), scale.data_ptr(), shift.data_ptr(), mean.data_ptr(),
                                        rstd.data_ptr(), C_, _stream()), "mg_bn_batch_fold_f32")
    return scale, shift, mean, rstd


def bn_apply(z, scale, shift, res=None, relu=True):
    _need_gpu(z, res)
    assert z.dtype == BF16 and z.ndim == 2 and z.is_contiguous()
    y = torch.empty_like(z)
    check(L.load().mg_bn_apply_bf16(z.data_ptr(), scale.data_ptr(), shift.data_ptr(), _p(res), int(bool(relu)), y.data_ptr(),
                                    z.shape[0], z.shape[1], _stream()), "mg_bn_apply_bf16")
    return y


def bn_bwd_dz(g, z, mean, rstd, gamma, dgamma, dbeta):
    _need_gpu(g, z)
    assert g.dtype == BF16 and z.dtype == BF16 and g.shape == z.shape and g.is_contiguous() and z.is_contiguous()
    dz = torch.empty_like(z)
    check(L.load().mg_bn_bwd_dz_bf16(g.data_ptr(), z.data_ptr(), mean.data_ptr(), rstd.data_ptr(), gamma.data_ptr(),
                                     dgamma.data_ptr(), dbeta.data_ptr(), dz.data_ptr(), z.shape[0], z.shape[1], _stream()),
          "mg_bn_bwd_dz_bf16")
    return dz


def conv_weight_relayout(w: torch.Tensor, mode: int, scale: Optional[torch.Tensor] = None) -> torch.Tensor:
    """[Cout, Cin, k, k] bf16 conv weight -> row-major GEMM operand with K zero padded to a multiple of 64.
    mode 0: [Cout, k*k*Cin] in (ky, kx, ci) order; mode 1: [Cin, k*k*Cout] with flipped taps times scale[co] (dgrad)."""
    _need_gpu(w)
    assert w.dtype == BF16 and w.ndim == 4 and w.is_contiguous() and w.shape[2] == w.shape[3]
    cout, cin, k, _ = w.shape
    rows, inner = (cout, cin) if mode == 0 else (cin, cout)
    ldo = ceil_to(k * k * inner, 64)
    out = torch.empty(rows, ldo, dtype=BF16, device=w.device)
    check(L.load().mg_conv_weight_relayout_bf16(w.data_ptr(), _p(scale), out.data_ptr(), ldo, cout, cin, k, mode, _stream()),
          "mg_conv_weight_relayout_bf16")
    return out


def _job_table(jobs, device) -> torch.Tensor:
    """ctypes job structs -> device byte tensor (the DEVICE array the batched entry points read)."""
    arr = (type(jobs[0]) * len(jobs))(*jobs)
    return torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(device)


class ConvOperandPlan:
    """GEMM operands of MANY trainable convolutions + the folded affine of their frozen-statistics BatchNorms, re-derived from the
    current weights by TWO launches (mg_bn_fold_batch, then mg_conv_weight_relayout_batch) -- the training engine calls ``refresh``
    once per forward instead of ~3 small launches per convolution.  ``units``: list of (weight bf16 [Cout,Cin,k,k] with k in (1, 3),
    gamma, beta, mean, var fp32 [Cout], eps); weight / statistics storage must stay where it is (the tables hold raw pointers).
    Per unit i afterwards: ``fwd[i]`` [Cout, ld] (mode 0), ``dgrad[i]`` [Cin, ld] (mode 1, BN scale folded), ``scale[i]``, ``shift[i]``."""

    def __init__(self, units, device):
        from .lib import BnFoldJob, RelayoutJob
        self.keep = units
        ctot = sum(u[0].shape[0] for u in units)
        self._aff = torch.empty(2, ctot, dtype=torch.float32, device=device)
        self.scale, self.shift, self.fwd, self.dgrad = [], [], [], []
        sizes = []
        for (w, *_rest) in units:
            _need_gpu(w)
            assert w.dtype == BF16 and w.ndim == 4 and w.is_contiguous() and w.shape[2] == w.shape[3] and w.shape[2] in (1, 3)
            cout, cin, k, _ = w.shape
            # (a 1x1 weight whose Cin is a whole number of 64-element K-tiles already IS its forward operand: no copy)
            sizes.append((0 if (k == 1 and cin % 64 == 0) else cout * ceil_to(k * k * cin, 64), cin * ceil_to(k * k * cout, 64)))
        self._ops = torch.empty(sum(a + b for a, b in sizes), dtype=BF16, device=device)
        bn_jobs, rl_jobs, c0, o0, bb, rb = [], [], 0, 0, 0, 0
        for (w, gamma, beta, mean, var, eps), (n0, n1) in zip(units, sizes):
            cout, cin, k, _ = w.shape
            for t in (gamma, beta, mean, var):
                _need_gpu(t)
                assert t.dtype == torch.float32 and t.is_contiguous() and t.numel() == cout
            sc, sh = self._aff[0, c0:c0 + cout], self._aff[1, c0:c0 + cout]
            c0 += cout
            bn_jobs.append(BnFoldJob(gamma.data_ptr(), beta.data_ptr(), mean.data_ptr(), var.data_ptr(), sc.data_ptr(), sh.data_ptr(),
                                     float(eps), cout, bb))
            bb += (cout + 255) // 256
            g = self._ops[o0 + n0:o0 + n0 + n1].view(cin, -1)
            if n0:
                f = self._ops[o0:o0 + n0].view(cout, -1)
                rl_jobs.append(RelayoutJob(w.data_ptr(), None, f.data_ptr(), f.shape[1], cout, cin, k, 0, rb))
                rb += (n0 + 255) // 256
            else:
                f = w.view(cout, cin)
            o0 += n0 + n1
            rl_jobs.append(RelayoutJob(w.data_ptr(), sc.data_ptr(), g.data_ptr(), g.shape[1], cout, cin, k, 1, rb))
            rb += (n1 + 255) // 256
            self.scale.append(sc); self.shift.append(sh); self.fwd.append(f); self.dgrad.append(g)
        self._bn_tab, self._rl_tab = _job_table(bn_jobs, device), _job_table(rl_jobs, device)
        self._nbn, self._bn_blocks, self._nrl, self._rl_blocks = len(bn_jobs), bb, len(rl_jobs), rb

    def refresh(self):
        lib = L.load()
        check(lib.mg_bn_fold_batch(self._bn_tab.data_ptr(), self._nbn, self._bn_blocks, _stream()), "mg_bn_fold_batch")
        check(lib.mg_conv_weight_relayout_batch(self._rl_tab.data_ptr(), self._nrl, self._rl_blocks, _stream()), "mg_conv_weight_relayout_batch")


# ---- image preprocessing (reference magma/transforms.py:121-134) ------------------------------------------------
def resample_u8(img: torch.Tensor, out_size: int, axis: int, coeffs: torch.Tensor, bounds: torch.Tensor) -> torch.Tensor:
    """One pass of Pillow's 8-bit antialiased resampling on an HWC uint8 RGB image (axis 1: width, axis 0: height)."""
    _need_gpu(img)
    assert img.dtype == torch.uint8 and img.ndim == 3 and img.shape[2] == 3 and img.is_contiguous()
    assert coeffs.dtype == torch.int32 and bounds.dtype == torch.int32 and coeffs.is_contiguous() and bounds.is_contiguous()
    H, W, _ = img.shape
    out = torch.empty((H, out_size, 3) if axis == 1 else (out_size, W, 3), dtype=torch.uint8, device=img.device)
    check(L.load().mg_resample_u8(img.data_ptr(), H, W, out.data_ptr(), out_size, axis, coeffs.data_ptr(), bounds.data_ptr(),
                                  coeffs.shape[1], _stream()), "mg_resample_u8")
    return out


def crop_normalize(img: torch.Tensor, top: int, left: int, n: int, mean, std) -> torch.Tensor:
    """HWC uint8 -> [3, n, n] fp32, (v / 255 - mean) / std  (CenterCrop + ToTensor + Normalize)."""
    _need_gpu(img)
    assert img.dtype == torch.uint8 and img.ndim == 3 and img.shape[2] == 3 and img.is_contiguous()
    out = torch.empty(3, n, n, dtype=torch.float32, device=img.device)
    m3, s3 = (C.c_float * 3)(*mean), (C.c_float * 3)(*std)
    check(L.load().mg_crop_normalize_f32(img.data_ptr(), img.shape[0], img.shape[1], top, left, n, m3, s3, out.data_ptr(), _stream()),
          "mg_crop_normalize_f32")
    return out


# ---- fp8 (OCP e4m3) GEMM operands: BASELINE config 5 ---------------------------------------------------------------
def quantize_rows_fp8(x: torch.Tensor, ldq: Optional[int] = None):
    """bf16 [M, K] -> (uint8 e4m3 [M, ldq] zero padded, fp32 row scales [M]);  x ~= q * scale[:, None]."""
    _need_gpu(x)
    assert x.dtype == BF16 and x.ndim == 2 and x.stride(1) == 1
    M, K = x.shape
    ldq = ceil_to(K, 128) if ldq is None else ldq
    q = torch.empty(M, ldq, dtype=torch.uint8, device=x.device)
    scale = torch.empty(M, dtype=torch.float32, device=x.device)
    check(L.load().mg_quantize_rows_fp8(x.data_ptr(), x.stride(0), M, K, q.data_ptr(), ldq, scale.data_ptr(), _stream()),
          "mg_quantize_rows_fp8")
    return q, scale


class PackedLinearFP8:
    """A [N, K] weight as e4m3 bytes with one fp32 scale per output channel (W ~= q * scale[:, None]), in the same two
    layouts as PackedLinear: row-major [N, Kp] and fragment-tiled (the bf16 tiling applied to byte PAIRS, Kp % 128 == 0)."""

    def __init__(self, weight: torch.Tensor, bias: Optional[torch.Tensor] = None, tiled: bool = True, rowmajor: bool = False):
        _need_gpu(weight)
        self.N, self.K = weight.shape
        assert self.K % 16 == 0, "fp8 GEMM needs K % 16 == 0"
        self.Kp = ceil_to(self.K, 128)
        n16 = ceil_to(self.N, 16)
        w = torch.zeros(n16, self.K, dtype=BF16, device=weight.device)
        w[: self.N] = weight.detach().to(BF16)
        q, sc = quantize_rows_fp8(w, self.Kp)
        self.scale = sc[: self.N].contiguous()
        self.bias = None if bias is None else bias.detach().to(torch.float32).contiguous()
        self.rm = q[: self.N].contiguous() if rowmajor else None
        self.ft = PackedLinear.tile(q.view(torch.int16)).view(torch.uint8) if tiled else None   # pairs of bytes tile like bf16

    def dequant(self) -> torch.Tensor:
        q = self.rm if self.rm is not None else PackedLinear.untile(self.ft.view(torch.int16)).view(torch.uint8)[: self.N]
        return q[:, : self.K].view(torch.float8_e4m3fn).float() * self.scale[:, None]

    @classmethod
    def of_live(cls, weight: torch.Tensor, bias: Optional[torch.Tensor] = None) -> "PackedLinearFP8":
        """The fragment-tiled pack of a TRAINABLE [N, K] bf16 weight (N % 16 == 0, K % 128 == 0), rebuilt every optimizer step: two
        launches (quantise, tile), no padding copies; ``bias`` is kept by reference (a live fp32 master view)."""
        self = cls.__new__(cls)
        self.N, self.K = weight.shape
        assert weight.dtype == BF16 and weight.stride(1) == 1 and self.N % 16 == 0 and self.K % 128 == 0
        self.Kp = self.K
        q, self.scale = quantize_rows_fp8(weight, self.Kp)
        self.bias, self.rm = bias, None
        self.ft = PackedLinear.tile(q.view(torch.int16)).view(torch.uint8)
        return self


def gemm_fp8(aq: torch.Tensor, a_scale: torch.Tensor, w: PackedLinearFP8, out: Optional[torch.Tensor] = None, *,
             act: int = MG_ACT_NONE, residuals: Sequence[torch.Tensor] = (), act_after: int = MG_ACT_NONE,
             use_bias: bool = True, out_dtype=BF16, layout: Optional[str] = None, split_k: int = 0,
             aux=None, aux_mode: int = MG_AUX_NONE, aux_after: bool = False, out2: Optional[torch.Tensor] = None,
             tile: int = 0, mx_out=None, no_out: bool = False):
    """out[M,N] = epilogue((aq @ wq^T) * a_scale[m] * w.scale[n]) on the fp8 MFMA (fp32 accumulate).
    ``tile``: 0 lets the library choose between the 128x128 and the 256x256 kernel, 128 / 256 force one.
    ``mx_out`` (from mx_empty): the epilogue also writes the MX e4m3 copy of the result; with ``no_out`` only that copy
    (returns None)."""
    _need_gpu(aq)
    assert aq.dtype == torch.uint8 and aq.ndim == 2 and aq.stride(1) == 1 and aq.shape[1] >= w.K
    M = aq.shape[0]
    assert not no_out or (mx_out is not None and out is None)
    if out is None and not no_out:
        out = torch.empty(M, ceil_to(w.N, 8), dtype=out_dtype, device=aq.device)[:, : w.N]
    d = GemmDesc()
    d.A, d.lda = aq.data_ptr(), aq.stride(0)
    if layout is None:
        layout = "ft" if w.ft is not None else "rm"
    if layout == "ft":
        d.W, d.ldw, d.w_layout = w.ft.data_ptr(), w.Kp, MG_W_FRAGTILED
    else:
        d.W, d.ldw, d.w_layout = w.rm.data_ptr(), w.rm.stride(0), MG_W_ROWMAJOR
    d.M, d.N, d.K = M, w.N, w.K
    d.a_mode = MG_A_DENSE
    d.tile_hint = tile
    d.zero_page = zero_page(aq.device).data_ptr()
    d.split_k = split_k
    if split_k != 1:
        ws = splitk_workspace(aq.device)
        d.workspace, d.workspace_bytes = ws.data_ptr(), ws.numel() * 4
    d.ep = _epilogue(out, w.N, w.bias if use_bias else None, w.scale, act, residuals, act_after, aux, aux_mode, aux_after, out2,
                     mx_out=mx_out, M=M)
    check(L.load().mg_gemm_fp8(C.byref(d), a_scale.data_ptr(), _stream()), "mg_gemm_fp8")
    return out


# ---- OCP MX fp8 (BASELINE config 5 as stated: e4m3 elements + one E8M0 scale per 32 K-elements) ---------------------------------
def mx_scales_rowmajor(scales: torch.Tensor, rows: int, K: int) -> torch.Tensor:
    """The quantiser's scale bytes (include/magma_hip.h: [chunk][block][row / 64][row % 16][(row % 64) / 16]) as a plain
    uint8 [rows, ceil(K / 128) * 4] matrix of E8M0 exponents, one per (row, 32-block)."""
    chunks, rg = (K + 127) // 128, (rows + 63) // 64
    v = scales.view(chunks, 4, rg, 16, 4).permute(2, 4, 3, 0, 1).reshape(rg * 64, chunks * 4)
    return v[:rows]


def mx_empty(M: int, N: int, device):
    """(q, scales) of an [M, N] MX operand for a GEMM epilogue to fill (``mx_out``): what quantize_mx_fp8 returns.  Columns
    N .. ceil(N / 128) * 128 of q are zeroed (the consumer's K loop reads whole 128-element chunks) and every scale byte is
    2^0 until written."""
    Kp = ceil_to(N, 128)
    q = torch.empty(M, Kp, dtype=torch.uint8, device=device)
    if Kp != N:
        q[:, N:].zero_()
    # scale bytes start at 127 (2^0), as the quantiser writes them for the zero blocks past N: the C8 epilogue only writes the
    # blocks with n < N, and the consuming GEMM walks whole 128-element chunks -- a leftover 0xFF byte is NaN in E8M0.  When every
    # byte WILL be written (whole 64-row groups, whole 128-column chunks: the training shapes) the fill launch is skipped.
    nbytes = int(L.load().mg_mx_scale_bytes(M, N))
    if M % 64 == 0 and Kp == N:
        return q, torch.empty(nbytes, dtype=torch.uint8, device=device)
    return q, torch.full((nbytes,), 127, dtype=torch.uint8, device=device)


def quantize_mx_fp8(x: torch.Tensor):
    """bf16 [M, K] -> (uint8 e4m3 [M, ldq] in K order, zero padded to ldq = ceil(K / 128) * 128; uint8 block scales in the
    layout of include/magma_hip.h mg_quantize_mx_fp8 -- mx_scales_rowmajor turns them into a [M, ldq / 32] matrix)."""
    _need_gpu(x)
    assert x.dtype == BF16 and x.ndim == 2 and x.stride(1) == 1
    M, K = x.shape
    ldq = ceil_to(K, 128)
    q = torch.empty(M, ldq, dtype=torch.uint8, device=x.device)
    nbytes = int(L.load().mg_mx_scale_bytes(M, K))
    if M % 64 == 0:      # whole 64-row groups: the kernel writes every scale byte (one per 32-column block of ldq, per row)
        scales = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
    else:
        scales = torch.full((nbytes,), 127, dtype=torch.uint8, device=x.device)
    check(L.load().mg_quantize_mx_fp8(x.data_ptr(), x.stride(0), M, K, q.data_ptr(), ldq, scales.data_ptr(), _stream()),
          "mg_quantize_mx_fp8")
    return q, scales


def mx_dequant(q: torch.Tensor, scales: torch.Tensor, K: int) -> torch.Tensor:
    """fp32 [R, K] values an MX operand stands for (tests; the fp8 'dequantised oracle')."""
    R, ld = q.shape
    e = mx_scales_rowmajor(scales, R, ld).float()                                   # [R, ld / 32]
    v = q.view(torch.float8_e4m3fn).float().view(R, ld // 32, 32) * torch.exp2(e - 127.0)[:, :, None]
    return v.reshape(R, ld)[:, :K]


class PackedLinearMX:
    """A [N, K] weight as MX fp8: e4m3 bytes (row-major and / or fragment-tiled exactly like PackedLinearFP8 -- the tiling acts
    on the byte image) + E8M0 block scales [N, Kp / 128] int32."""

    def __init__(self, weight: torch.Tensor, bias: Optional[torch.Tensor] = None, tiled: bool = True, rowmajor: bool = False):
        _need_gpu(weight)
        self.N, self.K = weight.shape
        assert self.K % 16 == 0, "fp8 GEMM needs K % 16 == 0"
        self.Kp = ceil_to(self.K, 128)
        n16 = ceil_to(self.N, 16)
        w = torch.zeros(n16, self.K, dtype=BF16, device=weight.device)
        w[: self.N] = weight.detach().to(BF16)
        q, sc = quantize_mx_fp8(w[: self.N])          # scale slabs are indexed by the weight's own row count
        if n16 != self.N:
            q = torch.cat([q, torch.zeros(n16 - self.N, q.shape[1], dtype=torch.uint8, device=q.device)])
        self.scales = sc
        self.bias = None if bias is None else bias.detach().to(torch.float32).contiguous()
        self.rm = q[: self.N].contiguous() if rowmajor else None
        self.ft = PackedLinear.tile(q.view(torch.int16)).view(torch.uint8) if tiled else None
        self._n16 = n16

    def dequant(self) -> torch.Tensor:
        q = self.rm if self.rm is not None else PackedLinear.untile(self.ft.view(torch.int16)).view(torch.uint8)[: self.N]
        return mx_dequant(q.contiguous(), self.scales, self.K)

    @classmethod
    def of_live(cls, weight: torch.Tensor, bias: Optional[torch.Tensor] = None) -> "PackedLinearMX":
        """As PackedLinearFP8.of_live, with OCP MX block scales."""
        self = cls.__new__(cls)
        self.N, self.K = weight.shape
        assert weight.dtype == BF16 and weight.stride(1) == 1 and self.N % 16 == 0 and self.K % 128 == 0
        self.Kp, self._n16 = self.K, self.N
        q, self.scales = quantize_mx_fp8(weight)
        self.bias, self.rm = bias, None
        self.ft = PackedLinear.tile(q.view(torch.int16)).view(torch.uint8)
        return self


def gemm_mx_fp8(aq: torch.Tensor, a_scales: torch.Tensor, w: PackedLinearMX, out: Optional[torch.Tensor] = None, *,
                act: int = MG_ACT_NONE, residuals: Sequence[torch.Tensor] = (), act_after: int = MG_ACT_NONE, use_bias: bool = True,
                out_dtype=BF16, layout: Optional[str] = None, split_k: int = 0, act_n0: int = 0, tile: int = 0,
                aux=None, aux_mode: int = MG_AUX_NONE, aux_after: bool = False, out2: Optional[torch.Tensor] = None,
                mx_out=None, no_out: bool = False):
    """out[M,N] = epilogue(sum over 32-blocks of 2^(ea + ew) * (qa . qw)) on the block-scaled fp8 MFMA (fp32 accumulate).
    ``mx_out`` / ``no_out`` as for gemm_fp8."""
    _need_gpu(aq, a_scales)
    assert aq.dtype == torch.uint8 and aq.ndim == 2 and aq.stride(1) == 1 and aq.shape[1] == w.Kp
    M = aq.shape[0]
    assert a_scales.dtype == torch.uint8 and a_scales.numel() == int(L.load().mg_mx_scale_bytes(M, w.Kp))
    assert not no_out or (mx_out is not None and out is None)
    if out is None and not no_out:
        out = torch.empty(M, ceil_to(w.N, 8), dtype=out_dtype, device=aq.device)[:, : w.N]
    d = GemmDesc()
    d.A, d.lda = aq.data_ptr(), aq.stride(0)
    if layout is None:
        layout = "ft" if w.ft is not None else "rm"
    if layout == "ft":
        d.W, d.ldw, d.w_layout = w.ft.data_ptr(), w.Kp, MG_W_FRAGTILED
    else:
        d.W, d.ldw, d.w_layout = w.rm.data_ptr(), w.rm.stride(0), MG_W_ROWMAJOR
    d.M, d.N, d.K = M, w.N, w.Kp
    d.a_mode = MG_A_DENSE
    d.zero_page = zero_page(aq.device).data_ptr()
    d.tile_hint = tile
    d.split_k = split_k
    if split_k != 1:
        ws = splitk_workspace(aq.device)
        d.workspace, d.workspace_bytes = ws.data_ptr(), ws.numel() * 4
    d.ep = _epilogue(out, w.N, w.bias if use_bias else None, None, act, residuals, act_after, aux, aux_mode, aux_after, out2,
                     mx_out=mx_out, M=M)
    d.ep.act_n0 = act_n0
    check(L.load().mg_gemm_mx_fp8(C.byref(d), a_scales.data_ptr(), w.scales.data_ptr(), _stream()), "mg_gemm_mx_fp8")
    return out


def debug_mx_mfma(a: torch.Tensor, sa: torch.Tensor, b: torch.Tensor, sb: torch.Tensor) -> torch.Tensor:
    """One v_mfma_scale_f32_16x16x128_f8f6f4 on explicit per-lane operands: a, b int32 [64, 8]; sa, sb int32 [64] -> fp32 [64, 4]."""
    _need_gpu(a, sa, b, sb)
    out = torch.empty(64, 4, dtype=torch.float32, device=a.device)
    check(L.load().mg_debug_mx_mfma(a.data_ptr(), sa.data_ptr(), b.data_ptr(), sb.data_ptr(), out.data_ptr(), _stream()), "mg_debug_mx_mfma")
    return out
