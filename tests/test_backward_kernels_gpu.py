"""Backward / optimizer kernels against torch autograd (fp32) on the same inputs."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
BF16 = torch.bfloat16


def rnd(*shape, dev, scale=1.0, seed=0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(dev)


def rel(a, b):
    a, b = a.float(), b.float()
    return float((a - b).norm() / (b.norm() + 1e-12))


def test_transposes(dev):
    from magma_amd import ops
    x = rnd(200, 136, dev=dev, seed=1).to(BF16)
    assert torch.equal(ops.transpose(x), x.t().contiguous())
    y = rnd(13, 16, dev=dev, seed=1).to(BF16)                       # R not a multiple of 8 -> zero padded columns
    yt = ops.transpose(y)
    assert yt.shape == (16, 16) and torch.equal(yt[:, :13], y.t()) and bool((yt[:, 13:] == 0).all())
    B, H, S = 2, 3, 57
    src = rnd(B * S, H * 256, dev=dev, seed=2).to(BF16)          # [M, d] activation -> per-head transposed
    out = ops.head_transpose(src, B, H, S, sb=S * H * 256, ss=H * 256, sh=256)
    def untile(t):     # [B,H,T,256,32] column-tiled -> [B,H,256,T*32]
        return t.permute(0, 1, 3, 2, 4).reshape(t.shape[0], t.shape[1], 256, -1)
    ref = src.view(B, S, H, 256).permute(0, 2, 3, 1)
    assert out.shape == (B, H, 2, 256, 32)
    assert torch.equal(untile(out)[..., :S], ref) and bool((untile(out)[..., S:] == 0).all())
    q = rnd(B, H, S, 256, dev=dev, seed=3).to(BF16)               # [B,H,S,256] -> [B,H,256,S]
    out = ops.head_transpose(q, B, H, S, sb=H * S * 256, ss=256, sh=S * 256)
    assert torch.equal(untile(out)[..., :S], q.transpose(2, 3))


def test_colsum(dev):
    from magma_amd import ops
    x = rnd(1000, 520, dev=dev, seed=4).to(BF16)
    y = rnd(1000, 520, dev=dev, seed=5).to(BF16)
    out = torch.zeros(520, device=dev)
    ops.colsum(x, out)
    assert rel(out, x.float().sum(0)) < 1e-5
    ops.colsum(x, out, y)                                         # accumulates
    assert rel(out, x.float().sum(0) + (x.float() * y.float()).sum(0)) < 1e-5


@pytest.mark.parametrize("rows,d", [(9, 4096), (33, 512)])
def test_layernorm_bwd(dev, rows, d):
    from magma_amd import ops
    x = (rnd(rows, d, dev=dev, seed=6) * 2 + 0.5).to(BF16)
    dy = rnd(rows, d, dev=dev, seed=7).to(BF16)
    res = rnd(rows, d, dev=dev, seed=8).to(BF16)
    g = rnd(d, dev=dev, seed=9) * 0.1 + 1
    xf = x.float().requires_grad_(True)
    yref = F.layer_norm(xf, (d,), g, torch.zeros(d, device=dev), 1e-5)
    yref.backward(dy.float())
    dx, xh = ops.layernorm_bwd(dy, x, g, 1e-5, res=res, want_xhat=True)
    assert rel(dx, xf.grad + res.float()) < 4e-3
    assert rel(xh, F.layer_norm(x.float(), (d,))) < 4e-3


def test_ce_fwd_bwd(dev):
    from magma_amd import ops
    R, V = 21, 1053
    lg = (rnd(R, V, dev=dev, seed=10) * 3)
    tg = torch.randint(0, V, (R,), generator=torch.Generator().manual_seed(1))
    tg[::4] = -100
    lgr = lg.clone().requires_grad_(True)
    ref = F.cross_entropy(lgr, tg.to(dev), ignore_index=-100)
    ref.backward()
    loss, dl = ops.cross_entropy_fwd_bwd(lg, tg.to(dev), 1056)
    assert abs(float(loss) - float(ref)) < 1e-4
    assert rel(dl[:, :V], lgr.grad) < 4e-3 and bool((dl[:, V:] == 0).all())


def test_epilogue_aux_modes(dev):
    from magma_amd import ops
    M, N, K = 130, 264, 128
    a = rnd(M, K, dev=dev, seed=11).to(BF16)
    w = rnd(N, K, dev=dev, seed=12, scale=0.1).to(BF16)
    aux = rnd(M, N, dev=dev, seed=13).to(BF16)
    r0 = rnd(M, N, dev=dev, seed=14).to(BF16)
    lin = ops.PackedLinear(w, bias=rnd(N, dev=dev, seed=15))
    acc = a.float() @ w.float().t() + lin.bias
    out = ops.gemm(a, lin, aux=aux, aux_mode=ops.MG_AUX_RELU_GATE, residuals=(r0,))
    assert rel(out, acc * (aux.float() > 0) + r0.float()) < 4e-3
    out = ops.gemm(a, lin, aux=aux, aux_mode=ops.MG_AUX_RELU_GATE, residuals=(r0,), aux_after=True)
    assert rel(out, (acc + r0.float()) * (aux.float() > 0)) < 4e-3
    xa = aux.float().requires_grad_(True)
    gl = 0.5 * xa * (1 + torch.tanh(math.sqrt(2 / math.pi) * (xa + 0.044715 * xa ** 3)))
    gl.sum().backward()
    out = ops.gemm(a, lin, aux=aux, aux_mode=ops.MG_AUX_GELU_GRAD)
    assert rel(out, acc * xa.grad) < 4e-3
    out = ops.gemm(a, lin, aux=aux, aux_mode=ops.MG_AUX_MUL)
    assert rel(out, acc * aux.float()) < 4e-3
    pre = torch.empty(M, N, dtype=BF16, device=dev)
    out = ops.gemm(a, lin, act=ops.MG_ACT_GELU_NEW, out2=pre)
    assert rel(pre, acc) < 4e-3 and rel(out, F.gelu(acc, approximate="tanh")) < 4e-3


@pytest.mark.parametrize("B,H,S", [(1, 1, 64), (2, 2, 57), (1, 2, 152), (1, 1, 1), (1, 2, 300), (2, 1, 385), (1, 1, 1024)])
def test_attention_backward(dev, B, H, S):
    from magma_amd import ops
    d = H * 256
    q = rnd(B, H, S, 256, dev=dev, seed=20, scale=0.5).to(BF16)
    k = rnd(B, H, S, 256, dev=dev, seed=21, scale=0.5).to(BF16)
    v = rnd(B, H, S, 256, dev=dev, seed=22).to(BF16)
    dO = rnd(B * S, d, dev=dev, seed=23).to(BF16)
    vt = ops.head_transpose(v, B, H, S, sb=H * S * 256, ss=256, sh=S * 256)
    out = torch.empty(B * S, d, dtype=BF16, device=dev)
    lse = torch.empty(B, H, S, dtype=torch.float32, device=dev)
    ops.attn_prefill(q, k, vt, out, B, H, S, lse=lse)
    qt = ops.head_transpose(q, B, H, S, sb=H * S * 256, ss=256, sh=S * 256)
    kt = ops.head_transpose(k, B, H, S, sb=H * S * 256, ss=256, sh=S * 256)
    dOt = ops.head_transpose(dO, B, H, S, sb=S * d, ss=d, sh=256)
    dq, dk, dv = ops.attn_bwd(q, k, v, qt, kt, dO, dOt, out, lse, B, H, S)
    qf, kf, vf = (t.float().requires_grad_(True) for t in (q, k, v))
    sc = qf @ kf.transpose(-1, -2) / 16.0
    sc = sc.masked_fill(~torch.ones(S, S, dtype=torch.bool, device=dev).tril(), float("-inf"))
    o = (torch.softmax(sc, -1) @ vf).permute(0, 2, 1, 3).reshape(B * S, d)
    o.backward(dO.float())
    for got, ref, name in ((dq, qf.grad, "dq"), (dk, kf.grad, "dk"), (dv, vf.grad, "dv")):
        if float(ref.abs().max()) < 1e-6:      # S = 1: softmax over one key has zero gradient
            assert float(got.float().abs().max()) < 1e-6, name
        else:
            assert rel(got, ref) < 1.5e-2, (name, rel(got, ref))


@pytest.mark.parametrize("variant", [0, 1, 2, 3, 4, 5, 6, 7, 8])
@pytest.mark.parametrize("B,H,S", [(1, 1, 1), (2, 2, 57), (1, 2, 300), (2, 1, 385), (1, 1, 1024)])
def test_attention_backward_kernel_variants(dev, monkeypatch, variant, B, H, S):
    """MAGMA_ATTN_BWD = 0 (three 16-row-wave kernels) / 1, 2 (dK and dV in one 32-key-wave kernel, S and dP computed once)
    / 3, 4 (32-query-wave dQ as well): each against fp32 autograd of the same attention, separate [B,H,S,256] outputs and
    the merged dqkv output with the inverse rotary (whole-row stores through the LDS staging image)."""
    from magma_amd import ops
    monkeypatch.setenv("MAGMA_ATTN_BWD", str(variant))
    d = H * 256
    q = rnd(B, H, S, 256, dev=dev, seed=50, scale=0.5).to(BF16)
    k = rnd(B, H, S, 256, dev=dev, seed=51, scale=0.5).to(BF16)
    v = rnd(B, H, S, 256, dev=dev, seed=52).to(BF16)
    dO = rnd(B * S, d, dev=dev, seed=53).to(BF16)
    vt = ops.head_transpose(v, B, H, S, sb=H * S * 256, ss=256, sh=S * 256)
    out = torch.empty(B * S, d, dtype=BF16, device=dev)
    lse = torch.empty(B, H, S, dtype=torch.float32, device=dev)
    ops.attn_prefill(q, k, vt, out, B, H, S, lse=lse)
    qt = ops.head_transpose(q, B, H, S, sb=H * S * 256, ss=256, sh=S * 256)
    kt = ops.head_transpose(k, B, H, S, sb=H * S * 256, ss=256, sh=S * 256)
    dOt = ops.head_transpose(dO, B, H, S, sb=S * d, ss=d, sh=256)
    dq, dk, dv = ops.attn_bwd(q, k, v, qt, kt, dO, dOt, out, lse, B, H, S)
    qf, kf, vf = (t.float().requires_grad_(True) for t in (q, k, v))
    sc = qf @ kf.transpose(-1, -2) / 16.0
    sc = sc.masked_fill(~torch.ones(S, S, dtype=torch.bool, device=dev).tril(), float("-inf"))
    o = (torch.softmax(sc, -1) @ vf).permute(0, 2, 1, 3).reshape(B * S, d)
    o.backward(dO.float())
    for got, ref, name in ((dq, qf.grad, "dq"), (dk, kf.grad, "dk"), (dv, vf.grad, "dv")):
        if float(ref.abs().max()) < 1e-6:      # S = 1: softmax over one key has zero gradient
            assert float(got.float().abs().max()) < 1e-6, name
        else:
            assert rel(got, ref) < 1.5e-2, (variant, name, rel(got, ref))
    rot = 64
    inv = 1.0 / (10000 ** (torch.arange(0, rot, 2, dtype=torch.float32, device=dev) / rot))
    ang = torch.arange(S + 3, dtype=torch.float32, device=dev)[:, None] * inv[None, :]
    sin_t, cos_t = ang.sin().contiguous(), ang.cos().contiguous()
    two_pass = ops.rotary_merge_bwd(dq, dk, dv, B, S, H, rot, sin_t, cos_t)
    merged = ops.attn_bwd_merged(q, k, v, qt, kt, dO, out, lse, B, H, S, rot, sin_t, cos_t)
    assert torch.equal(merged[:, 2 * d:], two_pass[:, 2 * d:])                     # dv: no rotary, same rounding
    if S > 1:
        for sl in (slice(0, d), slice(d, 2 * d)):
            assert rel(merged[:, sl], two_pass[:, sl]) < 6e-3


@pytest.mark.parametrize("B,H,S", [(2, 2, 57), (1, 2, 300)])
def test_attention_backward_merged_output(dev, B, H, S):
    """mg_attn_bwd_merged_bf16 (gradients written straight into dqkv with the inverse rotary) == mg_attn_bwd_bf16 +
    mg_rotary_merge_bwd_bf16, up to the second bf16 rounding the two-pass form has."""
    from magma_amd import ops
    d = H * 256
    q = rnd(B, H, S, 256, dev=dev, seed=30, scale=0.5).to(BF16)
    k = rnd(B, H, S, 256, dev=dev, seed=31, scale=0.5).to(BF16)
    v = rnd(B, H, S, 256, dev=dev, seed=32).to(BF16)
    dO = rnd(B * S, d, dev=dev, seed=33).to(BF16)
    vt = ops.head_transpose(v, B, H, S, sb=H * S * 256, ss=256, sh=S * 256)
    out = torch.empty(B * S, d, dtype=BF16, device=dev)
    lse = torch.empty(B, H, S, dtype=torch.float32, device=dev)
    ops.attn_prefill(q, k, vt, out, B, H, S, lse=lse)
    qt = ops.head_transpose(q, B, H, S, sb=H * S * 256, ss=256, sh=S * 256)
    kt = ops.head_transpose(k, B, H, S, sb=H * S * 256, ss=256, sh=S * 256)
    dOt = ops.head_transpose(dO, B, H, S, sb=S * d, ss=d, sh=256)
    rot = 64
    inv = 1.0 / (10000 ** (torch.arange(0, rot, 2, dtype=torch.float32, device=dev) / rot))
    ang = torch.arange(S + 3, dtype=torch.float32, device=dev)[:, None] * inv[None, :]
    sin_t, cos_t = ang.sin().contiguous(), ang.cos().contiguous()
    dq, dk, dv = ops.attn_bwd(q, k, v, qt, kt, dO, dOt, out, lse, B, H, S)
    two_pass = ops.rotary_merge_bwd(dq, dk, dv, B, S, H, rot, sin_t, cos_t)
    merged = ops.attn_bwd_merged(q, k, v, qt, kt, dO, out, lse, B, H, S, rot, sin_t, cos_t)   # transposes dO itself
    assert merged.shape == two_pass.shape == (B * S, 3 * d)
    assert torch.equal(merged[:, 2 * d:], two_pass[:, 2 * d:])                     # dv: no rotary, same rounding
    for name, sl in (("dq", slice(0, d)), ("dk", slice(d, 2 * d))):
        a, b = merged[:, sl].view(B * S, H, 256), two_pass[:, sl].view(B * S, H, 256)
        # columns outside the rotary: same arithmetic except the summation order of D = rowsum(dO o O) (fused prep)
        assert rel(a[..., rot:], b[..., rot:]) < 2e-3, (name, rel(a[..., rot:], b[..., rot:]))
        assert rel(a[..., :rot], b[..., :rot]) < 6e-3, (name, rel(a[..., :rot], b[..., :rot]))
    # O at the row stride of a wider buffer (the [ctx | t] operand of the training step's [W_out | W_up] GEMM): bit-identical
    wide = torch.full((B * S, d + 136), float("nan"), dtype=BF16, device=dev)
    wide[:, :d] = out
    assert torch.equal(ops.attn_bwd_merged(q, k, v, qt, kt, dO, wide[:, :d], lse, B, H, S, rot, sin_t, cos_t), merged)
    for got, ref in zip(ops.attn_bwd(q, k, v, qt, kt, dO, dOt, wide[:, :d], lse, B, H, S), (dq, dk, dv)):
        assert torch.equal(got, ref)


def test_rotary_split_train_emits_all_transposes(dev):
    """mg_rotary_split_train_bf16 == mg_rotary_split_bf16 + two mg_head_transpose_bf16, bit for bit."""
    from magma_amd import ops
    B, H, S, rot = 2, 3, 75, 64
    d = H * 256
    qkv = rnd(B * S, 3 * d, dev=dev, seed=40).to(BF16)
    inv = 1.0 / (10000 ** (torch.arange(0, rot, 2, dtype=torch.float32, device=dev) / rot))
    ang = torch.arange(S + 5, dtype=torch.float32, device=dev)[:, None] * inv[None, :]
    sin_t, cos_t = ang.sin().contiguous(), ang.cos().contiguous()
    ld = ops.ceil_to(S, 32)
    mk = lambda: torch.empty(B, H, S, 256, dtype=BF16, device=dev)
    mt = lambda: torch.full((B, H, ld // 32, 256, 32), 7.0, dtype=BF16, device=dev)
    q0, k0, v0, vt0 = mk(), mk(), mk(), mt()
    ops.rotary_split(qkv, B, S, H, rot, sin_t, cos_t, q0, k0, v0, pos0=0, vt=vt0)
    qt0 = ops.head_transpose(q0, B, H, S, sb=H * S * 256, ss=256, sh=S * 256)
    kt0 = ops.head_transpose(k0, B, H, S, sb=H * S * 256, ss=256, sh=S * 256)
    q1, k1, v1, vt1, qt1, kt1 = mk(), mk(), mk(), mt(), mt(), mt()
    ops.rotary_split_train(qkv, B, S, H, rot, sin_t, cos_t, q1, k1, v1, vt1, qt1, kt1)
    for a, b, name in ((q1, q0, "q"), (k1, k0, "k"), (v1, v0, "v"), (vt1, vt0, "vt"), (qt1, qt0, "qt"), (kt1, kt0, "kt")):
        assert torch.equal(a, b), name


def test_rotary_qk_inplace_equals_the_split_pass(dev):
    """mg_rotary_qk_inplace_bf16 leaves in the q / k sections of qkv exactly what mg_rotary_split_bf16 writes to q / kcache (same
    fp32 arithmetic, one bf16 rounding), v untouched; also at a row stride wider than 3 H 256."""
    from magma_amd import ops
    B, H, S, rot = 2, 3, 75, 64
    d = H * 256
    wide = rnd(B * S, 3 * d + 40, dev=dev, seed=41).to(BF16)
    qkv = wide[:, :3 * d]
    inv = 1.0 / (10000 ** (torch.arange(0, rot, 2, dtype=torch.float32, device=dev) / rot))
    ang = torch.arange(S + 5, dtype=torch.float32, device=dev)[:, None] * inv[None, :]
    sin_t, cos_t = ang.sin().contiguous(), ang.cos().contiguous()
    mk = lambda: torch.empty(B, H, S, 256, dtype=BF16, device=dev)
    q0, k0, v0 = mk(), mk(), mk()
    ops.rotary_split(qkv, B, S, H, rot, sin_t, cos_t, q0, k0, v0, pos0=0)
    before = wide.clone()
    ops.rotary_qk_inplace(qkv, B, S, H, rot, sin_t, cos_t)
    got = qkv.reshape(B, S, 3, H, 256).permute(2, 0, 3, 1, 4)
    assert torch.equal(got[0], q0) and torch.equal(got[1], k0) and torch.equal(got[2], v0)
    assert torch.equal(wide[:, 3 * d:], before[:, 3 * d:]) and torch.equal(wide[:, 2 * d:3 * d], before[:, 2 * d:3 * d])


@pytest.mark.parametrize("B,H,S", [(1, 1, 1), (2, 2, 57), (1, 2, 300), (2, 1, 385), (1, 1, 1024), (1, 3, 129)])
def test_attention_without_transposed_images(dev, B, H, S):
    """mg_attn_fwd_rows_bf16 / mg_attn_bwd_rows_bf16 (csrc/attention_tr.hip: V^T, Q^T, dO^T, K^T fragments read from the ROW images
    with ds_read_b64_tr_b16): forward and all three gradients against fp32 autograd of the same attention; against the kernels with
    transposed images (same MFMA order: the outputs agree to the last bit or nearly); and reading q / k / v straight from the fused qkv
    activation [B*S, 3 H 256] gives the SAME bits as reading [B,H,S,256] tensors -- separate and merged (inverse rotary) outputs."""
    from magma_amd import ops
    d = H * 256
    q = rnd(B, H, S, 256, dev=dev, seed=60, scale=0.5).to(BF16)
    k = rnd(B, H, S, 256, dev=dev, seed=61, scale=0.5).to(BF16)
    v = rnd(B, H, S, 256, dev=dev, seed=62).to(BF16)
    dO = rnd(B * S, d, dev=dev, seed=63).to(BF16)
    x = ops.AttnRows.of_bhsd(q, k, v)
    out = torch.empty(B * S, d, dtype=BF16, device=dev)
    lse = torch.empty(B, H, S, dtype=torch.float32, device=dev)
    ops.attn_fwd_rows(x, out, lse=lse)
    # the same operands as column ranges of one fused activation (what the training engine hands over), wider row stride
    fused = torch.zeros(B * S, 3 * d + 24, dtype=BF16, device=dev)
    fused[:, :3 * d] = torch.stack((q, k, v)).permute(1, 3, 0, 2, 4).reshape(B * S, 3 * d)
    xf = ops.AttnRows.of_qkv(fused[:, :3 * d], B, S, H)
    out_f = torch.full((B * S, d + 8), float("nan"), dtype=BF16, device=dev)
    lse_f = torch.empty_like(lse)
    ops.attn_fwd_rows(xf, out_f[:, :d], lse=lse_f)
    assert torch.equal(out_f[:, :d], out) and torch.equal(lse_f, lse)
    # the kernels with transposed images
    vt = ops.head_transpose(v, B, H, S, sb=H * S * 256, ss=256, sh=S * 256)
    out_old = torch.empty_like(out)
    lse_old = torch.empty_like(lse)
    ops.attn_prefill(q, k, vt, out_old, B, H, S, lse=lse_old)
    assert rel(out, out_old) < 1e-3 and float((lse - lse_old).abs().max()) < 1e-4
    qf, kf, vf = (t.float().requires_grad_(True) for t in (q, k, v))
    sc = qf @ kf.transpose(-1, -2) / 16.0
    sc = sc.masked_fill(~torch.ones(S, S, dtype=torch.bool, device=dev).tril(), float("-inf"))
    o = (torch.softmax(sc, -1) @ vf).permute(0, 2, 1, 3).reshape(B * S, d)
    assert rel(out, o.detach()) < 1e-2
    o.backward(dO.float())
    dq, dk, dv = ops.attn_bwd_rows(x, dO, out, lse)
    for got, ref, name in ((dq, qf.grad, "dq"), (dk, kf.grad, "dk"), (dv, vf.grad, "dv")):
        if float(ref.abs().max()) < 1e-6:      # S = 1: softmax over one key has zero gradient
            assert float(got.float().abs().max()) < 1e-6, name
        else:
            assert rel(got, ref) < 1.5e-2, (name, rel(got, ref))
    for a, b_ in zip(ops.attn_bwd_rows(xf, dO, out_f[:, :d], lse), (dq, dk, dv)):
        assert torch.equal(a, b_)
    rot = 64
    inv = 1.0 / (10000 ** (torch.arange(0, rot, 2, dtype=torch.float32, device=dev) / rot))
    ang = torch.arange(S + 3, dtype=torch.float32, device=dev)[:, None] * inv[None, :]
    sin_t, cos_t = ang.sin().contiguous(), ang.cos().contiguous()
    two_pass = ops.rotary_merge_bwd(dq, dk, dv, B, S, H, rot, sin_t, cos_t)
    merged = ops.attn_bwd_rows(xf, dO, out, lse, merged_rot=(rot, sin_t, cos_t))
    assert merged.shape == (B * S, 3 * d) and torch.equal(merged[:, 2 * d:], two_pass[:, 2 * d:])
    if S > 1:
        for sl in (slice(0, d), slice(d, 2 * d)):
            assert rel(merged[:, sl], two_pass[:, sl]) < 6e-3
    # BASELINE config[4]: the epilogues' OCP MX e4m3 copy of dqkv (operand of the qkv dgrad's MX GEMM) == mg_quantize_mx_fp8(dqkv) bit
    # for bit, next to the bf16 output and on its own (no_out)
    mx = ops.mx_empty(B * S, 3 * d, dev)
    merged_b = ops.attn_bwd_rows(xf, dO, out, lse, merged_rot=(rot, sin_t, cos_t), mx_out=mx)
    qref, sref = ops.quantize_mx_fp8(merged)
    assert torch.equal(merged_b, merged) and torch.equal(mx[0], qref) and torch.equal(mx[1], sref)
    mx2 = ops.mx_empty(B * S, 3 * d, dev)
    assert ops.attn_bwd_rows(xf, dO, out, lse, merged_rot=(rot, sin_t, cos_t), mx_out=mx2, no_out=True) is None
    assert torch.equal(mx2[0], qref) and torch.equal(mx2[1], sref)
    # first_rows (the bottom block of a frozen LM: only the first positions' gradients are wanted): the first ceil(P / 128) query /
    # key blocks carry the same bits as the full launch -- dK / dV of those keys still sum over EVERY later query
    for P in (1, 49, 144, 200):
        if P >= S:
            continue
        n = min(S, (P + 127) // 128 * 128)
        for a, b_ in zip(ops.attn_bwd_rows(x, dO, out, lse, first_rows=P), (dq, dk, dv)):
            assert torch.equal(a[:, :, :n], b_[:, :, :n]), P
        part = ops.attn_bwd_rows(xf, dO, out, lse, merged_rot=(rot, sin_t, cos_t), first_rows=P)
        assert torch.equal(part.view(B, S, 3 * d)[:, :n], merged.view(B, S, 3 * d)[:, :n]), P


def test_rotary_merge_bwd(dev):
    from magma_amd import ops
    from oracle.model import apply_rotary, rotary_tables
    B, S, H = 2, 37, 2
    sin_t, cos_t = rotary_tables(64, 64)
    dq, dk, dv = (rnd(B, H, S, 256, dev=dev, seed=30 + i).to(BF16) for i in range(3))
    out = ops.rotary_merge_bwd(dq, dk, dv, B, S, H, 64, sin_t.to(dev).contiguous(), cos_t.to(dev).contiguous())
    x = torch.zeros(B, S, H, 256, requires_grad=True)
    y = apply_rotary(x, torch.arange(S), 64)
    y.backward(dq.float().cpu().permute(0, 2, 1, 3))
    got = out.view(B, S, 3, H, 256).float().cpu()
    assert rel(got[:, :, 0], x.grad) < 4e-3
    assert torch.equal(got[:, :, 2], dv.float().cpu().permute(0, 2, 1, 3))


def test_conv_backward_helpers(dev):
    from magma_amd import ops
    B, H, W, C = 2, 6, 8, 16
    dy = rnd(B, H // 2, W // 2, C, dev=dev, seed=40).to(BF16)
    dx = ops.avgpool2_bwd(dy, B, H, W, C)
    xr = torch.zeros(B, C, H, W, device=dev, requires_grad=True)
    F.avg_pool2d(xr, 2).backward(dy.float().permute(0, 3, 1, 2))
    assert rel(dx, xr.grad.permute(0, 2, 3, 1)) < 4e-3
    a, b, gt = (rnd(64, 24, dev=dev, seed=41 + i).to(BF16) for i in range(3))
    assert rel(ops.add_gate(a, b, gt), (a.float() + b.float()) * (gt.float() > 0)) < 4e-3
    assert torch.equal(ops.add_gate(a), a)
    # im2col^T x dY^T == conv weight gradient
    x = rnd(B, H, W, C, dev=dev, seed=44).to(BF16)
    cols_t = ops.im2col_t(x, B, H, W, C)
    ref = F.unfold(x.float().permute(0, 3, 1, 2), 3, padding=1)          # [B, C*9, HW], row = c*9 + tap
    ref = ref.permute(1, 0, 2).reshape(9 * C, B * H * W)
    assert torch.equal(cols_t.float()[:, : B * H * W], ref)
    # frozen-stat BN affine grads
    M = B * H * W
    g = rnd(M, C, dev=dev, seed=45).to(BF16)
    y = rnd(M, C, dev=dev, seed=46).to(BF16)
    sub = rnd(M, C, dev=dev, seed=47).to(BF16)
    gamma, beta = rnd(C, dev=dev, seed=48) + 2, rnd(C, dev=dev, seed=49)
    dg, db = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
    ops.bn_param_grad(g, y, sub, gamma, beta, dg, db)
    assert rel(db, g.float().sum(0)) < 1e-4
    assert rel(dg, (g.float() * (y.float() - sub.float() - beta) / gamma).sum(0)) < 1e-4
    # ragged row counts around the kernel's 16-row groups / 256-row blocks, with and without the residual operand, C > 512
    for M2, C2, with_sub in ((1, 8, True), (267, 520, False), (513 + 7, 40, True)):
        g2, y2 = rnd(M2, C2, dev=dev, seed=52).to(BF16), rnd(M2, C2, dev=dev, seed=53).to(BF16)
        s2 = rnd(M2, C2, dev=dev, seed=54).to(BF16) if with_sub else None
        gm, bt = rnd(C2, dev=dev, seed=55) + 2, rnd(C2, dev=dev, seed=56)
        dg2, db2 = torch.zeros(C2, device=dev), torch.zeros(C2, device=dev)
        ops.bn_param_grad(g2, y2, s2, gm, bt, dg2, db2)
        yy = y2.float() - (s2.float() if with_sub else 0)
        assert rel(db2, g2.float().sum(0)) < 1e-4 and rel(dg2, (g2.float() * (yy - bt) / gm).sum(0)) < 1e-4, (M2, C2)
        # the same sums taken while g is transposed for the weight-gradient GEMM (one pass over g): the transpose bit for bit, the
        # sums accumulated ON TOP of what the buffers hold
        dg3, db3 = torch.full((C2,), 3.0, device=dev), torch.full((C2,), -2.0, device=dev)
        gt = ops.transpose_bn_param_grad(g2, y2, s2, gm, bt, dg3, db3)
        assert torch.equal(gt, ops.transpose(g2))
        assert rel(db3 + 2.0, g2.float().sum(0)) < 1e-4 and rel(dg3 - 3.0, (g2.float() * (yy - bt) / gm).sum(0)) < 1e-4, (M2, C2)


def test_adamw_and_clip(dev):
    from magma_amd import ops
    n = 10007
    p0 = rnd(n, dev=dev, seed=50)
    g = rnd(n, dev=dev, seed=51) * 3
    ref_p = p0.clone().requires_grad_(True)
    opt = torch.optim.AdamW([ref_p], lr=1e-2, betas=(0.9, 0.95), eps=1e-8, weight_decay=0.1)
    p, m, v = p0.clone(), torch.zeros(n, device=dev), torch.zeros(n, device=dev)
    pb = torch.empty(n, dtype=BF16, device=dev)
    for step in (1, 2, 3):
        ref_p.grad = g.clone()
        torch.nn.utils.clip_grad_norm_([ref_p], 1.0)
        opt.step()
        nsq = torch.zeros(1, device=dev)
        ops.sumsq(g, nsq)
        ops.adamw(p, m, v, g, pb, 1e-2, 0.9, 0.95, 1e-8, 0.1, step, max_norm=1.0, norm_sq=nsq)
    assert rel(p, ref_p.detach()) < 1e-5
    assert torch.equal(pb, p.to(BF16))


@pytest.mark.parametrize("cout,cin,k", [(24, 16, 3), (96, 48, 3), (40, 72, 1), (768, 768, 3)])
def test_conv_weight_relayout(dev, cout, cin, k):
    """bit-exact against the PyTorch expressions it replaces in the training engine"""
    from magma_amd import ops
    w = rnd(cout, cin, k, k, dev=dev, seed=50, scale=0.1).to(BF16)
    scale = rnd(cout, dev=dev, seed=51).abs() + 0.5
    f = ops.conv_weight_relayout(w, 0)
    ref_f = w.permute(0, 2, 3, 1).reshape(cout, k * k * cin)
    assert f.shape[1] % 64 == 0 and torch.equal(f[:, : k * k * cin], ref_f) and bool((f[:, k * k * cin:] == 0).all())
    d = ops.conv_weight_relayout(w, 1, scale)
    ref_d = (w.float() * scale.view(cout, 1, 1, 1)).flip(2, 3).permute(1, 2, 3, 0).reshape(cin, k * k * cout).to(BF16)
    assert d.shape[1] % 64 == 0 and torch.equal(d[:, : k * k * cout], ref_d) and bool((d[:, k * k * cout:] == 0).all())


def test_conv_operand_plan_matches_the_single_calls(dev):
    """mg_conv_weight_relayout_batch / mg_bn_fold_batch (ops.ConvOperandPlan: every convolution of the CLIP trunk re-derived by two
    launches per step) == mg_conv_weight_relayout_bf16 / mg_bn_fold_f32 per convolution, bit for bit -- 1x1 and 3x3, ragged
    channel counts, a 1x1 whose Cin is a multiple of 64 (no forward copy: the weight is its own operand); refresh() after the
    weights changed in place picks the new values up."""
    from magma_amd import ops
    shapes = [(24, 16, 3), (96, 48, 3), (40, 72, 1), (128, 64, 1), (384, 384, 3), (8, 8, 1), (3072, 768, 1)]
    units = []
    for i, (cout, cin, k) in enumerate(shapes):
        w = rnd(cout, cin, k, k, dev=dev, seed=600 + i, scale=0.1).to(BF16).contiguous()
        gamma, beta = rnd(cout, dev=dev, seed=620 + i) + 2, rnd(cout, dev=dev, seed=640 + i)
        mean, var = rnd(cout, dev=dev, seed=660 + i), rnd(cout, dev=dev, seed=680 + i).abs() + 0.1
        units.append((w, gamma.contiguous(), beta.contiguous(), mean.contiguous(), var.contiguous(), 1e-5 * (i + 1)))
    plan = ops.ConvOperandPlan(units, dev)
    for rnd_no in range(2):
        plan.refresh()
        for i, (w, gamma, beta, mean, var, eps) in enumerate(units):
            sc, sh = ops.bn_fold(gamma, beta, mean, var, eps)
            assert torch.equal(plan.scale[i], sc) and torch.equal(plan.shift[i], sh), i
            cout, cin, k, _ = w.shape
            f = ops.conv_weight_relayout(w, 0)
            if k == 1 and cin % 64 == 0:
                assert plan.fwd[i].data_ptr() == w.data_ptr() and torch.equal(plan.fwd[i], f[:, :cin])
            else:
                assert torch.equal(plan.fwd[i], f), i
            assert torch.equal(plan.dgrad[i], ops.conv_weight_relayout(w, 1, sc)), i
        for (w, gamma, *_r) in units:          # an optimizer step: same storage, new values
            w.mul_(1.5)
            gamma.add_(0.25)


@pytest.mark.parametrize("R,C", [(4096, 1024), (1000, 520), (77, 64)])
def test_transpose_colsum(dev, R, C):
    """transpose + column sums in one pass == transpose and colsum separately (bias gradient and weight-gradient operand of a Linear)."""
    from magma_amd import ops
    g = torch.Generator(device="cpu").manual_seed(R + C)
    x = torch.randn(R, C, generator=g).to(torch.bfloat16).to(dev)
    acc = torch.full((C,), 0.5, dtype=torch.float32, device=dev)           # accumulates on top of what is there
    xt = ops.transpose_colsum(x, acc)
    assert torch.equal(xt[:, :R], x.t())
    assert bool((xt[:, R:] == 0).all())
    ref = 0.5 + x.float().sum(0)
    assert float((acc - ref).abs().max()) <= 1e-3 * float(ref.abs().max()) + 1e-3
