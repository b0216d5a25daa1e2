"""Kernel-level parity: every C-ABI entry point against a plain PyTorch fp32
reference of the same op, on seeded random (never zero-filled, never symmetric)
inputs.  Tolerances are stated per test.  All calls go through libmagma_hip.so."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

BF16 = torch.bfloat16


def rnd(*shape, dev, scale=1.0, seed=0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(dev)


def rel_err(a, b):
    a, b = a.float(), b.float()
    return float((a - b).norm() / (b.norm() + 1e-12))


def assert_close(got, ref, tol, what=""):
    e = rel_err(got, ref)
    mx = float((got.float() - ref.float()).abs().max())
    assert math.isfinite(e) and e < tol, f"{what}: rel-L2 {e:.3e} (max abs {mx:.3e}) >= {tol}"


# bf16 output rounding alone is ~2^-9 = 2e-3 rel per element; inputs are exact bf16
GEMM_TOL = 4e-3


@pytest.mark.parametrize("layout", ["rm", "ft"])
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (300, 200, 192), (77, 1056, 96), (1216, 512, 4096), (130, 48, 432)])
def test_gemm_dense(dev, layout, M, N, K):
    from magma_amd import ops
    a = rnd(M, K, dev=dev, seed=1).to(BF16)
    w = rnd(N, K, dev=dev, seed=2, scale=0.05).to(BF16)
    lin = ops.PackedLinear(w, tiled=True, rowmajor=True)
    out = ops.gemm(a, lin, layout=layout)
    ref = a.float() @ w.float().t()
    assert_close(out, ref, GEMM_TOL, f"gemm {layout} {M}x{N}x{K}")


def test_gemm_transpose_detecting(dev):
    """A = I (rectangular pad) with an asymmetric W catches swapped C layouts."""
    from magma_amd import ops
    M = N = K = 128
    a = torch.eye(M, K, device=dev).to(BF16)
    w = (torch.arange(N * K, device=dev).float().reshape(N, K) % 251 - 125.0) / 128.0
    lin = ops.PackedLinear(w.to(BF16), tiled=True, rowmajor=True)
    for layout in ("rm", "ft"):
        out = ops.gemm(a, lin, layout=layout)
        assert torch.equal(out.float(), w.to(BF16).float().t().contiguous()), layout


def test_gemm_epilogue(dev):
    from magma_amd import ops
    M, N, K = 200, 328, 256
    a = rnd(M, K, dev=dev, seed=3).to(BF16)
    w = rnd(N, K, dev=dev, seed=4, scale=0.05).to(BF16)
    bias = rnd(N, dev=dev, seed=5)
    scale = rnd(N, dev=dev, seed=6).abs() + 0.5
    r0, r1, r2 = (rnd(M, N, dev=dev, seed=7 + i).to(BF16) for i in range(3))
    lin = ops.PackedLinear(w, bias=bias)
    acc = a.float() @ w.float().t()
    # bias + gelu_new
    out = ops.gemm(a, lin, act=ops.MG_ACT_GELU_NEW)
    x = acc + bias
    ref = 0.5 * x * (1 + torch.tanh(math.sqrt(2 / math.pi) * (x + 0.044715 * x ** 3)))
    assert_close(out, ref, GEMM_TOL, "bias+gelu")
    # scale + bias + relu  (folded BatchNorm)
    out = ops.gemm(a, lin, act=ops.MG_ACT_RELU, scale=scale)
    assert_close(out, F.relu(acc * scale + bias), GEMM_TOL, "scale+bias+relu")
    # bias + 3 residuals (GPT-J block sum), fp32 out
    out = ops.gemm(a, lin, residuals=(r0, r1, r2), out_dtype=torch.float32)
    assert out.dtype == torch.float32
    assert_close(out, acc + bias + r0.float() + r1.float() + r2.float(), 1e-4, "bias+3res f32")
    # scale+bias, residual, relu after (bottleneck tail)
    out = ops.gemm(a, lin, scale=scale, residuals=(r0,), act_after=ops.MG_ACT_RELU)
    assert_close(out, F.relu(acc * scale + bias + r0.float()), GEMM_TOL, "bn+identity+relu")
    # N not a multiple of 4 (vocab 50258 style), fp32 logits into a padded buffer
    N2 = 203
    lin2 = ops.PackedLinear(w[:N2], bias=bias[:N2])
    buf = torch.full((M, 208), 7.0, dtype=torch.float32, device=dev)
    ops.gemm(a, lin2, out=buf)
    assert_close(buf[:, :N2], acc[:, :N2] + bias[:N2], 1e-4, "odd N")
    assert bool((buf[:, N2:] == 7.0).all()), "wrote past N"
    # row stride that is a multiple of 4 but not of 8: the epilogue must fall back from 16-byte to 8-byte accesses
    for Mx, tile in ((M, 128), (1100, 256)):
        ax = rnd(Mx, K, dev=dev, seed=31).to(BF16)
        rx = rnd(Mx, 332, dev=dev, seed=32).to(BF16)
        bufx = torch.zeros(Mx, 332, dtype=BF16, device=dev)
        ops.gemm(ax, lin, out=bufx[:, :N], residuals=(rx,), tile=tile)
        assert_close(bufx[:, :N], ax.float() @ w.float().t() + bias + rx[:, :N].float(), GEMM_TOL, f"ldc%8!=0 tile{tile}")
        assert bool((bufx[:, N:] == 0).all())


@pytest.mark.parametrize("layout", ["rm", "ft"])
@pytest.mark.parametrize("B,H,W,Cin,Cout", [(2, 12, 10, 16, 24), (1, 7, 9, 48, 96), (2, 24, 24, 96, 96)])
def test_conv3x3(dev, layout, B, H, W, Cin, Cout):
    from magma_amd import ops
    x = rnd(B, Cin, H, W, dev=dev, seed=11).to(BF16)
    w = rnd(Cout, Cin, 3, 3, dev=dev, seed=12, scale=0.1).to(BF16)
    x_nhwc = x.permute(0, 2, 3, 1).contiguous()
    lin = ops.PackedLinear(w.permute(0, 2, 3, 1).reshape(Cout, 9 * Cin).contiguous(), tiled=True, rowmajor=True)
    out = ops.gemm(x_nhwc.view(B * H * W, Cin), lin, conv=(H, W, Cin), layout=layout)
    ref = F.conv2d(x.float(), w.float(), padding=1).permute(0, 2, 3, 1).reshape(B * H * W, Cout)
    assert_close(out, ref, GEMM_TOL, f"conv3x3 {layout}")


@pytest.mark.parametrize("M", [1, 8, 16])
@pytest.mark.parametrize("N,K,variant", [
    (4096, 4096, 0), (1056, 4096, 0), (1024, 1024, 0), (512, 512, 0), (208, 64, 0), (50258, 512, 0),
    (4096, 4096, 1 | 8 << 4 | 8 << 8), (4096, 4096, 2 | 4 << 4 | 16 << 8), (4096, 16384, 4 | 8 << 4 | 8 << 8),
    (4096, 4096, 2 | 8 << 4 | 16 << 8), (4096, 4096, 1 | 4 << 4 | 16 << 8), (4096, 4096, 4 | 4 << 4 | 8 << 8),
    (4096, 4096, 2 | 4 << 4 | 8 << 8), (4096, 4096, 2 | 8 << 4 | 8 << 8), (4096, 1024, 4 | 8 << 4 | 4 << 8),
])
def test_gemm_skinny(dev, M, N, K, variant):
    from magma_amd import ops
    x = rnd(M, K, dev=dev, seed=21).to(BF16)
    w = rnd(N, K, dev=dev, seed=22, scale=0.05).to(BF16)
    bias = rnd(N, dev=dev, seed=23)
    res = rnd(M, (N + 7) // 8 * 8, dev=dev, seed=24).to(BF16)[:, :N]   # padded row stride (ldr % 4 == 0)
    lin = ops.PackedLinear(w, bias=bias)
    out = ops.gemm_skinny(x, lin, residuals=(res,), variant=variant)
    ref = x.float() @ w.float().t() + bias + res.float()
    assert_close(out, ref, GEMM_TOL, f"skinny M={M} {N}x{K} v={variant}")
    out32 = ops.gemm_skinny(x, lin, act=ops.MG_ACT_RELU, out_dtype=torch.float32, variant=variant)
    assert_close(out32, F.relu(x.float() @ w.float().t() + bias), 1e-4, "skinny relu f32")


@pytest.mark.parametrize("N,K,base", [(4096, 16384, 1 | 4 << 4 | 16 << 8), (4096, 8192, 1 | 4 << 4 | 16 << 8), (4096, 4096, 1 | 8 << 4 | 8 << 8),
                                      (1024, 4096, 1 | 8 << 4 | 4 << 8), (4096, 1024, 1 | 8 << 4 | 4 << 8), (4096, 2048, 1 | 4 << 4 | 16 << 8)])
def test_gemm_skinny_pipelined_bursts_are_bit_identical(dev, N, K, base):
    """skinny_body<..., PIPE>: double-buffered weight bursts, same MFMA order per accumulator -> the same bits as the drained
    variant, LayerNorm fold included (odd burst counts exercise the peeled tail: K = 2048 -> one burst per wave)."""
    from magma_amd import ops
    M = 8
    x = rnd(M, K, dev=dev, seed=31).to(BF16)
    w = rnd(N, K, dev=dev, seed=32, scale=0.05).to(BF16)
    lin = ops.PackedLinear(w, bias=rnd(N, dev=dev, seed=33))
    a = ops.gemm_skinny(x, lin, out_dtype=torch.float32, variant=base)
    b = ops.gemm_skinny(x, lin, out_dtype=torch.float32, variant=base | 1 << 16)
    assert torch.equal(a, b)
    assert_close(a, x.float() @ w.float().t() + lin.bias, 1e-4 * (K / 1024) ** 0.5 + 1e-4, "skinny f32")
    colsum = w.float().sum(1).contiguous()
    fold = (colsum, K, 1e-5)
    a = ops.gemm_skinny(x, lin, out_dtype=torch.float32, variant=base, ln_fold=fold)
    b = ops.gemm_skinny(x, lin, out_dtype=torch.float32, variant=base | 1 << 16, ln_fold=fold)
    assert torch.equal(a, b)


def test_tile_roundtrip(dev):
    from magma_amd import ops
    w = rnd(48, 128, dev=dev, seed=31).to(BF16)
    assert torch.equal(ops.PackedLinear.untile(ops.PackedLinear.tile(w)), w)


@pytest.mark.parametrize("rows,d", [(8, 4096), (37, 512), (3, 16384)])
def test_layernorm(dev, rows, d):
    from magma_amd import ops
    x = (rnd(rows, d, dev=dev, seed=41) * 2 + 0.3).to(BF16)
    g = rnd(d, dev=dev, seed=42) * 0.1 + 1
    b = rnd(d, dev=dev, seed=43) * 0.1
    out = ops.layernorm(x, g, b, 1e-5)
    ref = F.layer_norm(x.float(), (d,), g, b, 1e-5)
    assert_close(out, ref, 3e-3, "layernorm")


@pytest.mark.parametrize("rows,d", [(8195, 4096), (8192, 2048), (9001, 264), (8194, 2056)])
def test_layernorm_rows_per_workgroup_form(dev, rows, d):
    """Activations of >= 8192 rows (d <= 4096) take the kernel that normalises four rows per workgroup (gamma / beta loaded once per
    workgroup, the next row in flight): against the fp32 reference, and BIT-IDENTICAL to the one-row-per-workgroup kernel that the
    same rows take when they are handed over in pieces of < 8192 rows; ragged row count, both vector-per-thread instantiations,
    a row stride wider than d."""
    from magma_amd import ops
    wide = (rnd(rows, d + 24, dev=dev, seed=44) * 2 + 0.3).to(BF16)
    x = wide[:, :d]
    g = rnd(d, dev=dev, seed=45) * 0.1 + 1
    b = rnd(d, dev=dev, seed=46) * 0.1
    out = ops.layernorm(x, g, b, 1e-5)
    ref = F.layer_norm(x.float(), (d,), g, b, 1e-5)
    assert_close(out, ref, 3e-3, "layernorm, rows per workgroup")
    pieces = torch.cat([ops.layernorm(x[i:i + 4096], g, b, 1e-5) for i in range(0, rows, 4096)])
    assert torch.equal(out, pieces)


def test_embedding(dev):
    from magma_amd import ops
    V, d, B, T = 1056, 512, 3, 7
    wte = rnd(V, d, dev=dev, seed=51).to(BF16)
    ids = torch.randint(0, V, (B, T), generator=torch.Generator().manual_seed(5)).to(dev)
    out = torch.zeros(B, 12, d, dtype=BF16, device=dev)
    ops.embedding(ids, wte, out, row_off=4)
    assert torch.equal(out[:, 4:11], wte[ids])
    assert bool((out[:, :4] == 0).all()) and bool((out[:, 11:] == 0).all())


def _rotary_ref(x, pos, rot):
    from oracle.model import apply_rotary
    return apply_rotary(x.float().cpu(), pos.cpu(), rot)


@pytest.mark.parametrize("B,S,H", [(2, 57, 2), (1, 152, 3), (2, 33, 1)])
def test_rotary_split_and_prefill_attention(dev, B, S, H):
    """rotary/split vs the oracle's apply_rotary; flash attention vs fp32 softmax(QK^T/16)V."""
    from magma_amd import ops
    from oracle.model import rotary_tables
    d = H * 256
    Smax = 192
    qkv = rnd(B * S, 3 * d, dev=dev, seed=61).to(BF16)
    sin_t, cos_t = rotary_tables(64, Smax)
    sin_t, cos_t = sin_t.to(dev).contiguous(), cos_t.to(dev).contiguous()
    q = torch.empty(B, H, S, 256, dtype=BF16, device=dev)
    kc = torch.zeros(B, H, Smax, 256, dtype=BF16, device=dev)
    vc = torch.zeros(B, H, Smax, 256, dtype=BF16, device=dev)
    vt_ld = (S + 31) // 32 * 32
    vt = torch.full((B, H, vt_ld // 32, 256, 32), float("nan"), dtype=BF16, device=dev)   # V^T in 32-key tiles
    ops.rotary_split(qkv, B, S, H, 64, sin_t, cos_t, q, kc, vc, pos0=0, vt=vt)
    x = qkv.view(B, S, 3, H, 256).float().cpu()
    pos = torch.arange(S)
    q_ref = _rotary_ref(x[:, :, 0], pos, 64).permute(0, 2, 1, 3)
    k_ref = _rotary_ref(x[:, :, 1], pos, 64).permute(0, 2, 1, 3)
    v_ref = x[:, :, 2].permute(0, 2, 1, 3)
    assert_close(q.cpu(), q_ref, 3e-3, "q rotary")
    assert_close(kc[:, :, :S].cpu(), k_ref, 3e-3, "k rotary")
    assert torch.equal(vc[:, :, :S].float().cpu(), v_ref)
    vt_flat = vt.permute(0, 1, 3, 2, 4).reshape(B, H, 256, vt_ld)       # [b,h,tile,d,i] -> [b,h,d,32*tile+i]
    assert torch.equal(vt_flat[:, :, :, :S].float().cpu(), v_ref.transpose(2, 3))
    assert bool((vt_flat[:, :, :, S:] == 0).all()), "V^T padding must be zero"
    # flash attention on the kernel's own (bf16-rounded) q,k,v
    out = torch.empty(B * S, d, dtype=BF16, device=dev)
    lse = torch.empty(B, H, S, dtype=torch.float32, device=dev)
    ops.attn_prefill(q, kc, vt, out, B, H, S, lse=lse)
    qf, kf, vf = q.float(), kc[:, :, :S].float(), vc[:, :, :S].float()
    sc = qf @ kf.transpose(-1, -2) / 16.0
    mask = torch.ones(S, S, dtype=torch.bool, device=dev).tril()
    sc = sc.masked_fill(~mask, float("-inf"))
    ref = (torch.softmax(sc, -1) @ vf).permute(0, 2, 1, 3).reshape(B * S, d)
    assert_close(out, ref, 8e-3, "flash attention")   # P is rounded to bf16 like the reference's cast
    assert_close(lse, torch.logsumexp(sc, -1), 1e-4, "lse")


@pytest.mark.parametrize("variant", [4, 5])
@pytest.mark.parametrize("B,H,S", [(1, 1, 1), (2, 2, 57), (1, 2, 300), (2, 1, 385), (1, 1, 1024)])
def test_attention_forward_kernel_variants(dev, monkeypatch, variant, B, H, S):
    """MAGMA_ATTN_FWD = 4 (16-query waves, two per SIMD) / 5 (32-query waves on the 32x32x16 MFMA, one per SIMD, O^T in AGPRs):
    output and lse against fp32 softmax(QK^T/16)V, the output also at the row stride of a wider buffer, and a late dominant
    key that forces the deferred-maximum rescale (guide rule 26)."""
    from magma_amd import ops
    monkeypatch.setenv("MAGMA_ATTN_FWD", str(variant))
    d = H * 256
    q = rnd(B, H, S, 256, dev=dev, seed=81, scale=0.5).to(BF16)
    k = rnd(B, H, S, 256, dev=dev, seed=82, scale=0.5).to(BF16)
    v = rnd(B, H, S, 256, dev=dev, seed=83).to(BF16)
    if S > 100:                      # one late key dominates a late query block: the running maximum jumps by far more than 2^8
        k[:, :, S - 40] = (q[:, :, S - 5] * 6).to(BF16)
    vt = ops.head_transpose(v, B, H, S, sb=H * S * 256, ss=256, sh=S * 256)
    out = torch.empty(B * S, d, dtype=BF16, device=dev)
    lse = torch.empty(B, H, S, dtype=torch.float32, device=dev)
    ops.attn_prefill(q, k, vt, out, B, H, S, lse=lse)
    sc = q.float() @ k.float().transpose(-1, -2) / 16.0
    sc = sc.masked_fill(~torch.ones(S, S, dtype=torch.bool, device=dev).tril(), float("-inf"))
    ref = (torch.softmax(sc, -1) @ v.float()).permute(0, 2, 1, 3).reshape(B * S, d)
    assert_close(out, ref, 8e-3, "flash attention")
    assert_close(lse, torch.logsumexp(sc, -1), 1e-4, "lse")
    wide = torch.full((B * S, d + 136), float("nan"), dtype=BF16, device=dev)
    ops.attn_prefill(q, k, vt, wide[:, :d], B, H, S, lse=lse)
    assert torch.equal(wide[:, :d], out)


def test_attention_online_softmax_rescale(dev):
    """Force the running-max rescale branch: one late key dominates (guide rule 26)."""
    from magma_amd import ops
    B, H, S, Smax = 1, 1, 128, 128
    q = rnd(B, H, S, 256, dev=dev, seed=71, scale=0.5).to(BF16)
    k = rnd(B, H, Smax, 256, dev=dev, seed=72, scale=0.5).to(BF16)
    v = rnd(B, H, Smax, 256, dev=dev, seed=73).to(BF16)
    k[0, 0, 100] = (q[0, 0, 120].float() * 4).to(BF16)   # spike for late queries at key 100
    vt = v.transpose(2, 3).reshape(B, H, 256, Smax // 32, 32).permute(0, 1, 3, 2, 4).contiguous()   # 32-key tiles
    out = torch.empty(B * S, 256, dtype=BF16, device=dev)
    ops.attn_prefill(q, k, vt, out, B, H, S)
    sc = q.float() @ k.float().transpose(-1, -2) / 16.0
    sc = sc.masked_fill(~torch.ones(S, S, dtype=torch.bool, device=dev).tril(), float("-inf"))
    ref = (torch.softmax(sc, -1) @ v.float()).reshape(S, 256)
    assert_close(out, ref, 8e-3, "rescale branch")


@pytest.mark.parametrize("ctx", [1, 57, 152, 184])
def test_decode_attention(dev, ctx):
    from magma_amd import ops
    from oracle.model import rotary_tables
    B, H, Smax = 3, 2, 192
    d = H * 256
    kc = rnd(B, H, Smax, 256, dev=dev, seed=81, scale=0.5).to(BF16)
    vc = rnd(B, H, Smax, 256, dev=dev, seed=82).to(BF16)
    qkv = rnd(B, 3 * d, dev=dev, seed=83, scale=0.5).to(BF16)
    sin_t, cos_t = rotary_tables(64, Smax)
    sin_t, cos_t = sin_t.to(dev).contiguous(), cos_t.to(dev).contiguous()
    d_pos = torch.tensor([ctx - 1], dtype=torch.int32, device=dev)
    q = torch.empty(B, H, 1, 256, dtype=BF16, device=dev)
    kc0, vc0 = kc.clone(), vc.clone()
    ops.rotary_split(qkv, B, 1, H, 64, sin_t, cos_t, q, kc, vc, d_pos=d_pos)
    x = qkv.view(B, 1, 3, H, 256).float().cpu()
    pos = torch.tensor([ctx - 1])
    k_new = _rotary_ref(x[:, :, 1], pos, 64).permute(0, 2, 1, 3)
    assert_close(kc[:, :, ctx - 1:ctx].cpu(), k_new, 3e-3, "appended k")
    assert torch.equal(vc[:, :, ctx - 1].float().cpu(), x[:, 0, 2])
    if ctx > 1:
        assert torch.equal(kc[:, :, :ctx - 1], kc0[:, :, :ctx - 1]) and torch.equal(vc[:, :, ctx:], vc0[:, :, ctx:])
    out = torch.empty(B, d, dtype=BF16, device=dev)
    ops.attn_decode(q, kc, vc, out, B, H, d_pos)
    sc = (q.float() @ kc[:, :, :ctx].float().transpose(-1, -2)) / 16.0
    ref = (torch.softmax(sc, -1) @ vc[:, :, :ctx].float()).reshape(B, d)
    assert_close(out, ref, 3e-3, "decode attention")
    # fused variant: rotary + append + attention in one launch, from the same qkv row
    kc2, vc2 = kc0.clone(), vc0.clone()
    out2 = torch.empty(B, d, dtype=BF16, device=dev)
    ops.attn_decode_fused(qkv, kc2, vc2, out2, B, H, d_pos, 64, sin_t, cos_t)
    assert torch.equal(kc2, kc) and torch.equal(vc2, vc), "fused append must write the same cache rows"
    assert_close(out2, ref, 3e-3, "fused decode attention")


def test_decode_colaunch_burst_variants_are_bit_identical(dev, monkeypatch):
    """MAGMA_DECODE_PIPE (attention || fc_out with double-buffered weight bursts) changes the load schedule only: outputs equal
    the default launch bit for bit."""
    from magma_amd import ops
    from oracle.model import rotary_tables
    B, H, Smax, ctx = 8, 2, 128, 61
    d = H * 256
    kc0 = rnd(B, H, Smax, 256, dev=dev, seed=81, scale=0.5).to(BF16)
    vc0 = rnd(B, H, Smax, 256, dev=dev, seed=82).to(BF16)
    qkv = rnd(B, 3 * d, dev=dev, seed=83, scale=0.5).to(BF16)
    sin_t, cos_t = rotary_tables(64, Smax)
    sin_t, cos_t = sin_t.to(dev).contiguous(), cos_t.to(dev).contiguous()
    d_pos = torch.tensor([ctx - 1], dtype=torch.int32, device=dev)
    x = rnd(B, 16384, dev=dev, seed=84).to(BF16)
    w = rnd(4096, 16384, dev=dev, seed=85, scale=0.05).to(BF16)
    lin = ops.PackedLinear(w, bias=rnd(4096, dev=dev, seed=86))

    def co(pipe):
        if pipe:
            monkeypatch.setenv("MAGMA_DECODE_PIPE", str(pipe))
        else:
            monkeypatch.delenv("MAGMA_DECODE_PIPE", raising=False)
        kc, vc = kc0.clone(), vc0.clone()
        att = torch.empty(B, d, dtype=BF16, device=dev)
        y = torch.empty(B, 4096, dtype=torch.float32, device=dev)
        ops.decode_attn_gemv(qkv, kc, vc, att, B, H, d_pos, 64, sin_t, cos_t, (x, lin, y, {"out_dtype": torch.float32}))
        return att, y, kc, vc
    ref = co(0)
    assert_close(ref[1], x.float() @ w.float().t() + lin.bias, 1e-3, "fc_out")
    for pipe in (16, 8):
        got = co(pipe)
        assert all(torch.equal(a, b) for a, b in zip(ref, got)), pipe
    monkeypatch.delenv("MAGMA_DECODE_PIPE", raising=False)


def test_argmax_and_pos(dev):
    from magma_amd import ops
    lg = rnd(8, 50258, dev=dev, seed=91)
    lg[3, 17] = lg[3, 40000] = 100.0     # tie -> first index
    tok = ops.argmax(lg)
    ref = torch.argmax(lg.cpu(), dim=-1)
    assert torch.equal(tok.cpu(), ref) and int(tok[3]) == 17
    p = torch.tensor([5], dtype=torch.int32, device=dev)
    ops.advance_pos(p, 2)
    assert int(p) == 7


def test_avgpool_and_stem(dev):
    from magma_amd import ops
    x = rnd(2, 24, 6, 8, dev=dev, seed=101).to(BF16)       # NCHW
    y = ops.avgpool2(x.permute(0, 2, 3, 1).contiguous())
    ref = F.avg_pool2d(x.float(), 2).permute(0, 2, 3, 1)
    assert_close(y, ref, 3e-3, "avgpool")
    img = rnd(2, 3, 20, 16, dev=dev, seed=102).to(BF16)
    w = rnd(8, 3, 3, 3, dev=dev, seed=103, scale=0.2).to(BF16)
    cols = ops.stem_im2col(img)
    wk = torch.zeros(8, 32, dtype=BF16, device=dev)
    wk[:, :27] = w.reshape(8, 27)
    out = ops.gemm(cols, ops.PackedLinear(wk))
    ref = F.conv2d(img.float(), w.float(), stride=2, padding=1).permute(0, 2, 3, 1).reshape(-1, 8)
    assert_close(out, ref, GEMM_TOL, "stem conv via im2col")


def test_build_labels_exact(dev):
    """Integer path: bit-exact against the oracle's literal restatement of reference utils.py:334-364."""
    from magma_amd import ops
    from oracle.model import build_labels
    eos, S, P = 1054, 64, 9
    g = torch.Generator().manual_seed(7)
    cap = torch.randint(0, 1000, (6, S), generator=g)
    cap[0, 20:] = eos            # normal padded caption
    cap[1, 0] = eos              # eos at position 0
    cap[2, :] = eos              # all eos
    # row 3: no eos at all; row 4: eos only inside the truncated tail
    cap[4, S - P + 2] = eos
    cap[5, 5] = eos; cap[5, 30] = eos
    got = ops.build_labels(cap.to(dev), P, eos).cpu()
    ref = build_labels(P, cap, eos)
    assert torch.equal(got, ref)
    with pytest.raises(AssertionError):
        ops.build_labels(cap[:, :4].contiguous().to(dev), P, eos)


def test_cross_entropy(dev):
    from magma_amd import ops
    R, V = 37, 1056
    lg = rnd(R, V, dev=dev, seed=111) * 3
    tg = torch.randint(0, V, (R,), generator=torch.Generator().manual_seed(3))
    tg[::5] = -100
    loss, rows = ops.cross_entropy(lg, tg.to(dev))
    ref = F.cross_entropy(lg.cpu(), tg, ignore_index=-100)
    assert abs(float(loss) - float(ref)) < 1e-4 * max(1.0, abs(float(ref)))


def test_errors_are_loud(dev):
    from magma_amd import ops
    from magma_amd.lib import MagmaHipError
    a = torch.zeros(4, 12, dtype=BF16, device=dev)      # K not a multiple of 8
    with pytest.raises((MagmaHipError, ValueError)):
        ops.gemm(a, ops.PackedLinear(torch.zeros(8, 12, dtype=BF16, device=dev)))
    with pytest.raises(MagmaHipError):
        ops.layernorm(torch.zeros(2, 64, dtype=BF16), torch.ones(64), torch.zeros(64))   # CPU tensor


def test_skinny_layernorm_fold_and_split(dev):
    """Decode fusion: LayerNorm folded into the GEMV + two output segments (qkv | gelu(fc_in))."""
    from magma_amd import ops
    M, K, N1, N2 = 8, 512, 96, 160
    x = (rnd(M, K, dev=dev, seed=201) * 3 + 0.7).to(BF16)
    w = rnd(N1 + N2, K, dev=dev, seed=202, scale=0.05).to(BF16)
    b = rnd(N1 + N2, dev=dev, seed=203)
    g = rnd(K, dev=dev, seed=204) * 0.2 + 1
    be = rnd(K, dev=dev, seed=205) * 0.2
    w2, b2, cs = ops.fold_layernorm(w, b, g, be)
    lin = ops.PackedLinear(w2, bias=b2[:N1].contiguous())
    out_a = torch.empty(M, N1, dtype=BF16, device=dev)
    out_b = torch.empty(M, N2, dtype=BF16, device=dev)
    ops.gemm_skinny(x, lin, out=out_a, ln_fold=(cs, K, 1e-5), split=(N1, out_b, ops.MG_ACT_GELU_NEW, b2[N1:].contiguous()))
    ln = F.layer_norm(x.float(), (K,), g, be, 1e-5)
    ref = ln @ w.float().t() + b
    assert_close(out_a, ref[:, :N1], 1e-2, "ln-fold segment a")
    assert_close(out_b, F.gelu(ref[:, N1:], approximate="tanh"), 1e-2, "ln-fold segment b (gelu)")
    # single segment, fp32 out (lm_head with ln_f folded)
    lin2 = ops.PackedLinear(w2, bias=b2)
    o32 = ops.gemm_skinny(x, lin2, out_dtype=torch.float32, ln_fold=(cs, K, 1e-5))
    assert_close(o32, ref, 1e-2, "ln-fold fp32")


@pytest.mark.parametrize("tile", [256])
@pytest.mark.parametrize("layout", ["rm", "ft"])
@pytest.mark.parametrize("M,N,K", [(256, 256, 128), (1000, 520, 256), (512, 1056, 1024), (2048, 4096, 4096)])
def test_gemm256_deep_pipeline(dev, layout, M, N, K, tile):
    """The 256x256 staggered / counted-vmcnt kernel against fp32 and against the 128x128 kernel;
    repeated launches on fresh data to screen for LDS-DMA races.  (The 32x32x16-MFMA form of the kernel, tile 258, measured
    5-10 % slower and lives in the ablation library only: `make ABL=1`.)"""
    from magma_amd import ops
    for rep in range(3):
        a = rnd(M, K, dev=dev, seed=300 + rep).to(BF16)
        w = rnd(N, K, dev=dev, seed=310 + rep, scale=0.05).to(BF16)
        bias = rnd(N, dev=dev, seed=320 + rep)
        res = rnd(M, N, dev=dev, seed=330 + rep).to(BF16)
        lin = ops.PackedLinear(w, bias=bias, tiled=True, rowmajor=True)
        out = ops.gemm(a, lin, layout=layout, residuals=(res,), tile=tile)
        ref = a.float() @ w.float().t() + bias + res.float()
        assert_close(out, ref, GEMM_TOL, f"gemm256 {layout} {M}x{N}x{K} rep{rep}")
        out128 = ops.gemm(a, lin, layout=layout, residuals=(res,), tile=128)
        assert float((out.float() - out128.float()).abs().max()) <= 2e-2 * float(ref.abs().max())


@pytest.mark.parametrize("group_m", [1, 2, 3, 5, 8, 32])
def test_gemm256_tile_walk_group_sizes(dev, monkeypatch, group_m):
    """MAGMA_G256_GROUP_M: the tile walk of the 256x256 kernel (group_m row-tiles x 32 / group_m column-tiles per XCD at a time) is
    a bijection onto the tiles for every group size, ragged last group included: same bits as the default walk, plain and with a
    residual + GELU epilogue (the two row-walk builds)."""
    from magma_amd import ops
    M, N, K = 1800, 1300, 256                     # 8 x 6 tiles, both edges ragged
    a = rnd(M, K, dev=dev, seed=401).to(BF16)
    w = rnd(N, K, dev=dev, seed=402, scale=0.05).to(BF16)
    res = rnd(M, 1304, dev=dev, seed=403).to(BF16)[:, :N]
    lin = ops.PackedLinear(w, bias=rnd(N, dev=dev, seed=404))
    monkeypatch.delenv("MAGMA_G256_GROUP_M", raising=False)
    ref_plain = ops.gemm(a, lin, tile=256, out_dtype=torch.float32)
    ref_res = ops.gemm(a, lin, tile=256, act=ops.MG_ACT_GELU_NEW, residuals=(res,))
    assert_close(ref_plain, a.float() @ w.float().t() + lin.bias, GEMM_TOL, "default walk")
    monkeypatch.setenv("MAGMA_G256_GROUP_M", str(group_m))
    assert torch.equal(ops.gemm(a, lin, tile=256, out_dtype=torch.float32), ref_plain)
    assert torch.equal(ops.gemm(a, lin, tile=256, act=ops.MG_ACT_GELU_NEW, residuals=(res,)), ref_res)


@pytest.mark.parametrize("tile,M", [(0, 77), (0, 456), (256, 456), (0, 1300)])
def test_gemm_activation_from_column(dev, tile, M):
    """ep.act_n0: the activation applies to output columns >= act_n0 only -- [q | k | v | fc_in] of a GPT-J block as ONE
    GEMM with gelu_new on the fc_in columns (engine._blocks_prefill); 128x128 (+ split-K fix-up), 256x256 kernels."""
    from magma_amd import ops
    N, K, n0 = 1536 + 2048, 512, 1536
    a = rnd(M, K, dev=dev, seed=41).to(BF16)
    w = rnd(N, K, dev=dev, seed=42, scale=0.05).to(BF16)
    bias = rnd(N, dev=dev, seed=43)
    lin = ops.PackedLinear(w, bias=bias)
    out = ops.gemm(a, lin, act=ops.MG_ACT_GELU_NEW, act_n0=n0, tile=tile)
    ref = a.float() @ w.float().t() + bias
    ref[:, n0:] = F.gelu(ref[:, n0:], approximate="tanh")
    assert_close(out, ref, GEMM_TOL, f"act_n0 tile={tile} M={M}")
    # the two column ranges as separate GEMMs on row views of the same packed weight: same values
    qkv = ops.gemm(a, lin.rows(0, n0, bias=bias[:n0].contiguous()))
    h = ops.gemm(a, lin.rows(n0, N, bias=bias[n0:].contiguous()), act=ops.MG_ACT_GELU_NEW)
    assert float((out[:, :n0].float() - qkv.float()).abs().max()) <= 2e-2 * float(ref.abs().max())
    assert float((out[:, n0:].float() - h.float()).abs().max()) <= 2e-2 * float(ref.abs().max())


@pytest.mark.parametrize("layout", ["rm", "ft"])
def test_gemm_split_k(dev, layout):
    """Split-K (fp32 slabs + fixed-order fixup) equals the single-pass kernel up to fp32 summation
    order, is run-to-run deterministic, and carries the whole epilogue."""
    from magma_amd import ops
    M, N, K = 456, 203, 1000      # ragged M/N, K-tiles = 16 (last one partial)
    a = rnd(M, K, dev=dev, seed=21).to(BF16)
    w = rnd(N, K, dev=dev, seed=22, scale=0.05).to(BF16)
    bias = rnd(N, dev=dev, seed=23)
    r0 = rnd(M, 208, dev=dev, seed=24).to(BF16)
    lin = ops.PackedLinear(w, bias=bias, tiled=True, rowmajor=True)
    ref = F.gelu(a.float() @ w.float().t() + bias, approximate="tanh") + r0[:, :N].float()
    base = ops.gemm(a, lin, layout=layout, act=ops.MG_ACT_GELU_NEW, residuals=(r0,), out_dtype=torch.float32, split_k=1)
    assert_close(base, ref, 1e-4, "split_k=1")
    for sk in (0, 2, 3, 7, 16):
        buf = torch.full((M, 208), 7.0, dtype=torch.float32, device=dev)
        ops.gemm(a, lin, out=buf, layout=layout, act=ops.MG_ACT_GELU_NEW, residuals=(r0,), split_k=sk)
        assert_close(buf[:, :N], ref, 1e-4, f"split_k={sk}")
        assert bool((buf[:, N:] == 7.0).all()), "wrote past N"
        again = ops.gemm(a, lin, layout=layout, act=ops.MG_ACT_GELU_NEW, residuals=(r0,), out_dtype=torch.float32, split_k=sk)
        assert torch.equal(again, buf[:, :N]), f"split_k={sk} is not deterministic"
    # implicit-im2col A operand: the split has to start mid-way through the taps
    B, H, W, Cin, Cout = 2, 7, 9, 48, 96
    x = rnd(B, Cin, H, W, dev=dev, seed=25).to(BF16)
    wc = rnd(Cout, Cin, 3, 3, dev=dev, seed=26, scale=0.1).to(BF16)
    linc = ops.PackedLinear(wc.permute(0, 2, 3, 1).reshape(Cout, 9 * Cin).contiguous(), tiled=True, rowmajor=True)
    xn = x.permute(0, 2, 3, 1).contiguous().view(B * H * W, Cin)
    refc = F.conv2d(x.float(), wc.float(), padding=1).permute(0, 2, 3, 1).reshape(B * H * W, Cout)
    for sk in (1, 2, 5):
        assert_close(ops.gemm(xn, linc, conv=(H, W, Cin), layout=layout, split_k=sk), refc, GEMM_TOL, f"conv split_k={sk}")
    with pytest.raises(Exception, match="split_k"):
        ops.gemm(a, lin, split_k=65)


@pytest.mark.parametrize("M,N,K,split", [(4096, 1024, 32768, 0), (1024, 4096, 32768, 0), (1024, 1024, 4096, 2), (600, 520, 2048, 4), (4096, 1024, 32768, 4),
                                           (456, 4096, 16384, 0)])     # the prefill's fc_out: 2 x 16 tiles x 8 splits by the automatic policy
def test_gemm256_split_k(dev, M, N, K, split):
    """The 256x256 kernel with the contraction cut across workgroups (round 4: the adapters' weight gradients, few output tiles over
    K = B*S = 32768): fp32 slabs + the deterministic fix-up, against fp32 and against the 128x128 kernel; split = 0 lets the library
    choose (it picks the split 256x256 form for the two wgrad shapes), otherwise tile 256 + split_k are forced; ragged M / N included;
    two launches give identical bits."""
    from magma_amd import ops
    a = rnd(M, K, dev=dev, seed=500).to(BF16)
    w = rnd(N, K, dev=dev, seed=501, scale=0.05).to(BF16)
    bias = rnd(N, dev=dev, seed=502)
    res = rnd(M, ops.ceil_to(N, 8), dev=dev, seed=503).to(BF16)[:, :N]
    lin = ops.PackedLinear(w, bias=bias, tiled=True, rowmajor=True)
    ref = a.float() @ w.float().t() + bias + res.float()
    for layout in ("rm", "ft"):
        kw = dict(layout=layout, residuals=(res,), out_dtype=torch.float32)
        out = ops.gemm(a, lin, tile=256 if split else 0, split_k=split, **kw)
        assert_close(out, ref, GEMM_TOL, f"gemm256 split-K {layout} {M}x{N}x{K} split {split}")
        out2 = ops.gemm(a, lin, tile=256 if split else 0, split_k=split, **kw)
        assert torch.equal(out, out2)
        o128 = ops.gemm(a, lin, tile=128, **kw)
        assert float((out - o128).abs().max()) <= 2e-3 * float(ref.abs().max())


@pytest.mark.parametrize("M,N,K,tile,split", [
    (96, 264, 1568, 0, 1),          # 128x128 kernel, un-split, ragged M / N (a late CLIP stage's weight gradient)
    (96, 264, 1568, 0, 0),          # ... cut along K by the automatic policy: the fix-up accumulates
    (300, 204, 2048, 128, 3),       # N % 8 != 0: the 4-column epilogue walk
    (512, 1056, 4096, 256, 1),      # 256x256 kernel, un-split
    (4096, 1024, 32768, 0, 0),      # the adapters' weight gradient: split 256x256 form + fix-up
])
def test_gemm_accumulates_into_fp32_output_with_row_scale(dev, M, N, K, tile, split):
    """mg_epilogue.accumulate / .row_scale (ABI 6): C (fp32) += row_scale[m] * (A W^T) in place -- what the weight-gradient GEMMs of
    the training engine do to the gradient buffer (reference: .grad accumulation over micro-batches, train_loop.py:22-33).  Every
    kernel form: both tile kernels, un-split and split-K (fix-up); columns past N untouched; a bf16 output refuses."""
    from magma_amd import ops
    a = rnd(M, K, dev=dev, seed=700).to(BF16)
    w = rnd(N, K, dev=dev, seed=701, scale=0.05).to(BF16)
    rs = rnd(M, dev=dev, seed=702).abs() + 0.5
    lin = ops.PackedLinear(w, tiled=True, rowmajor=True)
    prod = a.float() @ w.float().t()
    ld = ops.ceil_to(N, 4) + 4
    for layout in ("rm", "ft"):
        for use_rs in (False, True):
            buf = rnd(M, ld, dev=dev, seed=703).contiguous()
            before = buf.clone()
            ops.gemm(a, lin, out=buf[:, :N], layout=layout, use_bias=False, tile=tile, split_k=split, accumulate=True,
                     row_scale=rs if use_rs else None)
            ref = before[:, :N] + (prod * rs[:, None] if use_rs else prod)
            assert_close(buf[:, :N], ref, 2e-5, f"accumulate {layout} row_scale={use_rs}")
            assert torch.equal(buf[:, N:], before[:, N:]), "wrote past N"
            # twice = the product added twice (no hidden state in the workspace)
            ops.gemm(a, lin, out=buf[:, :N], layout=layout, use_bias=False, tile=tile, split_k=split, accumulate=True,
                     row_scale=rs if use_rs else None)
            assert_close(buf[:, :N], before[:, :N] + 2 * (prod * rs[:, None] if use_rs else prod), 2e-5, "second accumulation")
    with pytest.raises(Exception, match="accumulate"):
        d_out = torch.zeros(M, ops.ceil_to(N, 8), dtype=BF16, device=dev)
        ops.gemm(a, lin, out=d_out[:, :N], use_bias=False, accumulate=True)


@pytest.mark.parametrize("M,N,ld", [(300, 520, 528), (2048, 1024, 1024)])
def test_gelu_erf_passes(dev, M, N, ld):
    """torch.nn.GELU() (erf) and g * gelu'(pre) as stand-alone passes (adapters built with activation=nn.GELU, reference
    adapters.py:11,20), strided rows and in place, against torch."""
    from magma_amd import ops
    x = (rnd(M, ld, dev=dev, seed=600) * 2).to(BF16)[:, :N]
    g = rnd(M, ld, dev=dev, seed=601).to(BF16)[:, :N]
    assert_close(ops.gelu_erf(x), F.gelu(x.float()), 3e-3, "gelu (erf)")
    xr = x.float().clone().requires_grad_(True)
    F.gelu(xr).backward(torch.ones_like(xr))
    assert_close(ops.gelu_erf_grad_mul(g, x), g.float() * xr.grad, 3e-3, "gelu (erf) gradient")
    y = x.clone()
    ops.gelu_erf(y, out=y)
    assert torch.equal(y, ops.gelu_erf(x))
