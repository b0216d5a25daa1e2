"""fp8 (OCP e4m3) operand path of BASELINE config 5: quantiser and MFMA-scale GEMM against a torch restatement
(torch.float8_e4m3fn casts + fp32 matmul on the same device)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
BF16 = torch.bfloat16


def rel(a, b):
    a, b = a.float(), b.float()
    return float((a - b).norm() / (b.norm() + 1e-12))


def ref_quant(x):
    amax = x.float().abs().amax(1)
    scale = torch.where(amax > 0, amax * (1.0 / 448.0), torch.ones_like(amax))
    q = (x.float() * (1.0 / scale)[:, None]).clamp(-448, 448).to(torch.float8_e4m3fn)
    return q, scale


@pytest.mark.parametrize("M,K", [(5, 64), (37, 1000), (8, 4096), (3, 16384)])
def test_quantize_rows(dev, M, K):
    from magma_amd import ops
    g = torch.Generator(device=dev).manual_seed(M * K)
    x = (torch.randn(M, K, device=dev, generator=g) * torch.logspace(-3, 2, M, device=dev)[:, None]).to(BF16)
    x[0, 1] = 0
    q, sc = ops.quantize_rows_fp8(x)
    rq, rs = ref_quant(x)
    assert torch.equal(sc, rs)
    assert q.shape[1] % 128 == 0 and bool((q[:, K:] == 0).all())
    got = q[:, :K].view(torch.float8_e4m3fn).float()
    # same rounding (nearest even, saturating) as the torch cast: identical bytes up to the sign of zero
    assert torch.equal(got, rq.float())
    assert rel(got * sc[:, None], x) < 0.04          # e4m3: 3 mantissa bits


@pytest.mark.parametrize("layout", ["rm", "ft"])
@pytest.mark.parametrize("M,N,K", [(128, 128, 128), (300, 200, 192), (77, 1056, 1008), (456, 4096, 4096), (2048, 1024, 4096)])
def test_gemm_fp8(dev, layout, M, N, K):
    from magma_amd import ops
    g = torch.Generator(device=dev).manual_seed(M + N + K)
    a = torch.randn(M, K, device=dev, generator=g).to(BF16)
    w = (torch.randn(N, K, device=dev, generator=g) * 0.05).to(BF16)
    bias = torch.randn(N, device=dev, generator=g)
    res = torch.randn(M, ops.ceil_to(N, 8), device=dev, generator=g).to(BF16)
    lin = ops.PackedLinearFP8(w, bias=bias, tiled=True, rowmajor=True)
    aq, asc = ops.quantize_rows_fp8(a)
    # exact restatement of the arithmetic: products of the quantised values, fp32 accumulation, scales, epilogue
    aqf = aq[:, :K].view(torch.float8_e4m3fn).float()
    ref = F.gelu((aqf @ lin.dequant().t() / 1.0) * asc[:, None] + bias, approximate="tanh") + res[:, :N].float()
    out = ops.gemm_fp8(aq, asc, lin, layout=layout, act=ops.MG_ACT_GELU_NEW, residuals=(res,), out_dtype=torch.float32)
    assert rel(out, ref) < 1e-4, rel(out, ref)
    for sk in (1, 3):
        o2 = ops.gemm_fp8(aq, asc, lin, layout=layout, act=ops.MG_ACT_GELU_NEW, residuals=(res,), out_dtype=torch.float32, split_k=sk)
        assert rel(o2, ref) < 1e-4
    # and the quantisation error against the unquantised product stays at the e4m3 level
    full = a.float() @ w.float().t()
    plain = ops.gemm_fp8(aq, asc, lin, layout=layout, use_bias=False, out_dtype=torch.float32)
    assert rel(plain, full) < 0.06, rel(plain, full)
