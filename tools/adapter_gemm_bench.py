"""The four adapter GEMMs of a MAGMA_v1 training step in isolation (M = 16 x 2048 rows, row-major live weights): 128x128 vs 256x256 kernel.
up forward (N 4096, K 1024, bias + 3 residuals), down dgrad (N 4096, K 1024, 1 residual), down forward (N 1024, K 4096, ReLU),
up dgrad (N 1024, K 4096, ReLU gate aux)."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from magma_amd import ops
dev = torch.device("cuda:0")
BF = torch.bfloat16
M = 32768
def t(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
r1, r2, r3 = (torch.randn(M, 4096, device=dev).to(BF) for _ in range(3))
cases = {
    "up_fwd": (1024, 4096, dict(residuals=(r1, r2, r3))),
    "dn_dgrad": (1024, 4096, dict(residuals=(r1,), use_bias=False)),
    "dn_fwd": (4096, 1024, dict(act=ops.MG_ACT_RELU)),
    "up_dgrad": (4096, 1024, dict(aux=torch.randn(M, 1024, device=dev).to(BF), aux_mode=ops.MG_AUX_RELU_GATE, use_bias=False)),
}
for name, (K, N, kw) in cases.items():
    a = torch.randn(M, K, device=dev).to(BF)
    w = ops.RawWeight((torch.randn(N, K, device=dev) * 0.05).to(BF), bias=torch.randn(N, device=dev))
    out = torch.empty(M, N, dtype=BF, device=dev)
    r = {"case": name, "M": M, "N": N, "K": K}
    for tile in (128, 256):
        ms = t(lambda: ops.gemm(a, w, out=out, layout="rm", tile=tile, **kw))
        r[f"tile{tile}_us"] = ms * 1e3
        r[f"tile{tile}_tflops"] = 2.0 * M * N * K / ms / 1e9
    print(json.dumps(r))
