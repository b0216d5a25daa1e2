"""Split-K sweep of the three small-M GEMMs of a prefill layer after [q|k|v|fc_in] (M = 456 rows): fc_out (N 4096, K 16384), adapter-down
(N 1024, K 4096, ReLU), [W_out | W_up] (N 4096, K 5120, two residuals) -- time of GEMM + fix-up per forced split and for the automatic policy."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from magma_amd import ops
dev = torch.device("cuda:0")
BF = torch.bfloat16
M = int(os.environ.get("PM", 456))
def t(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
r1, r2 = torch.randn(M, 4096, device=dev).to(BF), torch.randn(M, 4096, device=dev).to(BF)
cases = {"fc_out": (16384, 4096, {}), "adapter_down": (4096, 1024, {"act": ops.MG_ACT_RELU}), "out_up": (5120, 4096, {"residuals": (r1, r2)})}
for name, (K, N, kw) in cases.items():
    a = torch.randn(M, K, device=dev).to(BF)
    w = ops.PackedLinear((torch.randn(N, K, device=dev) * 0.05).to(BF), bias=torch.randn(N, device=dev))
    out = torch.empty(M, N, dtype=BF, device=dev)
    row = {"case": name, "M": M, "N": N, "K": K}
    for tile, split in ((0, 0), (128, 1), (128, 2), (128, 4), (128, 8), (128, 16), (256, 1), (256, 2), (256, 4), (256, 8)):
        try:
            row[f"tile{tile}_split{split}_us"] = round(t(lambda: ops.gemm(a, w, out=out, tile=tile, split_k=split, **kw)), 2)
        except Exception as e:  # noqa: BLE001
            row[f"tile{tile}_split{split}_us"] = "n/a"
    print(json.dumps(row))
