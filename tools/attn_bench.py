"""Attention kernels in isolation at the training shape (S=2048): forward, backward."""
import os, sys, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from magma_amd import ops
dev = torch.device("cuda:0")
B, H, S = int(os.environ.get("AB", 4)), 16, int(os.environ.get("AS", 2048))
d = H * 256
BF = torch.bfloat16
q = (torch.randn(B, H, S, 256, device=dev) * 0.5).to(BF)
k = (torch.randn(B, H, S, 256, device=dev) * 0.5).to(BF)
v = torch.randn(B, H, S, 256, device=dev).to(BF)
dO = torch.randn(B * S, d, device=dev).to(BF)
hs = H * S * 256
vt = ops.head_transpose(v, B, H, S, sb=hs, ss=256, sh=S * 256)
out = torch.empty(B * S, d, dtype=BF, device=dev)
lse = torch.empty(B, H, S, dtype=torch.float32, device=dev)

def t(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

fwd = t(lambda: ops.attn_prefill(q, k, vt, out, B, H, S, lse=lse))
qt = ops.head_transpose(q, B, H, S, sb=hs, ss=256, sh=S * 256)
kt = ops.head_transpose(k, B, H, S, sb=hs, ss=256, sh=S * 256)
dOt = ops.head_transpose(dO, B, H, S, sb=S * d, ss=d, sh=256)
bwd = t(lambda: ops.attn_bwd(q, k, v, qt, kt, dO, dOt, out, lse, B, H, S))
fl = B * H * 4 * S * S * 256 / 2
print(json.dumps({"B": B, "S": S, "fwd_ms": fwd, "fwd_tflops_causal": fl / fwd / 1e9, "bwd_ms": bwd,
                  "bwd_tflops_causal(2.5x fwd flops)": 2.5 * fl / bwd / 1e9}))
