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

fwds = {}
for var in os.environ.get("AFWD", "4,5").split(","):       # MAGMA_ATTN_FWD variants (attention.hip: 4 = 16-query waves, 5 = 32-query waves)
    os.environ["MAGMA_ATTN_FWD"] = var
    fwds[var] = t(lambda: ops.attn_prefill(q, k, vt, out, B, H, S, lse=lse))
fwd = fwds[min(fwds)]
qt = ops.head_transpose(q, B, H, S, sb=hs, ss=256, sh=S * 256)
kt = ops.head_transpose(k, B, H, S, sb=hs, ss=256, sh=S * 256)
dOt = ops.head_transpose(dO, B, H, S, sb=S * d, ss=d, sh=256)
fl = B * H * 4 * S * S * 256 / 2
res = {"B": B, "S": S, "fwd_ms": round(fwd, 4), "fwd_tflops_causal": round(fl / fwd / 1e9, 1)}
for var, ms in fwds.items():
    res["fwd_ms_v" + var] = round(ms, 4)
# MAGMA_ATTN_BWD variants (attention_bwd.hip: 0 = three 16-row-wave kernels, 1/2 = merged dK+dV on 32-key waves, 3/4 = + 32-query dQ)
for var in os.environ.get("ABWD", "0,1,2,3,4").split(","):
    os.environ["MAGMA_ATTN_BWD"] = var
    bwd = t(lambda: ops.attn_bwd(q, k, v, qt, kt, dO, dOt, out, lse, B, H, S))
    res["bwd_ms_v" + var] = round(bwd, 4)
    res["bwd_tflops_causal_v" + var + "(2.5x fwd flops)"] = round(2.5 * fl / bwd / 1e9, 1)
print(json.dumps(res))
