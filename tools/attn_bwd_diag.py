"""Attention backward variants (MAGMA_ATTN_BWD) against fp32 autograd: relative errors per gradient and shape."""
import os, sys, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from magma_amd import ops
dev = torch.device("cuda:0")
BF = torch.bfloat16
def rel(a, b): return float((a.float() - b.float()).norm() / b.float().norm().clamp_min(1e-30))
for (B, H, S) in [(1, 1, 64), (2, 2, 57), (1, 2, 300), (1, 1, 1024)]:
    d = H * 256
    g = torch.Generator(device="cpu").manual_seed(20)
    q = (torch.randn(B, H, S, 256, generator=g) * 0.5).to(dev).to(BF)
    k = (torch.randn(B, H, S, 256, generator=g) * 0.5).to(dev).to(BF)
    v = torch.randn(B, H, S, 256, generator=g).to(dev).to(BF)
    dO = torch.randn(B * S, d, generator=g).to(dev).to(BF)
    vt = ops.head_transpose(v, B, H, S, sb=H * S * 256, ss=256, sh=S * 256)
    out = torch.empty(B * S, d, dtype=BF, device=dev)
    lse = torch.empty(B, H, S, dtype=torch.float32, device=dev)
    ops.attn_prefill(q, k, vt, out, B, H, S, lse=lse)
    qt = ops.head_transpose(q, B, H, S, sb=H * S * 256, ss=256, sh=S * 256)
    kt = ops.head_transpose(k, B, H, S, sb=H * S * 256, ss=256, sh=S * 256)
    dOt = ops.head_transpose(dO, B, H, S, sb=S * d, ss=d, sh=256)
    qf, kf, vf = (t.float().requires_grad_(True) for t in (q, k, v))
    sc = qf @ kf.transpose(-1, -2) / 16.0
    sc = sc.masked_fill(~torch.ones(S, S, dtype=torch.bool, device=dev).tril(), float("-inf"))
    o = (torch.softmax(sc, -1) @ vf).permute(0, 2, 1, 3).reshape(B * S, d)
    o.backward(dO.float())
    res = {}
    for var in os.environ.get("ABWD", "0,2,4").split(","):
        os.environ["MAGMA_ATTN_BWD"] = var
        dq, dk, dv = ops.attn_bwd(q, k, v, qt, kt, dO, dOt, out, lse, B, H, S)
        res[var] = (dq, dk, dv)
        print(json.dumps({"shape": [B, H, S], "variant": var, "dq": round(rel(dq, qf.grad), 5), "dk": round(rel(dk, kf.grad), 5),
                          "dv": round(rel(dv, vf.grad), 5)}))
    vs = list(res)
    for a in vs[1:]:
        print(json.dumps({"shape": [B, H, S], "pair": [vs[0], a], "dq": round(rel(res[a][0], res[vs[0]][0]), 5),
                          "dk": round(rel(res[a][1], res[vs[0]][1]), 5), "dv": round(rel(res[a][2], res[vs[0]][2]), 5)}))
