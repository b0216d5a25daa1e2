import os, sys, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from magma_amd import ops
dev = torch.device("cuda:0"); BF = torch.bfloat16
B, H, S = 1, 1, int(os.environ.get("AS", 64)); d = H * 256
g = torch.Generator(device="cpu").manual_seed(20)
q = (torch.randn(B, H, S, 256, generator=g) * 0.5).to(dev).to(BF)
k = (torch.randn(B, H, S, 256, generator=g) * 0.5).to(dev).to(BF)
v = torch.randn(B, H, S, 256, generator=g).to(dev).to(BF)
dO = torch.randn(B * S, d, generator=g).to(dev).to(BF)
vt = ops.head_transpose(v, B, H, S, sb=H * S * 256, ss=256, sh=S * 256)
out = torch.empty(B * S, d, dtype=BF, device=dev); lse = torch.empty(B, H, S, dtype=torch.float32, device=dev)
ops.attn_prefill(q, k, vt, out, B, H, S, lse=lse)
qt = ops.head_transpose(q, B, H, S, sb=H * S * 256, ss=256, sh=S * 256)
kt = ops.head_transpose(k, B, H, S, sb=H * S * 256, ss=256, sh=S * 256)
dOt = ops.head_transpose(dO, B, H, S, sb=S * d, ss=d, sh=256)
os.environ["MAGMA_ATTN_BWD"] = "0"; r0 = ops.attn_bwd(q, k, v, qt, kt, dO, dOt, out, lse, B, H, S)
os.environ["MAGMA_ATTN_BWD"] = "2"; r2 = ops.attn_bwd(q, k, v, qt, kt, dO, dOt, out, lse, B, H, S)
a, b = r0[1][0, 0].float(), r2[1][0, 0].float()
err_key = ((a - b).norm(dim=1) / a.norm(dim=1).clamp_min(1e-20))
print("per-key rel err:", [round(float(x), 3) for x in err_key])
err_d = ((a - b).norm(dim=0) / a.norm(dim=0).clamp_min(1e-20))
print("per-d rel err (first 64):", [round(float(x), 3) for x in err_d[:64]])
print("per-d rel err (all, x1000):", [int(float(x)*1000) for x in err_d])
for nm, ix in (("dq", 0), ("dv", 2)):
    print(nm, "max abs diff", float((r0[ix].float() - r2[ix].float()).abs().max()))
kk = int(err_key.argmax()); print("worst key", kk, "old", a[kk, :8].tolist(), "new", b[kk, :8].tolist())
