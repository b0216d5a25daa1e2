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

inv = 1.0 / (10000 ** (torch.arange(0, 64, 2, dtype=torch.float32, device=dev) / 64))
ang = torch.arange(S, dtype=torch.float32, device=dev)[:, None] * inv[None, :]
sin_t, cos_t = ang.sin().contiguous(), ang.cos().contiguous()
fwds = {}
for var in os.environ.get("AFWD", "4,5").split(","):       # MAGMA_ATTN_FWD variants (attention.hip: 4 = 16-query waves, 5 = 32-query waves)
    os.environ["MAGMA_ATTN_FWD"] = var
    fwds[var] = t(lambda: ops.attn_prefill(q, k, vt, out, B, H, S, lse=lse))
fwd = fwds[min(fwds)]
# round 6: the kernels without transposed images (attention_tr.hip), operands as [B,H,S,256] tensors and as column ranges of one fused qkv
# activation [B*S, 3 H 256] (what the training engine hands over)
rows = ops.AttnRows.of_bhsd(q, k, v)
fused = torch.stack((q, k, v)).permute(1, 3, 0, 2, 4).reshape(B * S, 3 * d).contiguous()
rows_f = ops.AttnRows.of_qkv(fused, B, S, H)
fwd_rows = t(lambda: ops.attn_fwd_rows(rows, out, lse=lse))
fwd_rows_f = t(lambda: ops.attn_fwd_rows(rows_f, out, lse=lse))
qt = ops.head_transpose(q, B, H, S, sb=hs, ss=256, sh=S * 256)
kt = ops.head_transpose(k, B, H, S, sb=hs, ss=256, sh=S * 256)
dOt = ops.head_transpose(dO, B, H, S, sb=S * d, ss=d, sh=256)
fl = B * H * 4 * S * S * 256 / 2
res = {"B": B, "S": S, "fwd_ms": round(fwd, 4), "fwd_tflops_causal": round(fl / fwd / 1e9, 1),
       "fwd_rows_ms": round(fwd_rows, 4), "fwd_rows_fused_qkv_ms": round(fwd_rows_f, 4)}
for var, ms in fwds.items():
    res["fwd_ms_v" + var] = round(ms, 4)
# MAGMA_ATTN_BWD variants (attention_bwd.hip: 0 = three 16-row-wave kernels, 1/2 = merged dK+dV on 32-key waves, 3/4 = + 32-query dQ)
for var in os.environ.get("ABWD", "0,1,2,3,4").split(","):
    os.environ["MAGMA_ATTN_BWD"] = var
    bwd = t(lambda: ops.attn_bwd(q, k, v, qt, kt, dO, dOt, out, lse, B, H, S))
    res["bwd_ms_v" + var] = round(bwd, 4)
    res["bwd_tflops_causal_v" + var + "(2.5x fwd flops)"] = round(2.5 * fl / bwd / 1e9, 1)
res["bwd_rows_ms"] = round(t(lambda: ops.attn_bwd_rows(rows, dO, out, lse)), 4)
res["bwd_rows_fused_qkv_merged_ms"] = round(t(lambda: ops.attn_bwd_rows(rows_f, dO, out, lse, merged_rot=(64, sin_t, cos_t))), 4)
res["rotary_qk_inplace_ms"] = round(t(lambda: ops.rotary_qk_inplace(fused, B, S, H, 64, sin_t, cos_t)), 4)
print(json.dumps(res))
