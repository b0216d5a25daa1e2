"""In-kernel segment times of attn_bwd_dkdv32_kernel (ablation library, MAGMA_ATTN_BWD32_ABL=6): s_memtime totals per wave."""
import os, sys, ctypes as C
os.environ["MAGMA_ATTN_BWD32_ABL"] = "6"; os.environ["MAGMA_ATTN_BWD"] = "2"
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from magma_amd import ops, lib as L
dev = torch.device("cuda:0"); BF = torch.bfloat16
B, H, S = 16, 16, 2048; d = H * 256
q = (torch.randn(B, H, S, 256, device=dev) * 0.5).to(BF); k = (torch.randn(B, H, S, 256, device=dev) * 0.5).to(BF)
v = torch.randn(B, H, S, 256, device=dev).to(BF); dO = torch.randn(B * S, d, device=dev).to(BF)
hs = H * S * 256
vt = ops.head_transpose(v, B, H, S, sb=hs, ss=256, sh=S * 256)
out = torch.empty(B * S, d, dtype=BF, device=dev); lse = torch.empty(B, H, S, dtype=torch.float32, device=dev)
ops.attn_prefill(q, k, vt, out, B, H, S, lse=lse)
qt = ops.head_transpose(q, B, H, S, sb=hs, ss=256, sh=S * 256); kt = ops.head_transpose(k, B, H, S, sb=hs, ss=256, sh=S * 256)
dOt = ops.head_transpose(dO, B, H, S, sb=S * d, ss=d, sh=256)
for _ in range(2):
    ops.attn_bwd(q, k, v, qt, kt, dO, dOt, out, lse, B, H, S)
torch.cuda.synchronize()
n = 4096 * 4 * 8
buf = (C.c_uint64 * n)()
L.check(L.load().mg_debug_attn_stamps(buf, C.c_int64(n * 8)), "stamps")
t = torch.tensor(list(buf), dtype=torch.float64).view(4096, 4, 8)
steps = t[:, :, 5]
names = ["wait+barrier", "dma0+phase1 (S)", "dma1+phase2 (dP, exp)", "dma2+phase3 (dV, dS)", "dma3+phase4 (dK)"]
act = steps > 0
print("s_memtime ticks (100 MHz units?) per ACTIVE tile step, mean over waves with steps > 0; by wave index")
tot = 0
for i, nm in enumerate(names):
    per = (t[:, :, i] / steps.clamp_min(1))
    m = float(per[act].mean()); tot += m
    print(f"  {nm:28s} {m:9.1f}   by wave: " + "  ".join(f"{float(per[:, w][act[:, w]].mean()):8.1f}" for w in range(4)))
print(f"  total {tot:9.1f}   steps/wave mean {float(steps[act].mean()):.1f}")
