"""fp8 attention forward (mg_rotary_split_fp8 + mg_attn_prefill_fp8) against the bf16 pair at the training shape: times, errors."""
import os, sys, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from magma_amd import ops
dev = torch.device("cuda:0"); BF = torch.bfloat16
B, H, S = int(os.environ.get("AB", 16)), 16, int(os.environ.get("AS", 2048)); d = H * 256; rot = 64
qkv = (torch.randn(B * S, 3 * d, device=dev) * 0.7).to(BF)
inv = 1.0 / (10000 ** (torch.arange(0, rot, 2, dtype=torch.float32, device=dev) / rot))
ang = torch.arange(S, dtype=torch.float32, device=dev)[:, None] * inv[None, :]
sin_t, cos_t = ang.sin().contiguous(), ang.cos().contiguous()
ld = ops.ceil_to(S, 32)
mk = lambda: torch.empty(B, H, S, 256, dtype=BF, device=dev)
mt = lambda: torch.empty(B, H, ld // 32, 256, 32, dtype=BF, device=dev)
q, k, v, vt, qt, kt = mk(), mk(), mk(), mt(), mt(), mt()
out16 = torch.empty(B * S, d, dtype=BF, device=dev); out8 = torch.empty_like(out16)
lse = torch.empty(B, H, S, dtype=torch.float32, device=dev)
def t(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
res = {"B": B, "S": S}
res["rotary_split_train_ms"] = round(t(lambda: ops.rotary_split_train(qkv, B, S, H, rot, sin_t, cos_t, q, k, v, vt, qt, kt)), 4)
op = ops.rotary_split_fp8(qkv, B, S, H, rot, sin_t, cos_t, q, k, v, qt, kt)
res["rotary_split_fp8_ms"] = round(t(lambda: ops.rotary_split_fp8(qkv, B, S, H, rot, sin_t, cos_t, q, k, v, qt, kt)), 4)
res["attn_fwd_bf16_ms"] = round(t(lambda: ops.attn_prefill(q, k, vt, out16, B, H, S, lse=lse)), 4)
res["attn_fwd_fp8_ms"] = round(t(lambda: ops.attn_prefill_fp8(op, out8, lse=lse)), 4)
fl = B * H * 4 * S * S * 256 / 2
res["fp8_tflops_causal"] = round(fl / res["attn_fwd_fp8_ms"] / 1e9, 1)
res["rel_fp8_vs_bf16"] = round(float((out8.float() - out16.float()).norm() / out16.float().norm()), 4)
print(json.dumps(res))
