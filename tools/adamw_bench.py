"""AdamW / sum-of-squares kernels alone at the size of MAGMA_v1's trainable set (165.6 M parameters): microseconds and TB/s of the
30 bytes per parameter the update moves (fp32 p, m, v, g in; p, m, v + bf16 p out).  MAGMA_ADAMW_VARIANT selects the form."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from magma_amd import ops
dev = torch.device("cuda:0")
n = int(os.environ.get("N", 165585296))
p, m, v, g = (torch.randn(n, device=dev) for _ in range(4))
v.abs_()
pb = torch.empty(n, dtype=torch.bfloat16, device=dev)
ns = torch.ones(1, device=dev)
def t(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
us = t(lambda: ops.adamw(p, m, v, g, pb, 1e-4, 0.9, 0.95, 1e-8, 0.01, 3, 1.0, ns, 1.0))
us2 = t(lambda: ops.sumsq(g, ns)) if hasattr(ops, "sumsq") else None
print(json.dumps({"n": n, "adamw_us": us, "adamw_TBps": 30 * n / us / 1e6,
                  "sumsq_us": us2, "sumsq_TBps": None if us2 is None else 4 * n / us2 / 1e6}))
