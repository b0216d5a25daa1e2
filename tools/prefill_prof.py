"""Driver for rocprofv3 --kernel-trace: PHASE=prefill|encoder, N iterations of just that phase."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from magma_amd import Magma

dev = torch.device("cuda:0")
model = Magma("MAGMA_v1", device=dev); model.eval()
B, res = 8, int(os.environ.get("RES", 224))
images = torch.randn(B, 3, res, res, device=dev).to(torch.bfloat16)
prompt = torch.randint(0, 50256, (B, 8), device=dev)
phase = os.environ.get("PHASE", "prefill")
with torch.no_grad():
    emb = model.embed([images, prompt])
    for it in range(int(os.environ.get("ITERS", 4))):
        if it == int(os.environ.get("ITERS", 4)) - 1:     # marker for tools/trace_by_grid.py: only the last iteration is summarised
            from magma_amd import ops
            torch.cuda.synchronize()
            ops.cast_f32_bf16(torch.zeros(64, device=dev), torch.zeros(64, dtype=torch.bfloat16, device=dev))
            torch.cuda.synchronize()
        if phase == "prefill":
            model.lm(inputs_embeds=emb, use_cache=True, cache_hint=32, reuse_cache=True)
        else:
            model.image_prefix(images)
    torch.cuda.synchronize()
