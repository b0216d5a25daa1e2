"""The decode GEMV sweep of bench.py (every weight-streaming launch of one token step on the full-size model) for a
rocprofv3 --pmc FETCH_SIZE pass; tools/pmc_table.py turns the counter CSV into profiles/r02_decode_gemv_fetch_table.json."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import decode_gemv_jobs  # noqa: E402
from magma_amd import Magma  # noqa: E402
dev = torch.device("cuda:0")
torch.manual_seed(1234)
model = Magma("MAGMA_v1", device=dev); model.eval()
eng = model.lm.engine
emb = torch.randn(8, 57, eng.d, device=dev).to(torch.bfloat16)
out = model.lm(inputs_embeds=emb, use_cache=True, cache_hint=64)
cache = out.past_key_values
eng.decode(out.logits[:, -1].argmax(-1, keepdim=True), cache)
jobs, wbytes, shapes = decode_gemv_jobs(eng, cache.decode_state)
import json
json.dump({"shapes": shapes, "wbytes": wbytes}, open(os.path.join(ROOT, "gpurun_out", "pmc_sweep_shapes.json"), "w"))
for _ in range(3):
    for fn in jobs:
        fn()
torch.cuda.synchronize()
print("done", len(jobs), wbytes)
