"""Where does one generate() call spend its time?  (encoder / prefix+embed / prefill / decode)"""
import os, sys, time, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from magma_amd import Magma

dev = torch.device("cuda:0")
model = Magma("MAGMA_v1", device=dev); model.eval()
B, res = 8, int(os.environ.get("RES", 224))
images = torch.randn(B, 3, res, res, device=dev).to(torch.bfloat16)
prompt = torch.randint(0, 50256, (B, 8), device=dev)

def timed(fn, n=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): r = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, r

with torch.no_grad():
    t_enc, feats = timed(lambda: model.image_prefix.enc(images))
    t_prefix, _ = timed(lambda: model.image_prefix(images))
    t_embed, emb = timed(lambda: model.embed([images, prompt]))
    t_prefill, out = timed(lambda: model.lm(inputs_embeds=emb, use_cache=True, cache_hint=32, reuse_cache=True))
    t_gen, _ = timed(lambda: model.generate(emb, max_steps=32, temperature=0.0, decode=False, stop_on_eos=False), n=3)
print(json.dumps({"res": res, "encoder_ms": t_enc, "image_prefix_ms": t_prefix, "embed_ms": t_embed, "prefill_ms": t_prefill,
                  "generate32_ms": t_gen, "S0": emb.shape[1]}))

# ---- potential of graph capture for the non-decode phases ----
def graphed(fn):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        r = fn()
    return g, r

with torch.no_grad():
    g_enc, _ = graphed(lambda: model.image_prefix(images))
    t_enc_g, _ = timed(lambda: g_enc.replay())
    g_pre, _ = graphed(lambda: model.lm(inputs_embeds=emb, use_cache=True, cache_hint=32, reuse_cache=True))
    t_pre_g, _ = timed(lambda: g_pre.replay())
print(json.dumps({"image_prefix_graph_ms": t_enc_g, "prefill_graph_ms": t_pre_g}))
