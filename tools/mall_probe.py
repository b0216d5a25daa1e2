"""Do the small decode GEMVs run faster when their weights are already in the 256 MiB Infinity Cache?  Same launch on the SAME
layer's weights back to back (hot) vs rotating over the 28 layers (cold), full-size MAGMA_v1, B = 8."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from magma_amd import Magma, ops
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = Magma("MAGMA_v1", device=dev); model.eval()
eng = model.lm.engine
emb = torch.randn(8, 57, eng.d, device=dev).to(torch.bfloat16)
out = model.lm(inputs_embeds=emb, use_cache=True, cache_hint=64)
cache = out.past_key_values
eng.decode(out.logits[:, -1].argmax(-1, keepdim=True), cache)
st = cache.decode_state
L = eng.layers

def timeit(fn, n):
    for i in range(4): fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n): fn(i)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

def sk2(ly):
    t = st.t[:, : ly.mlp_adapter[0].N]
    ops.gemm_skinny2((st.ctx, ly.out, st.a, {}), (st.m, ly.mlp_adapter[0], t, {"act": ops.MG_ACT_RELU}))
def up(ly):
    t = st.t[:, : ly.mlp_adapter[0].N]
    ops.gemm_skinny(t, ly.mlp_adapter[1], out=st.xb, residuals=(st.m, st.a, st.xa))
def fco(ly):
    ops.gemm_skinny(st.h, ly.fc_out, out=st.m)
res = {}
for name, fn in (("out_proj||dn (42 MB)", sk2), ("adapter-up (8.4 MB)", up), ("fc_out (134 MB)", fco)):
    g = {}
    for mode in ("hot", "cold"):
        gr = torch.cuda.CUDAGraph()
        seq = [L[0]] * 28 if mode == "hot" else L
        for ly in seq: fn(ly)
        torch.cuda.synchronize()
        with torch.cuda.graph(gr):
            for ly in seq: fn(ly)
        g[mode] = timeit(lambda i: gr.replay(), 10) / 28
    res[name] = g
print(json.dumps(res))
