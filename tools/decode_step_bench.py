"""Token-step time of the full-size MAGMA_v1 decode graph (B = 8) -- for A/B runs of decode tuning knobs (env vars)."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from magma_amd import Magma
dev = torch.device("cuda:0")
torch.manual_seed(1234)
model = Magma("MAGMA_v1", device=dev); model.eval()
eng = model.lm.engine
B = int(os.environ.get("DB", 8))
g = torch.Generator(device=dev).manual_seed(1)
images = torch.randn(B, 3, 224, 224, device=dev, generator=g).to(torch.bfloat16)
prompt = torch.randint(0, 50256, (B, 8), device=dev, generator=g)
emb = model.embed([images, prompt])
out = model.lm(inputs_embeds=emb, use_cache=True, cache_hint=200)
cache = out.past_key_values
tok = out.logits[:, -1].argmax(-1, keepdim=True)
for _ in range(3):
    eng.decode(tok, cache)
if os.environ.get("TRACE_MARK") == "1":      # marker for tools/trace_by_grid.py: only the replays below are summarised
    from magma_amd import ops
    torch.cuda.synchronize(); ops.cast_f32_bf16(torch.zeros(64, device=dev), torch.zeros(64, dtype=torch.bfloat16, device=dev)); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
res = []
for rep in range(3):
    torch.cuda.synchronize(); e0.record()
    for _ in range(40):
        eng.decode(tok, cache)
    e1.record(); torch.cuda.synchronize()
    res.append(e0.elapsed_time(e1) / 40)
print(json.dumps({"knobs": {k: v for k, v in os.environ.items() if k.startswith("MAGMA_")}, "token_step_ms": min(res), "all": res,
                  "hbm_frac": 12.156e9 / (min(res) * 1e-3) / 8e12}))
# optional sweep of the fused ln_1+qkv+fc_in GEMV's variant (nt | waves << 4 | kc << 8) inside this process:
#   DEC_IN_VARIANTS=2177,4225,... python tools/decode_step_bench.py
for v in [int(x) for x in os.environ.get("DEC_IN_VARIANTS", "").split(",") if x]:
    eng._dec_in_variant = v
    cache.decode_state.graphs.clear()
    cache.pos = int(emb.shape[1]); cache.d_pos.fill_(cache.pos)       # same context length for every variant
    try:
        for _ in range(3):
            eng.decode(tok, cache)
        best = 1e9
        for rep in range(3):
            cache.pos = int(emb.shape[1]); cache.d_pos.fill_(cache.pos)
            torch.cuda.synchronize(); e0.record()
            for _ in range(40):
                eng.decode(tok, cache)
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 40)
        print(json.dumps({"dec_in_variant": {"nt": v & 15, "waves": (v >> 4) & 15, "kc": v >> 8}, "token_step_ms": best}))
    except Exception as e:  # noqa: BLE001
        print(json.dumps({"dec_in_variant": v, "error": str(e)[:200]}))


# in-process sweep of environment knobs the library reads per call (MAGMA_DECODE_PIPE, ...):
#   DEC_ENV_SWEEP="MAGMA_DECODE_PIPE=16;MAGMA_DECODE_PIPE=8,DEC_IN=66689;-" python tools/decode_step_bench.py
for setting in [x for x in os.environ.get("DEC_ENV_SWEEP", "").split(";") if x]:
    kv = dict(item.split("=") for item in setting.split(",")) if setting != "-" else {}
    eng._dec_in_variant = int(kv.pop("DEC_IN", 0))      # nt | waves << 4 | kc << 8 | pipelined << 16 of the ln_1+qkv+fc_in GEMV
    eng._dec_dn_variant = int(kv.pop("DEC_DN", 0))      # the same for the adapter-down GEMV
    eng._dec_cat_variant = int(kv.pop("DEC_CAT", 0))    # and for the [W_out | W_up] GEMV
    kv_env = dict(kv)
    for k, v in kv_env.items():
        os.environ[k] = v
    cache.decode_state.graphs.clear()
    best = 1e9
    try:
        cache.pos = int(emb.shape[1]); cache.d_pos.fill_(cache.pos)
        for _ in range(3):
            eng.decode(tok, cache)
        for rep in range(4):
            cache.pos = int(emb.shape[1]); cache.d_pos.fill_(cache.pos)
            torch.cuda.synchronize(); e0.record()
            for _ in range(40):
                eng.decode(tok, cache)
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 40)
        print(json.dumps({"env": kv_env, "dec_in": eng._dec_in_variant, "dec_dn": eng._dec_dn_variant, "dec_cat": eng._dec_cat_variant, "token_step_ms": best}))
    except Exception as e:  # noqa: BLE001
        print(json.dumps({"env": kv_env, "dec_in": eng._dec_in_variant, "error": str(e)[:200]}))
    for k in kv_env:
        os.environ.pop(k, None)
