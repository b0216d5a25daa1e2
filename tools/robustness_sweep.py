"""Odd-shape sweep on a full-width 2-block model: generate() must run, stay finite and agree between the graph
decode and a fresh uncached forward at the last position, for batch sizes / prompt lengths / resolutions the
benchmarks do not touch."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from magma_amd import Magma
from magma_amd.language_model import GPTJConfig

dev = torch.device("cuda:0")
torch.manual_seed(0)
model = Magma("MAGMA_v1", device=dev, lm_config=GPTJConfig(num_layers=2, vocab_size=50258)); model.eval()
res_out = []
for (B, T, res, steps) in [(1, 1, 224, 5), (16, 8, 224, 5), (3, 600, 224, 9), (5, 37, 384, 4), (2, 1500, 384, 3), (8, 8, 160, 2)]:
    g = torch.Generator(device=dev).manual_seed(B * 1000 + T)
    images = torch.randn(B, 3, res, res, device=dev, generator=g).to(torch.bfloat16)
    prompt = torch.randint(0, 50256, (B, T), device=dev, generator=g)
    ok, err = True, None
    try:
        with torch.no_grad():
            emb = model.embed([images, prompt])
            toks = model.generate(emb, max_steps=steps, temperature=0.0, decode=False, stop_on_eos=False)
            new = toks[:, emb.shape[1]:]
            assert new.shape == (B, steps) and bool((new >= 0).all()) and bool((new < 50258).all())
            # replay: uncached forward over [emb | wte(new[:-1])] must pick the same last token where the margin is clear
            ext = torch.cat([emb, model.word_embedding(new[:, :-1]).to(emb.dtype)], dim=1)
            logits = model.lm(inputs_embeds=ext, use_cache=True).logits[:, -1].float()
            assert torch.isfinite(logits).all()
            top2 = logits.topk(2, -1).values
            clear = (top2[:, 0] - top2[:, 1]) > 0.05 * logits.std(-1)
            same = bool((logits.argmax(-1)[clear] == new[:, -1][clear]).all())
            assert same, "cached decode and uncached forward disagree on a clear-margin token"
            # sampled path runs too
            model.generate(emb, max_steps=2, temperature=0.7, top_k=5, top_p=0.9, decode=False, stop_on_eos=False)
    except Exception as e:  # noqa: BLE001
        ok, err = False, repr(e)[:300]
    res_out.append({"B": B, "T": T, "res": res, "S0": int(emb.shape[1]) if 'emb' in dir() else None, "steps": steps, "ok": ok, "err": err})
    print(json.dumps(res_out[-1]), flush=True)
print("ALL_OK" if all(r["ok"] for r in res_out) else "FAILED")
