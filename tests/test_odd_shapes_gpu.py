"""Shapes the benchmarks never touch, on a full-width 2-block model: generate() must run, stay in range, and the
graph decode must agree with a fresh uncached forward on every clear-margin token (KV-cache consistency at
batch 1 / 16, 600- and 1500-token prompts, 384-pixel and 160-pixel images)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def wide_model(dev):
    from magma_amd import Magma
    from magma_amd.language_model import GPTJConfig
    torch.manual_seed(0)
    model = Magma("MAGMA_v1", device=dev, lm_config=GPTJConfig(num_layers=2, vocab_size=50258))
    model.eval()
    return model


@pytest.mark.parametrize("B,T,res,steps", [(1, 1, 224, 5), (16, 8, 224, 5), (3, 600, 224, 9), (2, 1500, 384, 3), (8, 8, 160, 2)])
def test_generate_at_odd_shapes(wide_model, dev, B, T, res, steps):
    model = wide_model
    g = torch.Generator(device=dev).manual_seed(B * 1000 + T)
    images = torch.randn(B, 3, res, res, device=dev, generator=g).to(torch.bfloat16)
    prompt = torch.randint(0, 50256, (B, T), device=dev, generator=g)
    with torch.no_grad():
        emb = model.embed([images, prompt])
        assert emb.shape[1] == (res // 32) ** 2 + T
        toks = model.generate(emb, max_steps=steps, temperature=0.0, decode=False, stop_on_eos=False)
        new = toks[:, emb.shape[1]:]
        assert new.shape == (B, steps) and bool((new >= 0).all()) and bool((new < 50258).all())
        ext = torch.cat([emb, model.word_embedding(new[:, :-1]).to(emb.dtype)], dim=1)
        logits = model.lm(inputs_embeds=ext, use_cache=True).logits[:, -1].float()
        assert torch.isfinite(logits).all()
        top2 = logits.topk(2, -1).values
        clear = (top2[:, 0] - top2[:, 1]) > 0.05 * logits.std(-1)
        assert bool((logits.argmax(-1)[clear] == new[:, -1][clear]).all())
        sampled = model.generate(emb, max_steps=2, temperature=0.7, top_k=5, top_p=0.9, decode=False, stop_on_eos=False)
        assert sampled.shape == (B, emb.shape[1] + 2)
