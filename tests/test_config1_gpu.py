"""BASELINE config[0] at its STATED shape (reference README.md:84, magma/magma.py:176-212): the shipped MAGMA_v1.yml, one 224^2
image file + an 8-token prompt -> preprocess_inputs (resize to the model-native 384^2, 144 prefix tokens) -> (1, 152, 4096) ->
forward.  The reference runs this on device='cpu'; this build has no CPU execution path, so the CPU side of the
comparison is the oracle and the product side runs on the GPU -- one GPT-J block deep, everything else at full size
(RN50x16 trunk at 384^2, d 4096, ff 16384, V 50258)."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import fullwidth_common as F  # noqa: E402

pytestmark = pytest.mark.gpu
BF16 = torch.bfloat16
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def rel(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return float((a - b).norm() / (b.norm() + 1e-12))


def test_config1_preprocess_inputs_to_forward(dev, tmp_path):
    import PIL.Image as I
    from magma_amd import ImageInput, Magma
    from magma_amd.language_model import GPTJConfig
    from oracle.model import embed, lm_forward, magma_forward
    from oracle.preprocess import clip_preprocess_u8
    cfg = F.full_width_config()
    p = F.full_width_params(cfg)
    model = Magma(os.path.join(ROOT, "configs", "MAGMA_v1.yml"), device=dev,
                  lm_config=GPTJConfig(num_layers=1, vocab_size=50258))
    missing, unexpected = model.load_checkpoint_state(p)
    assert not missing and not unexpected, (missing[:4], unexpected[:4])
    model.eval()
    assert model.image_prefix.enc.input_resolution == 384 and model.image_prefix_seq_len == 144
    # a 224 x 224 RGB file (smooth + noise so that the bicubic taps matter)
    rng = np.random.RandomState(3)
    yy, xx = np.mgrid[0:224, 0:224]
    arr = np.stack([(xx * 255 / 223), (yy * 255 / 223), ((xx + yy) % 256)], -1) * 0.7 + rng.rand(224, 224, 3) * 76
    path = tmp_path / "image.png"
    I.fromarray(arr.clip(0, 255).astype(np.uint8)).save(path)
    prompt = "Describe"                                       # 8 tokens under the byte-level stand-in tokenizer
    assert model.tokenizer.encode(prompt, return_tensors="pt").shape == (1, 8)

    inputs = [ImageInput(str(path)), prompt]
    emb = model.preprocess_inputs(inputs)                     # reference magma.py:176-193: mutates the caller's list
    assert emb.shape == (1, 152, 4096) and emb.dtype == BF16
    assert inputs[0].shape == (1, 3, 384, 384) and inputs[1].shape == (1, 8) and inputs[1].dtype == torch.int64
    # pixels: bit-exact against the integer restatement of clip_preprocess (itself pinned to PIL, tests/test_oracle_pins.py)
    ref_img = torch.from_numpy(clip_preprocess_u8(np.array(I.open(path).convert("RGB")), 384))[None]
    assert torch.equal(inputs[0].float().cpu(), ref_img)
    with torch.no_grad():
        ref_emb = embed(p, cfg, [ref_img, inputs[1].cpu()])
        pb = {k: (v.to(BF16) if v.is_floating_point() else v) for k, v in p.items()}
        emb_b = embed(pb, cfg, [ref_img.to(BF16), inputs[1].cpu()])
        e, eb = rel(emb, ref_emb), rel(emb_b, ref_emb)
        print(f"embeddings (1,152,4096): HIP {e:.3e}, eager bf16 {eb:.3e}")
        assert ref_emb.shape == (1, 152, 4096) and e <= 2 * eb + 5e-3
        # LM logits on the SAME (bf16-representable) embeddings
        x = emb.float().cpu()
        r = lm_forward(F.lm_only(p), cfg, inputs_embeds=x)
        rb = lm_forward(F.lm_only(pb), cfg, inputs_embeds=x.to(BF16))
        got = model.lm(inputs_embeds=emb).logits
        e, eb = rel(got, r["logits"]), rel(rb["logits"], r["logits"])
        print(f"logits (1,152,50258): HIP {e:.3e}, eager bf16 {eb:.3e}")
        assert got.shape == (1, 152, 50258) and e <= 2 * eb + 4e-3
        # Magma.forward(images, captions): captions padded to seq_len 2048 (reference magma.py:238-276)
        caps = torch.full((1, model.seq_len), model.eos_token, dtype=torch.int64)
        caps[0, :19] = torch.randint(0, 255, (19,), generator=torch.Generator().manual_seed(1))
        out = model(images=inputs[0], captions=caps)
        ref = magma_forward(p, cfg, ref_img, caps)
        refb = magma_forward(pb, cfg, ref_img.to(BF16), caps)
        assert torch.equal(out.labels.cpu(), ref["labels"])
        l, lr, lb = float(out.loss), float(ref["loss"]), float(refb["loss"])
        print("loss", l, lr, lb)
        assert abs(l - lr) <= 2 * abs(lb - lr) + 3e-3 * abs(lr)
