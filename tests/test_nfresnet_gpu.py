"""encoder_name "nfresnet50" (reference magma/image_encoders.py:31-45, magma/image_prefix.py:17,67-72,96-101): timm's
NF-ResNet-50 + the pooled ImagePrefix branch on the HIP kernels against the oracle restatement (oracle/nfnet.py; timm is
un-vendored and absent: parity unpinned, see its header).  Tolerance: 2 x eager-bf16 + floor, as the other encoders."""
import pytest
import torch

pytestmark = pytest.mark.gpu
BF16 = torch.bfloat16


def rel(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return float((a - b).norm() / (b.norm() + 1e-12))


def bf16_params(p):
    return {k: (v.to(BF16) if v.is_floating_point() else v) for k, v in p.items()}


def test_weight_standardize_kernel(dev):
    """mg_weight_standardize_bf16 against the oracle's statement of timm ScaledStdConv2d, both column orders."""
    from magma_amd import ops
    from oracle.nfnet import RELU_GAMMA, standardized_weight
    g = torch.Generator().manual_seed(0)
    for cout, cin, k in [(64, 3, 7), (64, 256, 1), (128, 128, 3), (40, 24, 3)]:
        w = (torch.randn(cout, cin, k, k, generator=g) + 0.5).to(BF16)
        gain = (1 + 0.2 * torch.randn(cout, 1, 1, 1, generator=g)).to(BF16)
        ref = standardized_weight(w.float(), gain.float(), 1e-5).reshape(cout, -1)
        fan_in = cin * k * k
        got = ops.weight_standardize(w.cuda(), gain.cuda().reshape(-1), RELU_GAMMA * fan_in ** -0.5, 1e-5, ldo=(fan_in + 15) // 8 * 8)
        assert bool((got[:, fan_in:] == 0).all())
        assert rel(got[:, :fan_in], ref) < 4e-3                 # bf16 rounding of the output only
        if k == 3:
            got2 = ops.weight_standardize(w.cuda(), gain.cuda().reshape(-1), RELU_GAMMA * fan_in ** -0.5, 1e-5, to_khwc=True)
            ref2 = standardized_weight(w.float(), gain.float(), 1e-5).permute(0, 2, 3, 1).reshape(cout, -1)
            assert rel(got2[:, :fan_in], ref2) < 4e-3


def test_pool_and_im2col_kernels(dev):
    from magma_amd import ops
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(1)
    x = torch.randn(2, 16, 10, 14, generator=g).to(BF16)                   # NCHW reference tensors, NHWC for the kernels
    nhwc = x.permute(0, 2, 3, 1).contiguous().cuda()
    mp = ops.maxpool3x3s2(nhwc).permute(0, 3, 1, 2).float().cpu()
    assert torch.equal(mp, F.max_pool2d(x.float(), 3, stride=2, padding=1))
    ss = ops.subsample2(nhwc).permute(0, 3, 1, 2).cpu()
    assert torch.equal(ss, x[:, :, ::2, ::2])
    rm = ops.relu_mean_rows(nhwc.view(2, 140, 16)).float().cpu()
    assert rel(rm, F.relu(x.float()).mean(dim=(2, 3))) < 4e-3
    img = torch.randn(2, 3, 18, 22, generator=g).to(BF16)
    cols = ops.im2col_nchw(img.cuda(), 7, 2, 3, 160).float().cpu()
    ref = F.unfold(img.float(), 7, padding=3, stride=2).transpose(1, 2).reshape(-1, 147)    # column (c, ky, kx)
    assert torch.equal(cols[:, :147], ref) and bool((cols[:, 147:] == 0).all())


@pytest.mark.parametrize("res", [128, 256])
def test_nfresnet50_encoder_and_pooled_prefix(dev, res):
    """The full architecture (53 scaled-std convs, 23.5 M parameters) at two resolutions, then the pooled prefix on top."""
    from magma_amd.image_encoders import NFResNet50
    from magma_amd.image_prefix import ImagePrefix
    from magma_amd.testing import tiny_multimodal_config
    from oracle.model import pooled_prefix_fwd
    from oracle.nfnet import NFResNetConfig, encoder_fwd, init_params
    c = NFResNetConfig()
    p = init_params(c, seed=3)
    enc = NFResNet50(res, device=dev, dtype=BF16)
    enc.load_state_dict({k[len("image_prefix.enc."):]: t for k, t in p.items()}, strict=True)
    enc.invalidate_packed()
    x = torch.randn(2, 3, res, res, generator=torch.Generator().manual_seed(0)).to(BF16).float()
    with torch.no_grad():
        ref = encoder_fwd(p, c, x)
        eb = rel(encoder_fwd(bf16_params(p), c, x.to(BF16)), ref)
        got = enc(x.cuda())
    print(f"nf_resnet50 @{res}: HIP {rel(got, ref):.3e}, eager bf16 {eb:.3e}")
    assert got.shape == ref.shape == (2, 2048)
    assert rel(got, ref) <= 2 * eb + 5e-3, (rel(got, ref), eb)
    d, s = 512, 2
    cfg = tiny_multimodal_config(encoder_name="nfresnet50", image_seq_len=s, image_size=res)
    ip = ImagePrefix(cfg, out_dim=d, device=dev, dtype=BF16, enc=enc)
    assert ip.pooled and ip.out_seq_len == s and ip.proj.weight.shape == (s * d, 2048)
    g = torch.Generator().manual_seed(1)
    pp = dict(p)
    pp["image_prefix.proj.weight"] = torch.randn(s * d, 2048, generator=g) * 2048 ** -0.5
    pp["image_prefix.proj.bias"] = torch.randn(s * d, generator=g) * 0.02
    pp["image_prefix.ln.weight"] = 1.0 + torch.randn(d, generator=g) * 0.05
    pp["image_prefix.ln.bias"] = torch.randn(d, generator=g) * 0.02
    with torch.no_grad():
        ip.proj.weight.copy_(pp["image_prefix.proj.weight"]); ip.proj.bias.copy_(pp["image_prefix.proj.bias"])
        ip.ln.weight.copy_(pp["image_prefix.ln.weight"]); ip.ln.bias.copy_(pp["image_prefix.ln.bias"])
    ip.invalidate_packed()
    ip.eval()
    with torch.no_grad():
        ref2 = pooled_prefix_fwd(pp, d, s, ref)
        ppb = bf16_params(pp)
        eb2 = rel(pooled_prefix_fwd(ppb, d, s, encoder_fwd(ppb, c, x.to(BF16))), ref2)
        got2 = ip(x.cuda())
    assert got2.shape == (2, s, d)
    assert rel(got2, ref2) <= 2 * eb2 + 5e-3, (rel(got2, ref2), eb2)


def test_magma_with_nfresnet50_encoder(dev):
    """Magma built from a config that selects encoder_name "nfresnet50": checkpoint keys load by name, embed() yields
    image_seq_len prefix tokens per image, generate() runs; training the prefix + adapters on the frozen encoder steps."""
    from magma_amd.config import MultimodalConfig
    from magma_amd.language_model import GPTJConfig
    from magma_amd.magma import Magma
    from magma_amd.train_engine import MagmaEngine
    from oracle.nfnet import NFResNetConfig, init_params
    cfg = MultimodalConfig(batch_size=2, train_steps=1, encoder_name="nfresnet50", image_seq_len=4, image_size=128,
                           freeze_img_encoder=True, adapter_config={"mlp": {"adapter_type": "normal", "downsample_factor": 4}})
    lm_cfg = GPTJConfig(vocab_size=1056, hidden_size=512, num_layers=2, num_heads=2, rotary_dim=64, intermediate_size=2048,
                        max_position_embeddings=128)
    model = Magma(cfg, device=dev, lm_config=lm_cfg)
    missing, unexpected = model.load_checkpoint_state(init_params(NFResNetConfig(), seed=5))
    assert not unexpected and not any(k.startswith("image_prefix.enc.") for k in missing), (missing[:4], unexpected[:4])
    model.eval()
    assert model.image_prefix.pooled and model.image_prefix_seq_len == 4
    emb = model.embed([torch.randn(2, 3, 128, 128), torch.randint(0, 1000, (2, 5))])
    assert emb.shape == (2, 4 + 5, 512) and bool(torch.isfinite(emb.float()).all())
    toks = model.generate(emb, max_steps=3, temperature=0.0, decode=False, stop_on_eos=False)
    assert toks.shape == (2, 9 + 3)
    # the transform of the non-CLIP encoders (reference transforms.py:65-84) feeds it
    import numpy as np
    import PIL.Image as I
    img = model.transforms(I.fromarray(np.random.default_rng(0).integers(0, 256, (200, 300, 3), dtype=np.uint8)))
    assert img.shape == (1, 3, 128, 128) and float(img.min()) >= 0.0 and float(img.max()) <= 1.0
    # frozen encoder: prefix + adapters train on top of it
    model.config.gradient_accumulation_steps = 1
    eng = MagmaEngine(model)
    eng.train()
    caps = torch.full((2, model.seq_len), model.eos_token, dtype=torch.int64)
    caps[:, :9] = torch.randint(0, 1000, (2, 9))
    out = eng(torch.randn(2, 3, 128, 128).to(dev), caps.to(dev))
    eng.backward(out.loss)
    eng.step()
    assert bool(torch.isfinite(out.loss))
