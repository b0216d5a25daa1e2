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


@pytest.mark.parametrize("res", [96, 128, 224, 256])      # 96, 224: multiples of 32 that are not multiples of 64 (odd final map)
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


def test_nfnet_backward_kernels(dev):
    """weight_standardize_bwd / maxpool / subsample / relu-mean backward kernels against autograd on the same operands."""
    import torch.nn.functional as F
    from magma_amd import ops
    from oracle.nfnet import RELU_GAMMA, standardized_weight
    g = torch.Generator().manual_seed(2)
    for cout, cin, k in [(64, 3, 7), (48, 64, 1), (32, 24, 3)]:
        w = (torch.randn(cout, cin, k, k, generator=g) + 0.4).to(BF16)
        gain = (1 + 0.2 * torch.randn(cout, 1, 1, 1, generator=g)).to(BF16)
        fan_in = cin * k * k
        ld = (fan_in + 15) // 8 * 8
        dwh = torch.zeros(cout, ld)
        dwh[:, :fan_in] = torch.randn(cout, fan_in, generator=g)
        wf, gf = w.float().requires_grad_(True), gain.float().requires_grad_(True)
        (standardized_weight(wf, gf, 1e-5).reshape(cout, -1) * dwh[:, :fan_in] * 0.7).sum().backward()
        dw = torch.zeros(cout, fan_in, device=dev)
        dg = torch.zeros(cout, device=dev)
        ops.weight_standardize_bwd(w.cuda(), gain.cuda().reshape(-1), dwh.cuda(), dw, dg, RELU_GAMMA * fan_in ** -0.5, 1e-5, dmult=0.7)
        assert rel(dw, wf.grad.reshape(cout, -1)) < 2e-4 and rel(dg, gf.grad.reshape(-1)) < 2e-4
    x = torch.randn(2, 16, 10, 14, generator=g).to(BF16)
    dy = torch.randn(2, 16, 5, 7, generator=g).to(BF16)
    xf = x.float().requires_grad_(True)
    F.max_pool2d(xf, 3, stride=2, padding=1).backward(dy.float())
    got = ops.maxpool3x3s2_bwd(x.permute(0, 2, 3, 1).contiguous().cuda(), dy.permute(0, 2, 3, 1).contiguous().cuda())
    assert rel(got.permute(0, 3, 1, 2), xf.grad) < 4e-3          # bf16 rounding of sums of up to 4 window gradients
    ups = ops.subsample2_bwd(dy.permute(0, 2, 3, 1).contiguous().cuda(), 10, 14).permute(0, 3, 1, 2).cpu()
    ref = torch.zeros(2, 16, 10, 14, dtype=BF16)
    ref[:, :, ::2, ::2] = dy
    assert torch.equal(ups, ref)
    feats = torch.randn(2, 16, generator=g).to(BF16)
    xf = x.float().requires_grad_(True)
    (F.relu(xf).mean(dim=(2, 3)) * feats.float()).sum().backward()
    got = ops.relu_mean_rows_bwd(x.permute(0, 2, 3, 1).reshape(2, 140, 16).contiguous().cuda(), feats.cuda())
    assert rel(got.view(2, 10, 14, 16).permute(0, 3, 1, 2), xf.grad) < 4e-3


def test_nfresnet50_train_gradients(dev):
    """Training with encoder_name "nfresnet50" and the encoder UNFROZEN (the reference's default freeze_img_encoder: false):
    loss and the gradient of every trainable tensor -- all 53 scaled-std convs (weight, bias, gain, through the weight
    standardisation), the pooled prefix Linear + LayerNorm, the LM adapters -- against autograd through the fp32 oracle
    (oracle/nfnet.py: parity unpinned to timm, see its header).  Tolerance as tests/test_train_gpu.py."""
    import torch.nn.functional as F
    from magma_amd.config import MultimodalConfig
    from magma_amd.language_model import GPTJConfig
    from magma_amd.magma import Magma
    from magma_amd.train_engine import MagmaEngine
    from oracle.model import OracleConfig, build_labels, init_params as init_lm, lm_forward, pooled_prefix_fwd
    from oracle.nfnet import NFResNetConfig, encoder_fwd, init_params
    s_img, d, res = 2, 512, 128
    mcfg = MultimodalConfig(batch_size=2, train_steps=1, encoder_name="nfresnet50", image_seq_len=s_img, image_size=res,
                            freeze_img_encoder=False, use_image_embed_layernorm=True, image_embed_dropout_prob=0.1,
                            adapter_config={"mlp": {"adapter_type": "normal", "downsample_factor": 4}},
                            image_enc_lr=2.0e-6, lr_decay_iters=1000)
    lm_cfg = GPTJConfig(vocab_size=1056, hidden_size=d, num_layers=2, num_heads=2, rotary_dim=64, intermediate_size=2048,
                        max_position_embeddings=128)
    model = Magma(mcfg, device=dev, lm_config=lm_cfg)
    cfg, c = OracleConfig.tiny(n_positions=128), NFResNetConfig()
    params = {k: t for k, t in init_lm(cfg, seed=31).items() if not k.startswith("image_prefix.")}
    for k in params:
        if ".adapter." in k:
            params[k] = params[k] * 20
    params.update(init_params(c, seed=5))
    g = torch.Generator().manual_seed(9)
    params["image_prefix.proj.weight"] = torch.randn(s_img * d, 2048, generator=g) * 2048 ** -0.5
    params["image_prefix.proj.bias"] = torch.randn(s_img * d, generator=g) * 0.02
    params["image_prefix.ln.weight"] = 1.0 + torch.randn(d, generator=g) * 0.05
    params["image_prefix.ln.bias"] = torch.randn(d, generator=g) * 0.02
    missing, unexpected = model.load_checkpoint_state(params)
    assert not unexpected and not missing, (missing[:4], unexpected[:4])
    model.config.gradient_accumulation_steps = 1
    eng = MagmaEngine(model)
    eng.train()
    B, S = 2, model.seq_len
    images = torch.randn(B, 3, res, res, generator=g).to(BF16).float()
    caps = torch.full((B, S), cfg.eos_token, dtype=torch.int64)
    caps[0, :23] = torch.randint(0, 1000, (23,), generator=g)
    caps[1, :11] = torch.randint(0, 1000, (11,), generator=g)
    mask = (torch.rand(B, s_img, d, generator=g) < 0.9).float() / 0.9
    names = [k for k in params if ".adapter." in k or k.startswith("image_prefix.")]

    def oracle(dtype):
        p = {k: (t.detach().to(dtype).clone() if t.is_floating_point() else t) for k, t in params.items()}
        for k in names:
            p[k].requires_grad_(True)
        prefix = pooled_prefix_fwd(p, d, s_img, encoder_fwd(p, c, images.to(dtype)), dropout_mask=mask.to(dtype))
        labels = build_labels(s_img, caps, cfg.eos_token)
        words = F.embedding(caps, p["lm.transformer.wte.weight"]).to(prefix.dtype)
        out = lm_forward(p, cfg, inputs_embeds=torch.cat((prefix, words[:, : S - s_img, :]), dim=1), labels=labels)
        out["loss"].backward()
        return float(out["loss"].detach()), {k: p[k].grad.float() for k in names}

    loss_ref, g_ref = oracle(torch.float32)
    loss_bf, g_bf = oracle(BF16)
    out = eng(images.to(dev), caps.to(dev), dropout_mask=mask.to(dev))
    assert abs(float(out.loss) - loss_ref) <= 2 * abs(loss_bf - loss_ref) + 3e-3 * abs(loss_ref), (float(out.loss), loss_ref, loss_bf)
    eng.backward(out.loss)
    name_of = {id(p): n for n, p in model.named_parameters()}
    seen, bad, worst = set(), [], []
    dots = n1 = n2 = bd = b1 = 0.0
    for grp in eng.groups:
        for p in grp.params:
            n = name_of[id(p)]
            n = "lm." + n if n.startswith("transformer.") else n
            if n in seen or n not in g_ref:
                continue
            seen.add(n)
            got, ref, gb = eng.grad_of(p).float().cpu().reshape(-1), g_ref[n].reshape(-1), g_bf[n].reshape(-1)
            e_hip, e_bf = rel(got, ref), rel(gb, ref)
            worst.append((e_hip - 2 * e_bf, n, e_hip, e_bf))
            if e_hip > 2 * e_bf + 3e-2:
                bad.append((n, e_hip, e_bf))
            dots += float((got * ref).sum()); n1 += float((got * got).sum()); n2 += float((ref * ref).sum())
            bd += float((gb * ref).sum()); b1 += float((gb * gb).sum())
    worst.sort(reverse=True)
    print("nf train: loss", float(out.loss), loss_ref, loss_bf, "| worst:", [(n, f"{a:.2e}", f"{b:.2e}") for _, n, a, b in worst[:5]])
    assert len(seen) == len(g_ref) and len(seen) > 160, (len(seen), len(g_ref), sorted(set(g_ref) - seen)[:5])
    assert not bad, bad[:8]
    cos_hip, cos_bf = dots / (n1 ** 0.5 * n2 ** 0.5), bd / (b1 ** 0.5 * n2 ** 0.5)
    assert 1 - cos_hip <= 2 * (1 - cos_bf) + 1e-3, (cos_hip, cos_bf)
    eng.step()
    eng.eval()
    assert torch.isfinite(eng(images.to(dev), caps.to(dev)).loss)
