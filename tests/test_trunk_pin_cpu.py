"""Independent statement of the CLIP ModifiedResNet trunk for the oracle (openai/CLIP is un-vendored and
unavailable offline, so ``oracle.encoder_fwd`` cannot be pinned to CLIP itself -- DESIGN.md section 2 says so).

What CAN be checked here: the functional restatement (oracle/model.py, F.conv2d / F.batch_norm driven by a flat
parameter dict) against stock ``torch.nn`` modules -- nn.Conv2d / nn.BatchNorm2d / nn.AvgPool2d / nn.ReLU instances
assembled from the Bottleneck / ModifiedResNetTrunk MODULE TREE of magma_amd/image_encoders.py (CLIP's parameter
names and constructor geometry: strides, paddings, the avg-pool placement of the anti-aliased stride-2 blocks) and run
by nn.Module.__call__ on the CPU, loading the oracle's state dict BY NAME with strict=True.  Two statements written
from different ends agree, every parameter name resolves, and the CLIP constants (reference image_prefix.py:13,20)
hold."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from oracle import model as O


def torch_nn_trunk_forward(trunk, x):
    """Stock torch.nn execution of the module tree (CLIP ModifiedResNet.forward minus attnpool)."""
    relu = nn.ReLU()
    x = relu(trunk.bn1(trunk.conv1(x)))
    x = relu(trunk.bn2(trunk.conv2(x)))
    x = relu(trunk.bn3(trunk.conv3(x)))
    x = nn.AvgPool2d(2)(x)
    for layer in (trunk.layer1, trunk.layer2, trunk.layer3, trunk.layer4):
        for blk in layer:
            out = relu(blk.bn1(blk.conv1(x)))
            out = relu(blk.bn2(blk.conv2(out)))
            if blk.stride > 1:
                out = nn.AvgPool2d(blk.stride)(out)
            out = blk.bn3(blk.conv3(out))
            identity = blk.downsample(x) if blk.downsample is not None else x      # Sequential(AvgPool2d, Conv2d, BatchNorm2d)
            x = relu(out + identity)
    return x.flatten(2).transpose(1, 2)          # reference image_encoders.py:72-74 "b d h w -> b (h w) d"


def _check(cfg, res, B=2, seed=4):
    from magma_amd.image_encoders import ModifiedResNetTrunk
    p = O.init_params(O.OracleConfig(n_layer=0, vocab_in=8, vocab_out=8, enc_width=cfg.enc_width, enc_layers=cfg.enc_layers), seed=seed)
    trunk = ModifiedResNetTrunk(cfg.enc_layers, cfg.enc_width, res, device="cpu", dtype=torch.float32)
    sd = {k[len("image_prefix.enc."):]: v for k, v in p.items() if k.startswith("image_prefix.enc.")}
    own = trunk.state_dict()
    for k in own:                                     # BatchNorm's step counters are not part of the oracle's dict
        if k.endswith("num_batches_tracked"):
            sd[k] = own[k]
    trunk.load_state_dict(sd, strict=True)            # every CLIP parameter name resolves, none is left over
    trunk.eval()
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, 3, res, res, generator=g)
    with torch.no_grad():
        a = torch_nn_trunk_forward(trunk, x)
        b = O.encoder_fwd(p, cfg, x)
    assert a.shape == b.shape == (B, (res // 32) ** 2, cfg.enc_width * 32)
    assert torch.allclose(a, b, atol=2e-5, rtol=1e-4), float((a - b).abs().max())
    return trunk, p


def test_oracle_trunk_equals_torch_nn_module_tree_reduced():
    _check(O.OracleConfig.tiny(), 64)
    _check(O.OracleConfig.tiny(enc_width=32, enc_layers=(2, 1, 3, 2)), 96)


def test_oracle_trunk_equals_torch_nn_module_tree_rn50x16():
    """The full RN50x16 geometry ((6,8,18,8), width 96) at 224^2, B = 1."""
    cfg = O.OracleConfig.magma_v1()
    trunk, _ = _check(cfg, 224, B=1)
    n = sum(p.numel() for p in trunk.parameters())
    assert abs(n / 1e6 - 136.2) < 0.05 and trunk.out_dim == 3072        # SURVEY 8a a4; reference image_prefix.py:20
    assert (384 // 32) ** 2 == 144                                     # reference image_prefix.py:13


def test_batch_statistics_mode_matches_torch_nn():
    """after the first eval phase the reference's tower runs BatchNorm on batch statistics.  The oracle's
    train-mode statement (bn_train=True) against the same nn modules in .train() mode."""
    from magma_amd.image_encoders import ModifiedResNetTrunk
    cfg = O.OracleConfig.tiny()
    p = O.init_params(cfg, seed=2)
    trunk = ModifiedResNetTrunk(cfg.enc_layers, cfg.enc_width, 64, device="cpu", dtype=torch.float32)
    sd = {k[len("image_prefix.enc."):]: v.clone() for k, v in p.items() if k.startswith("image_prefix.enc.")}
    for k, v in trunk.state_dict().items():
        if k.endswith("num_batches_tracked"):
            sd[k] = v
    trunk.load_state_dict(sd, strict=True)
    trunk.train()
    x = torch.randn(4, 3, 64, 64, generator=torch.Generator().manual_seed(1))
    with torch.no_grad():
        a = torch_nn_trunk_forward(trunk, x)
        b = O.encoder_fwd({k: v.clone() for k, v in p.items()}, cfg, x, bn_train=True)
    assert torch.allclose(a, b, atol=2e-5, rtol=1e-4), float((a - b).abs().max())
