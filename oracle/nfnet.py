"""Oracle (TEST INFRASTRUCTURE): functional restatement of timm's ``nf_resnet50`` as the reference uses it --
``nn.Sequential(nn.Sequential(*list(timm.create_model("nf_resnet50").children())[:-1]), nn.AdaptiveAvgPool2d((1, 1)))``
(reference magma/image_encoders.py:31-45) feeding the pooled ImagePrefix branch (magma/image_prefix.py:17,67-72,96-101).

**Parity unpinned**: ``timm`` is an un-vendored dependency (reference requirements.txt) and is not installed in this image; there
are no weights, no source and no independent implementation offline.  The arithmetic below restates the PUBLISHED algorithm
(Brock et al., "Characterizing signal propagation to close the performance gap in unnormalized ResNets", ICLR 2021; timm
``models/nfnet.py``: ``_nfres_cfg(depths=(3, 4, 6, 3))`` -> NfCfg(channels=(256, 512, 1024, 2048), stem_type='7x7_pool', stem_chs=64,
bottle_ratio=0.25, alpha=0.2, act_layer='relu', std_conv_eps=1e-5, gamma_in_act=False) and ``layers/std_conv.py``
``ScaledStdConv2d``):

  * ScaledStdConv2d: W_hat = (W - mean_o) / sqrt(var_o + eps) * gain_o * gamma * fan_in^-0.5 per output channel o (biased variance
    over the fan-in, the F.batch_norm form of current timm), gamma = 1.7139588594436646 for ReLU, conv bias kept, eps = 1e-5;
  * stem '7x7_pool': ScaledStdConv2d(3, 64, 7, stride 2, padding 3) -> MaxPool2d(3, stride 2, padding 1) (no activation);
  * NormFreeBlock (pre-activation bottleneck): out = relu(x) * beta;  shortcut = downsample(out) if the block changes shape
    else x;  out = conv1(out); out = conv2(relu(out)) [3x3, carries the stride]; out = conv3(relu(out));  return out * alpha +
    shortcut;  beta = 1 / sqrt(expected_var): expected_var starts at 1, grows by alpha^2 per block and is reset to 1 (+ alpha^2)
    AFTER the first block of every stage -- a stage's first block still divides by the previous stage's value;
    DownsampleAvg = AvgPool2d(2, stride, ceil_mode=True, count_include_pad=False) when stride > 1, then
    a 1x1 ScaledStdConv2d;
  * final_conv = Identity (num_features = 0), final_act = ReLU, then the reference's AdaptiveAvgPool2d((1, 1)).

State-dict keys as the reference's module tree produces them: ``0.0.conv.*`` (stem), ``0.1.{stage}.{block}.conv{1,2,3}.*``,
``0.1.{stage}.{block}.downsample.conv.*``, each with ``weight``, ``bias``, ``gain`` ([Cout,1,1,1])."""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, List, Tuple

import torch
import torch.nn.functional as F

Params = Dict[str, torch.Tensor]
RELU_GAMMA = 1.7139588594436646          # timm _nonlin_gamma['relu'] = sqrt(2 / (1 - 1/pi))


@dataclass
class NFResNetConfig:
    depths: Tuple[int, ...] = (3, 4, 6, 3)
    channels: Tuple[int, ...] = (256, 512, 1024, 2048)
    stem_chs: int = 64
    bottle_ratio: float = 0.25
    alpha: float = 0.2
    eps: float = 1e-5

    @property
    def out_dim(self) -> int:
        return self.channels[-1]


def conv_specs(c: NFResNetConfig) -> List[Tuple[str, int, int, int]]:
    """(key prefix, cin, cout, kernel) of every ScaledStdConv2d, in forward order."""
    specs = [("0.0.conv", 3, c.stem_chs, 7)]
    prev = c.stem_chs
    for si, depth in enumerate(c.depths):
        out = c.channels[si]
        mid = int(out * c.bottle_ratio)
        for bi in range(depth):
            stride = 2 if (bi == 0 and si > 0) else 1
            pre = f"0.1.{si}.{bi}."
            if prev != out or stride != 1:
                specs.append((pre + "downsample.conv", prev, out, 1))
            specs += [(pre + "conv1", prev, mid, 1), (pre + "conv2", mid, mid, 3), (pre + "conv3", mid, out, 1)]
            prev = out
    return specs


def block_plan(c: NFResNetConfig):
    """[(prefix, stride, has_downsample, beta)] per block -- timm NormFreeNet.__init__'s expected_var bookkeeping."""
    plan, prev = [], c.stem_chs
    expected_var = 1.0          # NOT reset at a stage boundary: the first block of stage s > 0 divides by the variance the
    for si, depth in enumerate(c.depths):      # previous stage accumulated; the reset happens AFTER that block
        out = c.channels[si]
        for bi in range(depth):
            stride = 2 if (bi == 0 and si > 0) else 1
            plan.append((f"0.1.{si}.{bi}.", stride, prev != out or stride != 1, 1.0 / math.sqrt(expected_var)))
            if bi == 0:
                expected_var = 1.0
            expected_var += c.alpha ** 2
            prev = out
    return plan


def init_params(c: NFResNetConfig, seed: int = 0, prefix: str = "image_prefix.enc.") -> Params:
    """Seeded synthetic weights: N(0, 1) kernels (the standardisation makes their scale irrelevant -- a non-zero mean is added so
    that the mean subtraction is visible), gains around 1 (conv3 included: timm zero-initialises conv3's gain, which would
    silence every residual branch), small biases."""
    g = torch.Generator().manual_seed(seed)
    p: Params = {}
    for name, cin, cout, k in conv_specs(c):
        p[prefix + name + ".weight"] = torch.randn(cout, cin, k, k, generator=g) + 0.3 * torch.randn(cout, 1, 1, 1, generator=g)
        p[prefix + name + ".bias"] = 0.1 * torch.randn(cout, generator=g)
        p[prefix + name + ".gain"] = 1.0 + 0.2 * torch.randn(cout, 1, 1, 1, generator=g)
    return p


def standardized_weight(w: torch.Tensor, gain: torch.Tensor, eps: float) -> torch.Tensor:
    """timm ScaledStdConv2d.forward: F.batch_norm(weight.reshape(1, Cout, -1), training=True, weight=gain * scale, eps)."""
    cout = w.shape[0]
    fan_in = w[0].numel()
    flat = w.reshape(cout, -1).float()
    mean = flat.mean(1, keepdim=True)
    var = flat.var(1, unbiased=False, keepdim=True)
    scale = RELU_GAMMA * fan_in ** -0.5
    return ((flat - mean) * torch.rsqrt(var + eps) * (gain.reshape(cout, 1).float() * scale)).reshape(w.shape).to(w.dtype)


def _conv(p: Params, c: NFResNetConfig, key: str, x: torch.Tensor, stride: int = 1) -> torch.Tensor:
    w = p[key + ".weight"]
    return F.conv2d(x, standardized_weight(w, p[key + ".gain"], c.eps), p[key + ".bias"], stride=stride, padding=w.shape[-1] // 2)


def encoder_fwd(p: Params, c: NFResNetConfig, x: torch.Tensor, prefix: str = "image_prefix.enc.") -> torch.Tensor:
    """(B, 3, H, W) -> (B, 2048): stem, 16 NormFreeBlocks, ReLU, global average pool."""
    y = _conv(p, c, prefix + "0.0.conv", x, stride=2)
    y = F.max_pool2d(y, 3, stride=2, padding=1)
    for pre, stride, down, beta in block_plan(c):
        out = F.relu(y) * beta
        shortcut = y
        if down:
            s = F.avg_pool2d(out, 2, stride, ceil_mode=True, count_include_pad=False) if stride > 1 else out
            shortcut = _conv(p, c, prefix + pre + "downsample.conv", s)
        out = _conv(p, c, prefix + pre + "conv1", out)
        out = _conv(p, c, prefix + pre + "conv2", F.relu(out), stride=stride)
        out = _conv(p, c, prefix + pre + "conv3", F.relu(out))
        y = out * c.alpha + shortcut
    return F.relu(y).mean(dim=(2, 3))
