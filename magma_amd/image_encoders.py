"""Image encoders for the MAGMA path (reference magma/image_encoders.py:48-91).

``clip_resnet_large`` (CLIP RN50x16 trunk with the attention pool replaced by
``b d h w -> b (h w) d``) is what the shipped YAMLs select; ``clip_resnet``
(RN50x4) is the same trunk at another width/depth; ``clip`` (ViT-B/32) is the
VisionTransformer below, feeding the pooled ImagePrefix branch; ``nfresnet50`` is
timm's NF-ResNet-50 (NFResNet50 below, same pooled branch).  The CLIP module trees carry openai/CLIP's
parameter names (conv1..3, bn1..3, layer{1..4}.{j}.{conv,bn}{1..3},
downsample.{0,1}) so reference checkpoints load by name, but the arithmetic is
NOT torch: forward() drives the HIP kernels -- NHWC activations, every conv an
MFMA GEMM (1x1: plain, 3x3: implicit im2col, stem: explicit im2col) with the
eval-mode BatchNorm folded into the epilogue's per-channel scale/shift, ReLU
and the bottleneck's residual add fused as well."""
from __future__ import annotations

from collections import OrderedDict

import torch
import torch.nn as nn

from . import ops

CLIP_RESNETS = {
    # name: (layers, width, input_resolution)
    "clip_resnet_large": ((6, 8, 18, 8), 96, 384),  # RN50x16 (MAGMA_v1 / v2)
    "clip_resnet": ((4, 6, 10, 6), 80, 288),         # RN50x4 (reference image_encoders.py:58-59): same trunk, 2560 channels
}
_ALIASES = {"RN50x16": "clip_resnet_large", "RN50x4": "clip_resnet"}


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, **kw):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False, **kw)
        self.bn1 = nn.BatchNorm2d(planes, **kw)
        self.conv2 = nn.Conv2d(planes, planes, 3, padding=1, bias=False, **kw)
        self.bn2 = nn.BatchNorm2d(planes, **kw)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False, **kw)
        self.bn3 = nn.BatchNorm2d(planes * 4, **kw)
        self.stride = stride
        self.downsample = None
        if stride > 1 or inplanes != planes * 4:
            self.downsample = nn.Sequential(OrderedDict([
                ("-1", nn.AvgPool2d(stride)),
                ("0", nn.Conv2d(inplanes, planes * 4, 1, stride=1, bias=False, **kw)),
                ("1", nn.BatchNorm2d(planes * 4, **kw)),
            ]))


class ModifiedResNetTrunk(nn.Module):
    """CLIP ModifiedResNet without attnpool: (B,3,H,W) -> (B, H/32*W/32, 32*width)."""

    def __init__(self, layers=(6, 8, 18, 8), width=96, input_resolution=384, device=None, dtype=None):
        super().__init__()
        kw = dict(device=device, dtype=dtype)
        self.input_resolution = input_resolution
        self.width = width
        self.conv1 = nn.Conv2d(3, width // 2, 3, stride=2, padding=1, bias=False, **kw)
        self.bn1 = nn.BatchNorm2d(width // 2, **kw)
        self.conv2 = nn.Conv2d(width // 2, width // 2, 3, padding=1, bias=False, **kw)
        self.bn2 = nn.BatchNorm2d(width // 2, **kw)
        self.conv3 = nn.Conv2d(width // 2, width, 3, padding=1, bias=False, **kw)
        self.bn3 = nn.BatchNorm2d(width, **kw)
        self._inplanes = width
        self.layer1 = self._make_layer(width, layers[0], 1, kw)
        self.layer2 = self._make_layer(width * 2, layers[1], 2, kw)
        self.layer3 = self._make_layer(width * 4, layers[2], 2, kw)
        self.layer4 = self._make_layer(width * 8, layers[3], 2, kw)
        self.out_dim = width * 32
        self._packed = None
        self.eval()  # clip.load(...) returns the tower in eval mode

    def _make_layer(self, planes, blocks, stride, kw):
        mods = [Bottleneck(self._inplanes, planes, stride, **kw)]
        self._inplanes = planes * 4
        for _ in range(1, blocks):
            mods.append(Bottleneck(self._inplanes, planes, **kw))
        return nn.Sequential(*mods)

    # ---- weight packing (one-off re-layout; invalidated when params change) ----
    def invalidate_packed(self):
        self._packed = None

    def _pack_conv(self, conv: nn.Conv2d, bn: nn.BatchNorm2d):
        w = conv.weight.detach()
        cout, cin, kh, _ = w.shape
        if kh == 1:
            w2 = w.reshape(cout, cin)
        elif cin == 3:  # stem conv1: explicit im2col columns (c,ky,kx), padded to 32
            w2 = torch.zeros(cout, 32, dtype=w.dtype, device=w.device)
            w2[:, :27] = w.reshape(cout, 27)
        else:
            w2 = w.permute(0, 2, 3, 1).reshape(cout, 9 * cin)
        scale = (bn.weight.detach().float() / torch.sqrt(bn.running_var.float() + bn.eps)).contiguous()
        shift = (bn.bias.detach().float() - bn.running_mean.float() * scale).contiguous()
        return ops.PackedLinear(w2, bias=shift), scale

    def _ensure_packed(self):
        if self._packed is not None:
            return self._packed
        if self.training:
            raise RuntimeError("the packed (BatchNorm-folded) operands of the trunk are inference operands: in training mode the "
                               "tower runs through magma_amd.train_engine (batch-statistics BatchNorm, running-stat updates) -- "
                               "call .eval() for the module-call path (SURVEY Q5)")
        pk = {"conv1": self._pack_conv(self.conv1, self.bn1), "conv2": self._pack_conv(self.conv2, self.bn2),
              "conv3": self._pack_conv(self.conv3, self.bn3)}
        for li in range(1, 5):
            for j, blk in enumerate(getattr(self, f"layer{li}")):
                pre = f"layer{li}.{j}."
                pk[pre + "conv1"] = self._pack_conv(blk.conv1, blk.bn1)
                pk[pre + "conv2"] = self._pack_conv(blk.conv2, blk.bn2)
                pk[pre + "conv3"] = self._pack_conv(blk.conv3, blk.bn3)
                if blk.downsample is not None:
                    pk[pre + "down"] = self._pack_conv(blk.downsample[1], blk.downsample[2])
        self._packed = pk
        return pk

    # ---- forward on the HIP kernels ----
    @staticmethod
    def _conv(a, packed, relu=True, conv=None, residual=None):
        lin, scale = packed
        if residual is None:
            return ops.gemm(a, lin, scale=scale, act=ops.MG_ACT_RELU if relu else ops.MG_ACT_NONE, conv=conv)
        return ops.gemm(a, lin, scale=scale, residuals=(residual,), act_after=ops.MG_ACT_RELU, conv=conv)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        pk = self._ensure_packed()
        if x.ndim != 4 or x.shape[1] != 3 or x.shape[2] % 32 or x.shape[3] % 32:
            raise ValueError(f"expected (B,3,H,W) with H,W multiples of 32, got {tuple(x.shape)}")
        x = x.to(torch.bfloat16).contiguous()
        B, _, H, W = x.shape
        h, w = H // 2, W // 2
        y = self._conv(ops.stem_im2col(x), pk["conv1"])                       # [B*h*w, width/2]
        y = self._conv(y, pk["conv2"], conv=(h, w, y.shape[1]))
        y = self._conv(y, pk["conv3"], conv=(h, w, y.shape[1]))
        y = ops.avgpool2(y.view(B, h, w, -1))
        h, w = h // 2, w // 2
        y = y.view(B * h * w, -1)
        for li in range(1, 5):
            for j, blk in enumerate(getattr(self, f"layer{li}")):
                pre = f"layer{li}.{j}."
                identity = y
                out = self._conv(y, pk[pre + "conv1"])
                out = self._conv(out, pk[pre + "conv2"], conv=(h, w, out.shape[1]))
                if blk.stride > 1:
                    out = ops.avgpool2(out.view(B, h, w, -1)).view(B * (h // 2) * (w // 2), -1)
                if blk.downsample is not None:
                    if blk.stride > 1:
                        identity = ops.avgpool2(identity.view(B, h, w, -1)).view(B * (h // 2) * (w // 2), -1)
                    identity = self._conv(identity, pk[pre + "down"], relu=False)
                if blk.stride > 1:
                    h, w = h // 2, w // 2
                y = self._conv(out, pk[pre + "conv3"], residual=identity)
        return y.view(B, h * w, -1)   # NHWC rows == "b (h w) d": the rearrange is free


class _ViTAttention(nn.Module):
    """Parameter container with nn.MultiheadAttention's names (in_proj_weight / in_proj_bias / out_proj.*)."""

    def __init__(self, width, **kw):
        super().__init__()
        self.in_proj_weight = nn.Parameter(torch.empty(3 * width, width, **kw))
        self.in_proj_bias = nn.Parameter(torch.empty(3 * width, **kw))
        self.out_proj = nn.Linear(width, width, **kw)


class _ViTBlock(nn.Module):
    def __init__(self, width, **kw):
        super().__init__()
        self.attn = _ViTAttention(width, **kw)
        self.ln_1 = nn.LayerNorm(width, **kw)
        self.mlp = nn.Sequential(OrderedDict([("c_fc", nn.Linear(width, 4 * width, **kw)), ("gelu", nn.Identity()),
                                              ("c_proj", nn.Linear(4 * width, width, **kw))]))
        self.ln_2 = nn.LayerNorm(width, **kw)


class _ViTTransformer(nn.Module):
    def __init__(self, width, layers, **kw):
        super().__init__()
        self.resblocks = nn.Sequential(*[_ViTBlock(width, **kw) for _ in range(layers)])


class VisionTransformer(nn.Module):
    """CLIP VisionTransformer (encoder_name "clip" = ViT-B/32, reference image_encoders.py:56-63), used whole -- ln_post and
    the 512-d projection included -- so it returns (B, output_dim) and feeds the pooled branch of ImagePrefix.  openai/CLIP
    parameter names; forward() drives the HIP kernels: patchify + GEMM (the stride-32 patch conv), class/positional embedding,
    LayerNorm, QKV / out_proj / MLP GEMMs with bias, QuickGELU and residual adds in the epilogues, short-sequence attention."""

    def __init__(self, input_resolution=224, patch_size=32, width=768, layers=12, heads=12, output_dim=512, device=None, dtype=None):
        super().__init__()
        if width != heads * 64:
            raise ValueError("the short-sequence attention kernel is specialised for head dim 64 (CLIP ViT-B)")
        kw = dict(device=device, dtype=dtype)
        self.input_resolution, self.patch_size, self.width, self.heads, self.out_dim = input_resolution, patch_size, width, heads, output_dim
        scale = width ** -0.5
        self.conv1 = nn.Conv2d(3, width, patch_size, stride=patch_size, bias=False, **kw)
        self.class_embedding = nn.Parameter(scale * torch.randn(width, **kw))
        self.positional_embedding = nn.Parameter(scale * torch.randn((input_resolution // patch_size) ** 2 + 1, width, **kw))
        self.ln_pre = nn.LayerNorm(width, **kw)
        self.transformer = _ViTTransformer(width, layers, **kw)
        self.ln_post = nn.LayerNorm(width, **kw)
        self.proj = nn.Parameter(scale * torch.randn(width, output_dim, **kw))
        with torch.no_grad():
            for blk in self.transformer.resblocks:
                blk.attn.in_proj_weight.normal_(std=scale)
                blk.attn.in_proj_bias.zero_()
        self._packed = None
        self.eval()

    def invalidate_packed(self):
        self._packed = None

    def _ensure_packed(self):
        if self._packed is None:
            f32 = lambda t: t.detach().float().contiguous()  # noqa: E731
            pk = {"conv1": ops.PackedLinear(self.conv1.weight.detach().reshape(self.width, -1)),
                  "proj": ops.PackedLinear(self.proj.detach().t().contiguous()),
                  "ln_pre": (f32(self.ln_pre.weight), f32(self.ln_pre.bias)), "ln_post": (f32(self.ln_post.weight), f32(self.ln_post.bias)),
                  "blocks": []}
            for blk in self.transformer.resblocks:
                pk["blocks"].append({
                    "ln_1": (f32(blk.ln_1.weight), f32(blk.ln_1.bias)), "ln_2": (f32(blk.ln_2.weight), f32(blk.ln_2.bias)),
                    "qkv": ops.PackedLinear(blk.attn.in_proj_weight, blk.attn.in_proj_bias),
                    "out": ops.PackedLinear(blk.attn.out_proj.weight, blk.attn.out_proj.bias),
                    "c_fc": ops.PackedLinear(blk.mlp.c_fc.weight, blk.mlp.c_fc.bias),
                    "c_proj": ops.PackedLinear(blk.mlp.c_proj.weight, blk.mlp.c_proj.bias)})
            self._packed = pk
        return self._packed

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        pk = self._ensure_packed()
        P, R = self.patch_size, self.input_resolution
        if x.ndim != 4 or x.shape[1] != 3 or x.shape[2] != R or x.shape[3] != R:
            raise ValueError(f"expected (B,3,{R},{R}) (fixed positional embedding), got {tuple(x.shape)}")
        x = x.to(torch.bfloat16).contiguous()
        B = x.shape[0]
        patches = ops.gemm(ops.patchify(x, P), pk["conv1"])                                   # [B*G, width]
        t = ops.vit_embed(patches, self.class_embedding.detach(), self.positional_embedding.detach(), B)
        S, w = t.shape[1], self.width
        t = ops.layernorm(t.view(B * S, w), *pk["ln_pre"], self.ln_pre.eps)
        for blk, b in zip(self.transformer.resblocks, pk["blocks"]):
            h = ops.layernorm(t, *b["ln_1"], blk.ln_1.eps)
            ctx = ops.attn_small(ops.gemm(h, b["qkv"]), B, S, self.heads)
            t = ops.gemm(ctx, b["out"], residuals=(t,))
            h = ops.layernorm(t, *b["ln_2"], blk.ln_2.eps)
            h = ops.gemm(h, b["c_fc"], act=ops.MG_ACT_QUICK_GELU)
            t = ops.gemm(h, b["c_proj"], residuals=(t,))
        cls = ops.layernorm(t.view(B, S, w)[:, 0, :], *pk["ln_post"], self.ln_post.eps)       # strided rows: the class tokens
        return ops.gemm(cls, pk["proj"])                                                      # (B, output_dim)


class ScaledStdConv2d(nn.Module):
    """Parameter container with timm ScaledStdConv2d's names: weight [cout,cin,k,k], bias [cout], gain [cout,1,1,1]."""

    def __init__(self, cin, cout, k, **kw):
        super().__init__()
        self.weight = nn.Parameter(torch.randn(cout, cin, k, k, **kw))
        self.bias = nn.Parameter(torch.zeros(cout, **kw))
        self.gain = nn.Parameter(torch.ones(cout, 1, 1, 1, **kw))
        self.k = k


class _NFDownsample(nn.Module):
    def __init__(self, cin, cout, **kw):
        super().__init__()
        self.conv = ScaledStdConv2d(cin, cout, 1, **kw)


class NormFreeBlock(nn.Module):
    def __init__(self, cin, cout, mid, stride, beta, **kw):
        super().__init__()
        self.downsample = _NFDownsample(cin, cout, **kw) if (cin != cout or stride != 1) else None
        self.conv1 = ScaledStdConv2d(cin, mid, 1, **kw)
        self.conv2 = ScaledStdConv2d(mid, mid, 3, **kw)
        self.conv3 = ScaledStdConv2d(mid, cout, 1, **kw)
        self.stride, self.beta = stride, beta


class NFResNet50(nn.Module):
    """timm ``nf_resnet50`` as the reference wraps it (image_encoders.py:31-45):
    ``Sequential(Sequential(stem, stages, final_conv, final_act), AdaptiveAvgPool2d((1,1)))`` -> state-dict keys
    ``0.0.conv.*`` (stem), ``0.1.{stage}.{block}.{conv1,conv2,conv3,downsample.conv}.*`` -- kept, so a reference checkpoint loads
    by name.  (B,3,H,W) -> (B, 2048) for the pooled ImagePrefix branch.

    Published architecture (Brock et al. 2021; timm models/nfnet.py ``_nfres_cfg(depths=(3,4,6,3))``, layers/std_conv.py):
    scaled-weight-standardised convs (no BatchNorm), 7x7/2 stem + 3x3/2 max pool, 16 pre-activation bottlenecks
    ``y = alpha * conv3(relu(conv2(relu(conv1(relu(x) * beta))))) + shortcut`` with alpha 0.2, ReLU, global average pool.
    ``timm`` is not installed here: the arithmetic is checked against the restatement in oracle/nfnet.py (parity unpinned).

    forward() drives the HIP kernels: NHWC bf16 activations; every conv an MFMA GEMM over the standardised weights
    (mg_weight_standardize_bf16, re-done only when the weights change); the stem through an explicit im2col of the NCHW image
    (K = 147 -> 160); beta, alpha, the conv bias and the residual add in the GEMM epilogues; max pool, the stride-2 sampling
    of the three strided 3x3 convs, ReLU and ReLU + global mean as small HBM-bound kernels."""

    ALPHA, EPS, GAMMA = 0.2, 1e-5, 1.7139588594436646
    DEPTHS, CHANNELS, STEM = (3, 4, 6, 3), (256, 512, 1024, 2048), 64

    def __init__(self, input_resolution=256, device=None, dtype=None, depths=None, channels=None, stem_chs=None):
        super().__init__()
        kw = dict(device=device, dtype=dtype)
        depths, channels = tuple(depths or self.DEPTHS), tuple(channels or self.CHANNELS)
        stem_chs = stem_chs or self.STEM
        self.input_resolution = input_resolution
        self.out_dim = channels[-1]
        stem = nn.Sequential(OrderedDict([("conv", ScaledStdConv2d(3, stem_chs, 7, **kw)), ("pool", nn.Identity())]))
        stages, prev, expected_var = [], stem_chs, 1.0
        for si, depth in enumerate(depths):
            blocks = []
            for bi in range(depth):
                stride = 2 if (bi == 0 and si > 0) else 1
                blocks.append(NormFreeBlock(prev, channels[si], channels[si] // 4, stride, expected_var ** -0.5, **kw))
                if bi == 0:
                    expected_var = 1.0
                expected_var += self.ALPHA ** 2
                prev = channels[si]
            stages.append(nn.Sequential(*blocks))
        inner = nn.Sequential(stem, nn.Sequential(*stages), nn.Identity(), nn.Identity())   # stem, stages, final_conv, final_act
        self.add_module("0", inner)
        self.add_module("1", nn.Identity())                                                 # AdaptiveAvgPool2d((1, 1))
        self._packed = None
        self.eval()

    @property
    def stem_conv(self):
        return getattr(self, "0")[0].conv

    @property
    def stages(self):
        return getattr(self, "0")[1]

    def invalidate_packed(self):
        self._packed = None

    def _pack(self, conv: ScaledStdConv2d, scale_mul: float = 1.0, khwc: bool = False, ldo=None, scale_bias: bool = False):
        """(PackedLinear over the standardised weights, per-channel epilogue scale): out = acc * scale_mul + bias, the bias
        multiplied by scale_mul as well when ``scale_bias`` (alpha * (W t + b)); not for beta, which scales the conv INPUT."""
        cout, cin, k, _ = conv.weight.shape
        w = ops.weight_standardize(conv.weight.detach().to(torch.bfloat16).contiguous(), conv.gain.detach().to(torch.bfloat16).reshape(-1),
                                   self.GAMMA * (cin * k * k) ** -0.5, self.EPS, to_khwc=khwc, ldo=ldo)
        bias = conv.bias.detach().float() * (scale_mul if scale_bias else 1.0)
        scale = torch.full((cout,), scale_mul, dtype=torch.float32, device=w.device)
        return ops.PackedLinear(w, bias=bias), scale

    def _ensure_packed(self):
        if self._packed is None:
            pk = {"stem": self._pack(self.stem_conv, ldo=160)}
            for si, stage in enumerate(self.stages):
                for bi, blk in enumerate(stage):
                    pre = f"{si}.{bi}."
                    pk[pre + "conv1"] = self._pack(blk.conv1, blk.beta)           # conv1(relu(x) * beta) = beta * (W relu(x)) + b
                    pk[pre + "conv2"] = self._pack(blk.conv2, khwc=True)
                    pk[pre + "conv3"] = self._pack(blk.conv3, self.ALPHA, scale_bias=True)     # alpha * (W t + b) + shortcut
                    if blk.downsample is not None:
                        pk[pre + "down"] = self._pack(blk.downsample.conv, blk.beta)
            c = max(self.CHANNELS[-1], self.out_dim)
            pk["one"] = torch.ones(c, dtype=torch.float32, device=self.stem_conv.weight.device)
            pk["zero"] = torch.zeros(c, dtype=torch.float32, device=self.stem_conv.weight.device)
            self._packed = pk
        return self._packed

    @staticmethod
    def _conv(a, packed, relu=False, conv=None, residual=None):
        lin, scale = packed
        # epilogue: act(acc * scale[n] + bias[n]) (+ residual)
        return ops.gemm(a, lin, scale=scale, act=ops.MG_ACT_RELU if relu else ops.MG_ACT_NONE, conv=conv,
                        residuals=() if residual is None else (residual,))

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        pk = self._ensure_packed()
        if x.ndim != 4 or x.shape[1] != 3 or x.shape[2] % 32 or x.shape[3] % 32:
            raise ValueError(f"expected (B,3,H,W) with H,W multiples of 32 (the inputs of the three stride-2 stages must be even: 224 -> 112, 56, 28, 14, 7), got {tuple(x.shape)}")
        x = x.to(torch.bfloat16).contiguous()
        B, _, H, W = x.shape
        h, w = H // 2, W // 2
        y = self._conv(ops.im2col_nchw(x, 7, 2, 3, 160), pk["stem"])                     # [B*h*w, 64], no activation
        y = ops.maxpool3x3s2(y.view(B, h, w, -1))
        h, w = y.shape[1], y.shape[2]
        y = y.view(B * h * w, -1)
        for si, stage in enumerate(self.stages):
            for bi, blk in enumerate(stage):
                pre = f"{si}.{bi}."
                c = y.shape[1]
                r = ops.bn_apply(y, pk["one"][:c], pk["zero"][:c], relu=True)            # relu(x); beta rides in the epilogues
                shortcut = y
                if blk.downsample is not None:
                    s = r
                    if blk.stride > 1:
                        s = ops.avgpool2(r.view(B, h, w, c)).view(B * (h // 2) * (w // 2), c)
                    shortcut = self._conv(s, pk[pre + "down"])
                out = self._conv(r, pk[pre + "conv1"], relu=True)                        # relu feeds conv2
                out = self._conv(out, pk[pre + "conv2"], relu=True, conv=(h, w, out.shape[1]))   # relu feeds conv3
                if blk.stride > 1:     # 3x3, stride 2, padding 1 == the stride-1 map at the even positions (ReLU commutes)
                    out = ops.subsample2(out.view(B, h, w, -1))
                    h, w = out.shape[1], out.shape[2]
                    out = out.view(B * h * w, -1)
                y = self._conv(out, pk[pre + "conv3"], residual=shortcut)
        return ops.relu_mean_rows(y.view(B, h * w, -1))                                   # final_act + global average pool


def clip_encoder(device=None, name: str = "clip_resnet_large", dtype=None) -> nn.Module:
    name = _ALIASES.get(name, name)
    if name in ("clip", "ViT-B/32"):
        return VisionTransformer(224, 32, 768, 12, 12, 512, device=device, dtype=dtype)
    if name in CLIP_RESNETS:
        layers, width, res = CLIP_RESNETS[name]
        return ModifiedResNetTrunk(layers, width, res, device=device, dtype=dtype)
    raise NotImplementedError(f"encoder {name!r} is not implemented on the MI355X path (SURVEY 8f row 4): clip_resnet_large "
                              "(RN50x16), clip_resnet (RN50x4) and clip (ViT-B/32) are")


def nfresnet50(device=None, pretrained: bool = False, dtype=None, input_resolution: int = 256) -> nn.Module:
    """reference image_encoders.py:31-45.  ``pretrained`` would download timm weights: no network here -- weights come from
    the MAGMA checkpoint (Magma.from_checkpoint) like every other tensor."""
    if pretrained:
        raise RuntimeError("pretrained nf_resnet50 weights need timm + network access; load a MAGMA checkpoint instead "
                           "(pretrained_img_encoder: false)")
    return NFResNet50(input_resolution, device=device, dtype=dtype)


def get_image_encoder(name: str, device=None, pretrained: bool = False, dtype=None, image_size: int = 256) -> nn.Module:
    """reference image_encoders.py:78-91."""
    if name == "nfresnet50":
        return nfresnet50(device=device, pretrained=pretrained, dtype=dtype, input_resolution=image_size)
    if "clip" in name:
        return clip_encoder(device=device, name=name, dtype=dtype)
    raise ValueError(f"image encoder {name} not recognized")
