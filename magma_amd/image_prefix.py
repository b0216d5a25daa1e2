"""ImagePrefix: image encoder -> Linear(enc_dim -> d) -> dropout -> LayerNorm
(reference magma/image_prefix.py:35-109).  Parameter names (enc.*, proj.*, ln.*)
match the reference; the arithmetic runs on the HIP kernels: one MFMA GEMM with
fused bias, then the LayerNorm kernel."""
import torch
import torch.nn as nn

from . import ops
from .image_encoders import get_image_encoder

# reference magma/image_prefix.py:11-21
ENCODER_SEQ_LENS = {"clip_resnet": 49, "clip_resnet_large": 144}
ENCODER_OUT_DIMS = {"nfresnet50": 2048, "clip": 512, "clip_resnet": 2560, "clip_resnet_large": 3072}


class ImagePrefix(nn.Module):
    def __init__(self, config, out_dim: int = 2048, device=None, dtype=None, enc: nn.Module = None):
        super().__init__()
        self.config = config
        self.encoder_type = config.encoder_name
        self.enc = enc if enc is not None else get_image_encoder(
            config.encoder_name, device=device, pretrained=config.pretrained_img_encoder, dtype=dtype,
            image_size=config.image_size)
        self.encoder_out_dim = getattr(self.enc, "out_dim", None) or ENCODER_OUT_DIMS[self.encoder_type]
        self.out_dim = out_dim
        # encoders with a token grid (CLIP ResNets) project every token; pooled encoders (ViT class token) are projected to
        # image_seq_len tokens at once (reference image_prefix.py:60-72)
        self.pooled = self.encoder_type not in ENCODER_SEQ_LENS
        self.out_seq_len = config.image_seq_len if self.pooled else ENCODER_SEQ_LENS[self.encoder_type]
        self.proj = nn.Linear(self.encoder_out_dim, self.out_dim * self.out_seq_len if self.pooled else self.out_dim,
                              device=device, dtype=dtype)
        self.dropout = nn.Dropout(config.image_embed_dropout_prob)
        self.use_layernorm = config.use_image_embed_layernorm
        if self.use_layernorm:
            self.ln = nn.LayerNorm(self.out_dim, device=device, dtype=dtype)
        self._packed = None

    def invalidate_packed(self):
        self._packed = None
        if hasattr(self.enc, "invalidate_packed"):
            self.enc.invalidate_packed()

    def _ensure_packed(self):
        if self._packed is None:
            pk = {"proj": ops.PackedLinear(self.proj.weight, bias=self.proj.bias)}
            if self.use_layernorm:
                pk["ln_g"] = self.ln.weight.detach().float().contiguous()
                pk["ln_b"] = self.ln.bias.detach().float().contiguous()
            self._packed = pk
        return self._packed

    def forward(self, x: torch.Tensor, dropout_mask: torch.Tensor = None) -> torch.Tensor:
        """x (b,c,h,w) -> (b, seq, out_dim) bf16.  ``dropout_mask`` (b,seq,out_dim),
        already scaled by 1/(1-p), is applied in training mode when given."""
        feats = self.enc(x)                                    # (B, P, enc_dim), or (B, enc_dim) from a pooled encoder
        if self.pooled:
            # reference image_prefix.py:85-101: Linear(enc_dim -> s*d), "b (s d) -> b s d" -- the GEMM output row IS the
            # (s, d) block, so the rearrange is a view; dropout mask and LayerNorm then act on [B*s, d] rows
            assert feats.ndim == 2
            pk = self._ensure_packed()
            y = ops.gemm(feats, pk["proj"]).reshape(feats.shape[0] * self.out_seq_len, self.out_dim)
            if self.training and self.dropout.p > 0:
                if dropout_mask is None:
                    keep = 1.0 - self.dropout.p
                    dropout_mask = (torch.rand(y.shape, device=y.device) < keep).to(y.dtype) / keep
                y = ops.mul(y.contiguous(), dropout_mask.reshape(y.shape).to(y.dtype).contiguous())
            if self.use_layernorm:
                y = ops.layernorm(y, pk["ln_g"], pk["ln_b"], self.ln.eps)
            return y.view(feats.shape[0], self.out_seq_len, self.out_dim)
        assert feats.ndim == 3, "clip resnet encoders return (b, hw, d)"
        B, P, E = feats.shape
        pk = self._ensure_packed()
        mask = None
        if self.training and self.dropout.p > 0:      # inverted dropout, applied in the projection's epilogue
            if dropout_mask is None:
                keep = 1.0 - self.dropout.p
                dropout_mask = (torch.rand(B * P, self.out_dim, device=feats.device) < keep).to(feats.dtype) / keep
            mask = dropout_mask.reshape(B * P, self.out_dim).to(feats.dtype).contiguous()
        y = ops.gemm(feats.reshape(B * P, E), pk["proj"], aux=mask,
                     aux_mode=ops.MG_AUX_MUL if mask is not None else ops.MG_AUX_NONE)
        if self.use_layernorm:
            y = ops.layernorm(y, pk["ln_g"], pk["ln_b"], self.ln.eps)
        return y.view(B, P, self.out_dim)
