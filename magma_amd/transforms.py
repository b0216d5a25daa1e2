"""Host-side image preprocessing.  Only the CLIP pipeline of reference
magma/transforms.py:121-134 is on the hot path's entry (clip_preprocess):
bicubic resize of the short side -> center crop -> RGB -> [0,1] tensor ->
batch dim -> CLIP mean/std.  torchvision is not available in this image, so the
same steps are written with PIL + torch (T.Resize(n, BICUBIC) on a PIL image is
PIL's own bicubic resampling, which is what is called here)."""
import numpy as np
import PIL.Image as PilImage
import torch

CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


def maybe_add_batch_dim(t):
    return t.unsqueeze(0) if t.ndim == 3 else t


def _resize_short_side(img, n_px):
    w, h = img.size
    if w <= h:
        nw, nh = n_px, int(n_px * h / w)      # torchvision: long side = int(size * long / short)
    else:
        nw, nh = int(n_px * w / h), n_px
    return img.resize((nw, nh), PilImage.BICUBIC)


def _center_crop(img, n_px):
    w, h = img.size
    left = int(round((w - n_px) / 2.0))
    top = int(round((h - n_px) / 2.0))
    return img.crop((left, top, left + n_px, top + n_px))


def clip_preprocess(n_px, use_pad=False):
    if use_pad:
        raise NotImplementedError("pad mode is not used by any shipped config")
    mean = torch.tensor(CLIP_MEAN).view(3, 1, 1)
    std = torch.tensor(CLIP_STD).view(3, 1, 1)

    def fn(image):
        image = _center_crop(_resize_short_side(image, n_px), n_px).convert("RGB")
        t = torch.from_numpy(np.asarray(image, dtype=np.uint8).copy()).permute(2, 0, 1).float() / 255.0
        return maybe_add_batch_dim((t - mean) / std)

    return fn


def get_transforms(image_size, encoder_name, input_resolution=None, use_extra_transforms=False):
    """reference magma/transforms.py:87-111.  Only the clip branch is in scope
    (both shipped YAMLs use clip_resnet_large)."""
    if "clip" in encoder_name:
        assert input_resolution is not None
        return clip_preprocess(input_resolution)
    raise NotImplementedError(f"transforms for encoder {encoder_name!r} are out of scope (SURVEY 8f row 4)")
