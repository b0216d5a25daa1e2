"""Host-side image preprocessing.  Only the CLIP pipeline of reference
magma/transforms.py:121-134 is on the hot path's entry (clip_preprocess):
bicubic resize of the short side -> center crop -> RGB -> [0,1] tensor ->
batch dim -> CLIP mean/std.  torchvision is not available in this image, so the
same steps are written with PIL + torch (T.Resize(n, BICUBIC) on a PIL image is
PIL's own bicubic resampling, which is what is called here)."""
import functools
import math

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


# ---- Pillow's resampling coefficients (host side of the device path) --------------------------------------------
_PRECISION_BITS = 32 - 8 - 2


def _bicubic_kernel(x, a=-0.5):
    x = -x if x < 0.0 else x
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


@functools.lru_cache(maxsize=64)
def pil_bicubic_tables(in_size: int, out_size: int):
    """(coeffs int32 [out, ksize], bounds int32 [out, 2]) of Pillow's ImagingResample for BICUBIC, 8 bits per
    channel: precompute_coeffs + normalize_coeffs_8bpc (Pillow src/libImaging/Resample.c), doubles, same
    expression order -- the integer tables are identical to Pillow's, so the device passes are bit-exact."""
    scale = filterscale = float(in_size) / out_size
    if filterscale < 1.0:
        filterscale = 1.0
    support = 2.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    kk = np.zeros((out_size, ksize), dtype=np.int32)
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    inv = 1.0 / filterscale
    one = float(1 << _PRECISION_BITS)
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        lo = max(int(center - support + 0.5), 0)
        n = min(int(center + support + 0.5), in_size) - lo
        w = [_bicubic_kernel((i + lo - center + 0.5) * inv) for i in range(n)]
        total = 0.0
        for v in w:
            total += v
        for i, v in enumerate(w):
            if total != 0.0:
                v = v / total
            kk[xx, i] = int(-0.5 + v * one) if v < 0 else int(0.5 + v * one)
        bounds[xx] = (lo, n)
    return kk, bounds


def _device_pipeline(image, n_px, device):
    """RGB PIL image -> [1, 3, n, n] fp32 on `device`: upload the raw pixels once, resize / crop / normalise
    with the HIP kernels (bit-identical to the host path below)."""
    from . import ops
    w, h = image.size
    nw, nh = (n_px, int(n_px * h / w)) if w <= h else (int(n_px * w / h), n_px)
    img = torch.from_numpy(np.array(image, dtype=np.uint8)).to(device)
    if nw != w:      # Pillow: horizontal pass first, uint8 intermediate
        kx, bx = pil_bicubic_tables(w, nw)
        img = ops.resample_u8(img, nw, 1, torch.from_numpy(kx).to(device), torch.from_numpy(bx).to(device))
    if nh != h:
        ky, by = pil_bicubic_tables(h, nh)
        img = ops.resample_u8(img, nh, 0, torch.from_numpy(ky).to(device), torch.from_numpy(by).to(device))
    left, top = int(round((nw - n_px) / 2.0)), int(round((nh - n_px) / 2.0))
    return ops.crop_normalize(img, top, left, n_px, CLIP_MEAN, CLIP_STD).unsqueeze(0)


class ClipPreprocess:
    """The callable clip_preprocess returns (reference transforms.py:87-134).  A class, not a closure: DataLoader worker
    processes are SPAWNED (train_engine.deepspeed_io), so the dataset and its transform travel by pickle."""

    def __init__(self, n_px, use_pad=False, device=None):
        self.n_px, self.use_pad, self.device = n_px, use_pad, device
        self.on_gpu = device is not None and torch.device(device).type == "cuda"
        # the same transform with the tensors left on the host (bit-identical pixels): what a DataLoader worker process runs
        self.host = ClipPreprocess(n_px, use_pad, device=None) if self.on_gpu else self

    def pad_img(self, im):
        """reference transforms.py:87-108: long side -> n_px (LANCZOS, PIL's old ANTIALIAS), pasted centred on black."""
        n_px = self.n_px
        old = im.size
        ratio = float(n_px) / max(old)
        new = tuple(int(x * ratio) for x in old)
        im = im.resize(new, PilImage.LANCZOS)
        canvas = PilImage.new("RGB", (n_px, n_px))
        canvas.paste(im, ((n_px - new[0]) // 2, (n_px - new[1]) // 2))
        return canvas

    def __call__(self, image):
        n_px = self.n_px
        if self.on_gpu and not self.use_pad and image.mode == "RGB" and min(image.size) >= 2:
            return _device_pipeline(image, n_px, self.device)
        image = _resize_short_side(image, n_px)
        image = (self.pad_img(image) if self.use_pad else _center_crop(image, n_px)).convert("RGB")
        t = torch.from_numpy(np.asarray(image, dtype=np.uint8).copy()).permute(2, 0, 1).float() / 255.0
        mean = torch.tensor(CLIP_MEAN).view(3, 1, 1)
        std = torch.tensor(CLIP_STD).view(3, 1, 1)
        out = maybe_add_batch_dim((t - mean) / std)
        # one transform -> one device: grey / palette / alpha images take the PIL branch but must land where the RGB
        # ones do, or collate_fn's torch.cat over a mixed batch raises (reference dataset.py:155-160)
        return out.to(self.device) if self.on_gpu else out


def clip_preprocess(n_px, use_pad=False, device=None):
    """``device``: a GPU -> RGB images are resized / cropped / normalised on it (same bits as the host path);
    other modes (palette, alpha, grey) and ``device=None`` use PIL on the host like the reference."""
    return ClipPreprocess(n_px, use_pad, device)


def pad_to_size(x, size=256):
    """reference transforms.py:8-18 (PIL ImageOps.expand by the missing margin; a larger image is cropped by the negative one)."""
    from PIL import ImageOps
    delta_w, delta_h = size - x.size[0], size - x.size[1]
    padding = (delta_w // 2, delta_h // 2, delta_w - (delta_w // 2), delta_h - (delta_h // 2))
    return ImageOps.expand(x, padding)


def _random_crop(img, size):
    """torchvision T.RandomCrop(size) on a PIL image: offsets from torch.randint, as torchvision draws them."""
    w, h = img.size
    th = tw = size
    if w == tw and h == th:
        return img
    i = int(torch.randint(0, h - th + 1, size=(1,)).item())
    j = int(torch.randint(0, w - tw + 1, size=(1,)).item())
    return img.crop((j, i, j + tw, i + th))


def _resize_bilinear_short_side(img, size):
    """torchvision T.Resize(size) on a PIL image: short side -> size, BILINEAR (PIL's antialiased resampler)."""
    w, h = img.size
    if (w <= h and w == size) or (h <= w and h == size):
        return img
    nw, nh = (size, int(size * h / w)) if w <= h else (int(size * w / h), size)
    return img.resize((nw, nh), PilImage.BILINEAR)


class RandCropResize:
    """reference transforms.py:42-62: pad to the target size, random square crop, random resize to [9/8, 12/8] x target, random
    crop to the target -- the augmentation of the non-CLIP encoders (nfresnet50).  Host-side (PIL), as in the reference."""

    def __init__(self, target_size):
        self.target_size = target_size

    def __call__(self, img):
        import random
        img = pad_to_size(img, self.target_size)
        d_min = min(img.size)
        img = _random_crop(img, d_min)
        t_min = min(d_min, round(9 / 8 * self.target_size))
        t_max = min(d_min, round(12 / 8 * self.target_size))
        t = random.randint(t_min, t_max + 1)
        img = _resize_bilinear_short_side(img, t)
        if min(img.size) < 256:
            img = _resize_bilinear_short_side(img, 256)
        return _random_crop(img, self.target_size)


class ColorJitter:
    """torchvision T.ColorJitter(brightness, contrast, saturation, hue) on a PIL image (reference transforms.py:78-80 uses
    (0.1, 0.1, 0.1, 0.05)), restated from torchvision's published algorithm (un-vendored, absent here): the four operations in
    a random order (torch.randperm), factors uniform in [max(0, 1 - v), 1 + v] / [-hue, hue] (torch's RNG, as torchvision
    draws them); brightness / contrast / saturation through PIL.ImageEnhance (what torchvision's PIL backend calls), hue as a
    cyclic shift of the H channel of the HSV image by uint8(hue * 255)."""

    def __init__(self, brightness=0.0, contrast=0.0, saturation=0.0, hue=0.0):
        self.b = (max(0.0, 1 - brightness), 1 + brightness) if brightness else None
        self.c = (max(0.0, 1 - contrast), 1 + contrast) if contrast else None
        self.s = (max(0.0, 1 - saturation), 1 + saturation) if saturation else None
        self.h = (-hue, hue) if hue else None

    @staticmethod
    def _adjust_hue(img, f):
        if img.mode in ("L", "1", "I", "F"):
            return img
        h, s, v = img.convert("HSV").split()
        nh = np.asarray(h, dtype=np.uint8)
        with np.errstate(over="ignore"):
            nh = nh + np.uint8(int(f * 255) % 256)           # uint8 wrap-around = cyclic shift of the hue
        return PilImage.merge("HSV", (PilImage.fromarray(nh, "L"), s, v)).convert(img.mode)

    def __call__(self, img):
        from PIL import ImageEnhance
        order = torch.randperm(4)
        draw = lambda r: None if r is None else float(torch.empty(1).uniform_(r[0], r[1]))  # noqa: E731
        fb, fc, fs, fh = draw(self.b), draw(self.c), draw(self.s), draw(self.h)
        for k in order.tolist():
            if k == 0 and fb is not None:
                img = ImageEnhance.Brightness(img).enhance(fb)
            elif k == 1 and fc is not None:
                img = ImageEnhance.Contrast(img).enhance(fc)
            elif k == 2 and fs is not None:
                img = ImageEnhance.Color(img).enhance(fs)
            elif k == 3 and fh is not None:
                img = self._adjust_hue(img, fh)
        return img


class BaseTransforms:
    """reference transforms.py:71-84: RGB -> RandCropResize -> RandomHorizontalFlip(0.5) [-> ColorJitter(0.1, 0.1, 0.1, 0.05)
    with use_extra_transforms] -> ToTensor -> batch dim.  A picklable class for the same reason as ClipPreprocess."""

    def __init__(self, image_size, use_extra_transforms=False, device=None):
        self.jitter = ColorJitter(0.1, 0.1, 0.1, 0.05) if use_extra_transforms else None
        self.crop = RandCropResize(image_size)
        self.device = device
        self.on_gpu = device is not None and torch.device(device).type == "cuda"
        self.host = BaseTransforms(image_size, use_extra_transforms, device=None) if self.on_gpu else self

    def __call__(self, img):
        img = img.convert("RGB") if img.mode != "RGB" else img
        img = self.crop(img)
        if float(torch.rand(1)) < 0.5:
            img = img.transpose(PilImage.FLIP_LEFT_RIGHT)
        if self.jitter is not None:
            img = self.jitter(img)
        t = torch.from_numpy(np.asarray(img, dtype=np.uint8).copy()).permute(2, 0, 1).float() / 255.0
        t = maybe_add_batch_dim(t)
        return t.to(self.device) if self.on_gpu else t


def base_transforms(image_size, use_extra_transforms=False, device=None):
    return BaseTransforms(image_size, use_extra_transforms, device)


def get_transforms(image_size, encoder_name, input_resolution=None, use_extra_transforms=False, device=None):
    """reference magma/transforms.py:65-84: the CLIP encoders take clip_preprocess at their input resolution; every other
    encoder (nfresnet50) the random crop / resize / flip pipeline at ``image_size``."""
    if "clip" in encoder_name:
        assert input_resolution is not None
        return clip_preprocess(input_resolution, device=device)
    return base_transforms(image_size, use_extra_transforms, device=device)
