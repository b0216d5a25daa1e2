"""Device image preprocessing (mg_resample_u8 + mg_crop_normalize_f32) against PIL / the oracle: integer work,
so the resized uint8 image and the final fp32 tensor must be BIT-EXACT (reference transforms.py:121-134)."""
import numpy as np
import PIL.Image as PilImage
import pytest
import torch

pytestmark = pytest.mark.gpu

GEOMS = [(480, 640, 384), (300, 200, 224), (224, 224, 384), (1000, 750, 384), (97, 131, 384), (384, 384, 384),
         (500, 333, 224), (64, 64, 224), (2, 3, 224), (1536, 2048, 384)]


@pytest.mark.parametrize("H,W,n_px", GEOMS)
def test_device_preprocess_bit_exact(dev, H, W, n_px):
    from magma_amd import ops
    from magma_amd.transforms import clip_preprocess, pil_bicubic_tables
    from oracle.preprocess import clip_preprocess_u8
    rng = np.random.default_rng(H * 1000 + W)
    img = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    pil = PilImage.fromarray(img)
    # the resize alone, against PIL
    nw, nh = (n_px, int(n_px * H / W)) if W <= H else (int(n_px * W / H), n_px)
    cur = torch.from_numpy(img).to(dev)
    if nw != W:
        kx, bx = pil_bicubic_tables(W, nw)
        cur = ops.resample_u8(cur, nw, 1, torch.from_numpy(kx).to(dev), torch.from_numpy(bx).to(dev))
    if nh != H:
        ky, by = pil_bicubic_tables(H, nh)
        cur = ops.resample_u8(cur, nh, 0, torch.from_numpy(ky).to(dev), torch.from_numpy(by).to(dev))
    assert np.array_equal(cur.cpu().numpy(), np.asarray(pil.resize((nw, nh), PilImage.BICUBIC)))
    # the whole transform: device path == host (PIL + torch) path == oracle, bit for bit
    got = clip_preprocess(n_px, device=dev)(pil)
    assert got.is_cuda and got.shape == (1, 3, n_px, n_px) and got.dtype == torch.float32
    host = clip_preprocess(n_px)(pil)
    assert torch.equal(got.cpu(), host)
    assert np.array_equal(got[0].cpu().numpy(), clip_preprocess_u8(img, n_px))


def test_non_rgb_images_take_the_host_path(dev):
    """Grey / palette / alpha images are resized IN THEIR OWN MODE by PIL and converted afterwards (reference
    transforms.py:121-134), so they take the host arithmetic -- but land on the transform's device like the RGB ones, or a
    mixed batch could not be collated (reference dataset.py:155-160)."""
    from magma_amd.datasets import collate_fn
    from magma_amd.transforms import clip_preprocess
    rng = np.random.default_rng(1)
    grey = PilImage.fromarray(rng.integers(0, 256, (50, 70), dtype=np.uint8), mode="L")
    rgb = PilImage.fromarray(rng.integers(0, 256, (50, 70, 3), dtype=np.uint8), mode="RGB")
    tf = clip_preprocess(224, device=dev)
    out = tf(grey)
    assert out.is_cuda and out.shape == (1, 3, 224, 224)
    assert torch.equal(out.cpu(), clip_preprocess(224)(grey))
    cap = torch.zeros(1, 8, dtype=torch.int64)
    images, _ = collate_fn([(tf(rgb), cap), (out, cap)], seq_len=8)
    assert images.is_cuda and images.shape == (2, 3, 224, 224)


def test_errors_are_loud(dev):
    from magma_amd import ops
    from magma_amd.lib import MagmaHipError
    img = torch.zeros(8, 8, 3, dtype=torch.uint8, device=dev)
    with pytest.raises(MagmaHipError):
        ops.crop_normalize(img, 4, 4, 8, (0, 0, 0), (1, 1, 1))      # window outside the image
