"""clip_preprocess: device path (upload + 2 resample passes + crop/normalise) vs the PIL host path, same images."""
import json, os, sys, time
import numpy as np
import PIL.Image as PilImage
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from magma_amd.transforms import clip_preprocess

dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
out = []
for (H, W) in [(480, 640), (1080, 1920), (3000, 4000)]:
    pil = PilImage.fromarray(rng.integers(0, 256, (H, W, 3), dtype=np.uint8))
    host, devf = clip_preprocess(384), clip_preprocess(384, device=dev)
    assert torch.equal(devf(pil).cpu(), host(pil))
    n = 20
    t0 = time.perf_counter()
    for _ in range(n): host(pil)
    th = (time.perf_counter() - t0) / n
    devf(pil); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): r = devf(pil)
    torch.cuda.synchronize()
    td = (time.perf_counter() - t0) / n
    # kernels alone (pixels already on the device)
    from magma_amd import ops
    from magma_amd.transforms import pil_bicubic_tables
    img = torch.from_numpy(np.asarray(pil)).to(dev)
    nw, nh = int(384 * W / H), 384
    kx, bx = (torch.from_numpy(a).to(dev) for a in pil_bicubic_tables(W, nw))
    ky, by = (torch.from_numpy(a).to(dev) for a in pil_bicubic_tables(H, nh))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n):
        a = ops.resample_u8(img, nw, 1, kx, bx); b = ops.resample_u8(a, nh, 0, ky, by)
        ops.crop_normalize(b, 0, (nw - 384) // 2, 384, (0.5, 0.5, 0.5), (0.25, 0.25, 0.25))
    e1.record(); torch.cuda.synchronize()
    tk = e0.elapsed_time(e1) / n
    out.append({"image": f"{H}x{W}", "host_pil_ms": th * 1e3, "device_end_to_end_ms": td * 1e3, "device_kernels_ms": tk,
                "source_GBps_kernels": H * W * 3 / (tk * 1e-3) / 1e9})
print(json.dumps({"clip_preprocess_384": out, "host_threads": torch.get_num_threads()}))
