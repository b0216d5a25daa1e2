"""Synthetic image-caption data with the output contract of the reference's
ImgCptDataset + collate_fn (reference magma/datasets/dataset.py:133-143,155-160):
images (B,3,H,W) float, captions (B,seq_len) int64 = T_i ~ U{8..64} random tokens
followed by EOS padding to seq_len.  Real dataset readers are out
of scope."""
import torch


class SyntheticImgCptDataset(torch.utils.data.Dataset):
    def __init__(self, n: int, image_size: int = 384, seq_len: int = 2048, eos: int = 50256, vocab: int = 50256,
                 seed: int = 1234, min_len: int = 8, max_len: int = 64):
        self.n, self.image_size, self.seq_len, self.eos, self.vocab = n, image_size, seq_len, eos, vocab
        self.seed, self.min_len, self.max_len = seed, min_len, max_len

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        g = torch.Generator().manual_seed(self.seed + i)
        img = torch.randn(1, 3, self.image_size, self.image_size, generator=g)
        t = int(torch.randint(self.min_len, self.max_len + 1, (1,), generator=g))
        cap = torch.full((1, self.seq_len), self.eos, dtype=torch.int64)
        cap[0, :t] = torch.randint(0, self.vocab, (t,), generator=g)
        return img, cap


def collate_fn(batch_data, seq_len=2048):
    images, captions = list(zip(*batch_data))
    return torch.cat(images), torch.cat([i[:, :seq_len] for i in captions])


def synthetic_batch(batch, image_size, seq_len, eos, vocab, seed, device=None, dtype=torch.float32):
    ds = SyntheticImgCptDataset(batch, image_size, seq_len, eos, vocab, seed)
    imgs, caps = collate_fn([ds[i] for i in range(batch)], seq_len)
    if device is not None:
        imgs, caps = imgs.to(device=device, dtype=dtype), caps.to(device)
    return imgs, caps
