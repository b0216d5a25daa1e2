"""Image-caption data with the output contract of the reference's ImgCptDataset +
collate_fn (reference magma/datasets/dataset.py:92-160): every item is
(image (1,3,H,W) float, caption (1,seq_len) int64 right-padded with EOS); a batch is
((B,3,H,W), (B,seq_len)).

``ImgCptDataset`` reads the reference's on-disk layout (``<dir>/image_data/*/*.json`` records
{"image_path": ..., "captions": [...]}, images relative to ``<dir>``); ``SyntheticImgCptDataset``
draws T_i ~ U{8..64} random tokens + EOS padding for the benchmarks.  The dataset
converters (reference convert_datasets.py) are out of scope."""
import json
import random
from pathlib import Path

import torch


class ImgCptDataset(torch.utils.data.Dataset):
    """On-disk image-caption records in the reference's standard format (dataset.py:92-153).

    ``transforms`` maps a PIL image to (1,3,H,W) (Magma.transforms: on the GPU path the resize / crop /
    normalise arithmetic runs as HIP kernels); ``tokenizer.encode(..., max_length=seq_len,
    padding="max_length", truncation=True)`` yields the (1,seq_len) caption.  One caption of the record is drawn
    at random per access; an unreadable image is replaced by another random item, as in the reference."""

    def __init__(self, data_dir, tokenizer, transforms, seq_len: int = 2048, load_data_in_memory: bool = False):
        self.data_dir = Path(data_dir)
        self.tokenizer, self.transforms, self.seq_len = tokenizer, transforms, seq_len
        self.paths = sorted((self.data_dir / "image_data").glob("*/*.json"))
        if not self.paths:
            raise FileNotFoundError(f"no records under {self.data_dir / 'image_data'}/*/*.json")
        self.records = [self._read(p) for p in self.paths] if load_data_in_memory else None

    @staticmethod
    def _read(path):
        try:
            with open(path) as f:
                return json.load(f)
        except (OSError, ValueError):
            return None

    def __len__(self):
        return len(self.paths)

    def _image_path(self, idx, rec):
        if "image_path" in rec:
            return self.data_dir / rec["image_path"]
        p = self.paths[idx]            # no path in the record: images/<shard>/<name>.jpg next to image_data/<shard>/<name>.json
        return self.data_dir / "images" / p.parent.name / (p.stem + ".jpg")

    def __getitem__(self, idx):
        import PIL.Image
        for _ in range(64):
            rec = self.records[idx] if self.records is not None else self._read(self.paths[idx])
            if rec is not None and rec.get("captions"):
                try:
                    img = PIL.Image.open(self._image_path(idx, rec))
                    image = self.transforms(img)
                    caption = self.tokenizer.encode(random.choice(rec["captions"]), return_tensors="pt",
                                                    max_length=self.seq_len, padding="max_length", truncation=True)
                    return image, caption
                except (PIL.UnidentifiedImageError, OSError, PIL.Image.DecompressionBombError, IndexError):
                    print(f"Warning: Could not load image of record {self.paths[idx]}")
            idx = random.randint(0, len(self) - 1)
        raise RuntimeError(f"no readable image-caption record found under {self.data_dir}")


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


def host_side_view(ds):
    """The same dataset with every ImgCptDataset's transform replaced by its host-only twin (``transforms.host``: same pixels,
    tensors stay on the CPU) -- what DataLoader worker processes iterate: a forked worker must not touch the GPU, and
    pinned-memory batches need CPU tensors.  Subset / ConcatDataset wrappers (random_split, lists of directories) are
    walked; datasets without a device-side transform are returned as they are."""
    import copy
    if isinstance(ds, torch.utils.data.Subset):
        inner = host_side_view(ds.dataset)
        return ds if inner is ds.dataset else torch.utils.data.Subset(inner, ds.indices)
    if isinstance(ds, torch.utils.data.ConcatDataset):
        inner = [host_side_view(d) for d in ds.datasets]
        return ds if all(a is b for a, b in zip(inner, ds.datasets)) else torch.utils.data.ConcatDataset(inner)
    host = getattr(getattr(ds, "transforms", None), "host", None)
    if host is None or host is ds.transforms:
        return ds
    view = copy.copy(ds)
    view.transforms = host
    return view


def on_disk(ds) -> bool:
    """True when ``ds`` (through Subset / ConcatDataset wrappers) reads image files: the case loader workers exist for."""
    if isinstance(ds, torch.utils.data.Subset):
        return on_disk(ds.dataset)
    if isinstance(ds, torch.utils.data.ConcatDataset):
        return any(on_disk(d) for d in ds.datasets)
    return isinstance(ds, ImgCptDataset)


def load_img_cpt_datasets(dataset_dir, tokenizer, transforms, seq_len: int = 2048, synthetic=None):
    """reference train.py:34-42 ``_load_img_cpt_datasets``: a list / tuple of directories -> ConcatDataset of their
    datasets, a str -> ImgCptDataset (a missing directory RAISES -- a typo must not train on noise), anything else ->
    TypeError.  The one extension: the literal string "synthetic" selects ``synthetic()`` (a callable returning a
    SyntheticImgCptDataset), which is what the shipped placeholder configs name."""
    if isinstance(dataset_dir, (list, tuple)):
        return torch.utils.data.ConcatDataset(
            [load_img_cpt_datasets(d, tokenizer, transforms, seq_len, synthetic) for d in dataset_dir])
    if isinstance(dataset_dir, str):
        if dataset_dir == "synthetic":
            if synthetic is None:
                raise ValueError("dataset 'synthetic' requested but no synthetic dataset factory was given")
            return synthetic()
        if not Path(dataset_dir).is_dir():
            raise FileNotFoundError(f"dataset directory {dataset_dir!r} does not exist "
                                    "(use the literal 'synthetic' for random image-caption pairs)")
        return ImgCptDataset(dataset_dir, tokenizer, transforms, seq_len=seq_len)
    raise TypeError("dataset dir wrong type")


def get_pretraining_datasets(config, tokenizer, transforms, seq_len: int = 2048, synthetic_train=None, synthetic_eval=None,
                             split_seed=None):
    """reference train.py:45-66: the train set from ``config.train_dataset_dir`` (str or list); the eval set from
    ``config.eval_dataset_dir``, or -- when that is None -- ``eval_dataset_pct`` of the train set split off at random."""
    train = load_img_cpt_datasets(config.train_dataset_dir, tokenizer, transforms, seq_len, synthetic_train)
    if config.eval_dataset_dir is None:
        eval_len = int(len(train) * config.eval_dataset_pct)
        train_len = len(train) - eval_len
        print(f"Randomly splitting train_dataset into two datasets of length {train_len} and {eval_len}")
        g = None if split_seed is None else torch.Generator().manual_seed(split_seed)   # the same split on every rank
        train, evals = torch.utils.data.random_split(train, [train_len, eval_len], generator=g)
    else:
        evals = load_img_cpt_datasets(config.eval_dataset_dir, tokenizer, transforms, seq_len, synthetic_eval)
    return train, evals


def collate_fn(batch_data, seq_len=2048):
    images, captions = list(zip(*batch_data))
    return torch.cat(images), torch.cat([i[:, :seq_len] for i in captions])


def synthetic_batch(batch, image_size, seq_len, eos, vocab, seed, device=None, dtype=torch.float32):
    ds = SyntheticImgCptDataset(batch, image_size, seq_len, eos, vocab, seed)
    imgs, caps = collate_fn([ds[i] for i in range(batch)], seq_len)
    if device is not None:
        imgs, caps = imgs.to(device=device, dtype=dtype), caps.to(device)
    return imgs, caps
