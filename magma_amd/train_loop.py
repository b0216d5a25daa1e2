"""train_step / eval_step / inference_step -- reference magma/train_loop.py:7-21,
48-60,85-98 on the MI355X engine.  Classification variants are out of scope
(SURVEY 2.1 row 7).  ``inference_step`` is routed to ``generate`` (the reference's
version passes an ``inference=True`` kwarg that Magma.forward does not have, Q7)."""
import torch

from .utils import reduce_losses


def _to_device(images, captions, device):
    return images.to(device=device, dtype=torch.bfloat16, non_blocking=True), captions.to(device, non_blocking=True)


class LazyLoss:
    """The step's mean loss, still on the device.  The reference returns ``.item()`` (train_loop.py:21), a host sync
    per step that is only ever read every ``log_every`` steps (train.py:146-152); this object syncs when -- and only
    when -- it is formatted, compared or converted."""

    def __init__(self, t: torch.Tensor):
        self.tensor = t

    def item(self) -> float:
        return float(self.tensor)

    __float__ = item

    def __format__(self, spec):
        return format(self.item(), spec)

    def __repr__(self):
        return repr(self.item())

    __str__ = __repr__

    def __lt__(self, o):
        return self.item() < float(o)

    def __gt__(self, o):
        return self.item() > float(o)


def train_step(config, train_loader, model_engine):
    losses = []
    for _ in range(config.gradient_accumulation_steps):
        images, captions = next(train_loader)
        host_caps = captions if not captions.is_cuda else None        # loader output: label index plumbing without a sync
        images, captions = _to_device(images, captions, model_engine.device)
        if config.run_blind:
            images = torch.zeros_like(images)
        outputs = model_engine(images, captions, captions_host=host_caps)
        loss = outputs.loss
        losses.append(loss)
        model_engine.backward(loss)
        model_engine.step()
    return LazyLoss(reduce_losses(torch.mean(torch.stack(losses))))


def eval_step(config, eval_loader, model_engine):
    losses = []
    for _ in range(config.eval_steps):
        images, captions = next(eval_loader)
        images, captions = _to_device(images, captions, model_engine.device)
        if config.run_blind:
            images = torch.zeros_like(images)
        losses.append(model_engine(images, captions).loss)
    return reduce_losses(torch.mean(torch.stack(losses))).item()


@torch.no_grad()
def inference_step(config, eval_loader, model_engine, max_steps: int = 15):
    images, _ = next(eval_loader)
    images = images.to(device=model_engine.device, dtype=torch.bfloat16)
    if config.run_blind:
        images = torch.zeros_like(images)
    model = model_engine.module
    width = min(2, images.shape[0])
    emb = model.embed([images[:width]])
    captions = model.generate(emb, max_steps=max_steps)
    caption = "".join(f"Caption {i}: \n{captions[i]}\n" for i in range(width))
    return images[:width], caption
