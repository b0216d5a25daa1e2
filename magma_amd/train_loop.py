"""train_step / eval_step / inference_step -- reference magma/train_loop.py:7-21,
48-60,85-98 on the MI355X engine.  Classification variants are out of scope
(SURVEY 2.1 row 7).  ``inference_step`` is routed to ``generate`` (the reference's
version passes an ``inference=True`` kwarg that Magma.forward does not have, Q7)."""
import torch

from .utils import reduce_losses


def _to_device(images, captions, device):
    return images.to(device=device, dtype=torch.bfloat16, non_blocking=True), captions.to(device, non_blocking=True)


def train_step(config, train_loader, model_engine):
    losses = []
    for _ in range(config.gradient_accumulation_steps):
        images, captions = next(train_loader)
        images, captions = _to_device(images, captions, model_engine.device)
        if config.run_blind:
            images = torch.zeros_like(images)
        outputs = model_engine(images, captions)
        loss = outputs.loss
        losses.append(loss)
        model_engine.backward(loss)
        model_engine.step()
    return reduce_losses(torch.mean(torch.stack(losses))).item()


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
