"""``ImageInput`` -- the image half of ``Magma.preprocess_inputs``' input list
(reference magma/image_input.py:6-23: constructor takes a local path or an
http(s) URL; ``get_image()`` / ``get_transformed_image(transform_fn)``).

Here the image is decoded lazily (the first time it is asked for) and local
paths may also be ``pathlib.Path`` objects or already-open PIL images, which
the synthetic-data tools use; URL fetching needs network access and raises a
descriptive error otherwise."""
from __future__ import annotations

import io
from pathlib import Path
from typing import Callable, Optional, Union

from PIL import Image

_URL_PREFIXES = ("http://", "https://")


class ImageInput:
    def __init__(self, path_or_url: Union[str, Path, Image.Image]):
        self.path_or_url = path_or_url
        self._image: Optional[Image.Image] = path_or_url if isinstance(path_or_url, Image.Image) else None

    # reference attribute name, kept as a property so callers reading `.pil_image` still work
    @property
    def pil_image(self) -> Image.Image:
        if self._image is None:
            self._image = self._decode()
        return self._image

    def _decode(self) -> Image.Image:
        src = self.path_or_url
        if isinstance(src, str) and src.startswith(_URL_PREFIXES):
            try:
                import requests
                payload = requests.get(src, timeout=30).content
            except Exception as exc:  # noqa: BLE001
                raise RuntimeError(f"Could not retrieve image from url:\n{src}") from exc
            return Image.open(io.BytesIO(payload))
        return Image.open(Path(src))

    def get_image(self) -> Image.Image:
        return self.pil_image

    def get_transformed_image(self, transform_fn: Callable):
        return transform_fn(self.pil_image)
