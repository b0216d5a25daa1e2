"""ImageInput -- same surface as reference magma/image_input.py:6-23 (local path
or URL -> PIL image -> transform)."""
from io import BytesIO

import PIL.Image as PilImage


class ImageInput:
    """Wrapper to handle image inputs both from local paths and urls."""

    def __init__(self, path_or_url):
        self.path_or_url = path_or_url
        if str(self.path_or_url).startswith(("http://", "https://")):
            try:
                import requests
                response = requests.get(path_or_url)
                self.pil_image = PilImage.open(BytesIO(response.content))
            except Exception as e:  # noqa: BLE001
                raise Exception(f"Could not retrieve image from url:\n{self.path_or_url}") from e
        else:
            self.pil_image = PilImage.open(path_or_url)

    def get_image(self):
        return self.pil_image

    def get_transformed_image(self, transform_fn):
        return transform_fn(self.pil_image)
