"""``Image`` container -- the boundary type of Jpeg.compress (src/image/image.py:26-149).

float32 H x W x C in [0, 1] plus the original shape and file extension.  File I/O (out of the hot path)
goes through Pillow because imageio is not part of this image.
"""
import os
from typing import Optional, Tuple, Type

import numpy as np


class Image:
    def __init__(self, data: np.ndarray, shape: Tuple[int, ...], extension: Optional[str]) -> None:
        self.data = data
        self.original_shape = shape
        self.extension = extension

    @classmethod
    def from_array(cls: Type["Image"], data: np.ndarray, shape: Optional[Tuple[int, ...]] = None,
                   extension: Optional[str] = None) -> "Image":
        if shape is None:
            shape = data.shape
        img = cls(data, shape, extension)
        img.reshape(shape)
        return img

    @classmethod
    def load(cls: Type["Image"], path: str) -> "Image":
        from PIL import Image as PILImage
        extension = os.path.splitext(path)[1]
        with PILImage.open(path) as im:
            if im.mode not in ("L", "RGB", "RGBA"):
                im = im.convert("RGB")
            img = np.asarray(im).astype(np.float32) / 255.0
        if img.ndim == 2:
            img = np.stack((img,) * 3, axis=-1)
        elif img.ndim == 3 and img.shape[2] == 3:
            pass
        elif img.ndim == 3 and img.shape[2] == 4:
            img = img[:, :, :3]
        else:
            raise ValueError(f"Unsupported image format: {img.shape}")
        return cls(img, img.shape, extension)

    def copy(self) -> "Image":
        return Image.from_array(self.data.copy(), self.original_shape, self.extension)

    def save(self, path: str) -> None:
        from PIL import Image as PILImage
        PILImage.fromarray((self.data * 255).astype(np.uint8)).save(path)

    def get_flattened(self) -> np.ndarray:
        return self.data.reshape(-1, self.original_shape[-1])

    def get_uint8(self) -> np.ndarray:
        return (self.data * 255).astype(np.uint8)

    def reshape(self, shape: Tuple[int, ...]) -> "Image":
        self.data = self.data.reshape(shape)
        return self

    def __str__(self) -> str:
        return self.data.__str__()
