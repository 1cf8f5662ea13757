"""``Image`` -- the boundary type ``Jpeg.compress`` takes and ``Jpeg.decompress`` returns (interface of src/image/image.py:26-149).

A float32 ``H x W x C`` array with values in [0, 1], the shape the pixels had when they arrived, and the file extension they came
from.  The encode path only ever reads ``data`` / ``original_shape`` / ``extension`` and calls ``get_flattened()``; file I/O is beside
the hot path and goes through Pillow (imageio, which the reference uses, is not part of this image).
"""
import os
from typing import Optional, Tuple

import numpy as np

_CHANNELS_KEPT = 3          # the codec works on three colour layers


def _to_rgb_float(pixels: np.ndarray) -> np.ndarray:
    """8-bit pixels as decoded from a file -> float32 RGB in [0, 1]: a true division by 255 (image.py:80), grey replicated to three
    channels, alpha dropped; anything else is refused with the reference's message."""
    scaled = pixels.astype(np.float32) / 255.0
    channels = 1 if scaled.ndim == 2 else scaled.shape[2] if scaled.ndim == 3 else 0
    if channels == 1 and scaled.ndim == 2:
        return np.repeat(scaled[:, :, None], _CHANNELS_KEPT, axis=2)
    if channels in (3, 4):
        return scaled[:, :, :_CHANNELS_KEPT]
    raise ValueError(f"Unsupported image format: {scaled.shape}")


class Image:
    def __init__(self, data: np.ndarray, shape: Tuple[int, ...], extension: Optional[str]) -> None:
        self.data = data
        self.original_shape = shape
        self.extension = extension

    @classmethod
    def from_array(cls, data: np.ndarray, shape: Optional[Tuple[int, ...]] = None, extension: Optional[str] = None) -> "Image":
        """Wrap an array; ``shape`` (default: the array's own) becomes both ``original_shape`` and the shape of ``data``."""
        target = tuple(data.shape) if shape is None else shape
        return cls(data, target, extension).reshape(target)

    @classmethod
    def load(cls, path: str) -> "Image":
        from PIL import Image as PILImage
        with PILImage.open(path) as handle:
            # imageio hands back L / RGB / RGBA files as they are; palette and other modes are expanded to RGB first
            decoded = np.asarray(handle if handle.mode in ("L", "RGB", "RGBA") else handle.convert("RGB"))
        rgb = _to_rgb_float(decoded)
        return cls(rgb, rgb.shape, os.path.splitext(path)[1])

    def copy(self) -> "Image":
        return type(self).from_array(np.array(self.data, copy=True), self.original_shape, self.extension)

    def get_uint8(self) -> np.ndarray:
        """``(data * 255)`` truncated to uint8, as the reference stores and displays it (image.py:121-128)."""
        return (self.data * 255).astype(np.uint8)

    def save(self, path: str) -> None:
        from PIL import Image as PILImage
        PILImage.fromarray(self.get_uint8()).save(path)

    def get_flattened(self) -> np.ndarray:
        """One row per pixel: ``(H * W, channels)`` (image.py:111-118)."""
        return self.data.reshape(-1, self.original_shape[-1])

    def reshape(self, shape: Tuple[int, ...]) -> "Image":
        self.data = self.data.reshape(shape)
        return self

    def __str__(self) -> str:
        return str(self.data)
