"""``JpegCompressionSettings`` -- same constructor, attributes and errors as src/jpeg/jpeg.py:36-174."""
from typing import List, Tuple

import numpy as np

from . import tables


class JpegCompressionSettings:
    """Settings class for JPEG compression parameters."""

    LUMINANCE_QUANTIZATION_MATRIX = tables.LUMINANCE_QUANTIZATION_MATRIX
    CHROMINANCE_QUANTIZATION_MATRIX = tables.CHROMINANCE_QUANTIZATION_MATRIX
    COLOR_SPACE_SETTINGS = {
        name: {
            "downsampling_ratios": ratios,
            "quantization_matrices": [tables.LUMINANCE_QUANTIZATION_MATRIX, tables.CHROMINANCE_QUANTIZATION_MATRIX,
                                      tables.CHROMINANCE_QUANTIZATION_MATRIX],
        }
        for name, ratios in tables.DOWNSAMPLING_RATIOS.items()
    }

    def __init__(self, color_space: str = "YCoCg", quality_range: Tuple[int, int] = (40, 80),
                 block_size_range: Tuple[int, int] = (4, 64)) -> None:
        if color_space not in self.COLOR_SPACE_SETTINGS:
            raise ValueError(f"Unsupported color space: {color_space}")
        self.color_space = color_space
        self.quality_range = quality_range
        self.block_size_range = block_size_range
        cfg = self.COLOR_SPACE_SETTINGS[color_space]
        self.downsampling_ratios: np.ndarray = cfg["downsampling_ratios"]
        self.quantization_matrices: List[np.ndarray] = cfg["quantization_matrices"]
