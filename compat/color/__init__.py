"""Drop-in alias for the reference's top-level ``color`` package (src/color/__init__.py:20-22)."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from adaptive_edge_aware_jpeg_amd import apply_normalization, convert, get_color_spaces  # noqa: E402

__all__ = ["apply_normalization", "convert", "get_color_spaces"]
