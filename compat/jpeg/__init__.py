"""Drop-in alias for the reference's top-level ``jpeg`` package (setup.py:26-27, src/jpeg/__init__.py:20-22)."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from adaptive_edge_aware_jpeg_amd import Jpeg, JpegCompressionSettings  # noqa: E402
from adaptive_edge_aware_jpeg_amd import edge_detection, quadtree  # noqa: E402,F401

sys.modules[__name__ + ".edge_detection"] = edge_detection
sys.modules[__name__ + ".quadtree"] = quadtree
__all__ = ["Jpeg", "JpegCompressionSettings"]
