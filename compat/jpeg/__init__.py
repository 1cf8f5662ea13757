"""Drop-in alias for the reference's top-level ``jpeg`` package (setup.py:26-27, src/jpeg/__init__.py:20-22)."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from adaptive_edge_aware_jpeg_amd import Jpeg, JpegCompressionSettings  # noqa: E402
from adaptive_edge_aware_jpeg_amd import edge_detection, quadtree, utils  # noqa: E402,F401
from adaptive_edge_aware_jpeg_amd import jpeg as _jpeg_module  # noqa: E402

# the reference's sub-modules (src/jpeg/{jpeg,edge_detection,quadtree,utils}.py) under their own names
sys.modules[__name__ + ".jpeg"] = _jpeg_module
sys.modules[__name__ + ".edge_detection"] = edge_detection
sys.modules[__name__ + ".quadtree"] = quadtree
sys.modules[__name__ + ".utils"] = utils
__all__ = ["Jpeg", "JpegCompressionSettings"]
