"""Drop-in alias for the reference's top-level ``image`` package (src/image/__init__.py:20-23)."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from adaptive_edge_aware_jpeg_amd import EvaluationMetrics, Image  # noqa: E402

__all__ = ["EvaluationMetrics", "Image"]
