"""Drop-in alias for the reference's top-level ``image`` package (src/image/__init__.py:20-23); the quality
metrics (EvaluationMetrics) are evaluation-only and out of scope."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from adaptive_edge_aware_jpeg_amd import Image  # noqa: E402

__all__ = ["Image"]
