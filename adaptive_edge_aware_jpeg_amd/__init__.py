"""adaptive_edge_aware_jpeg_amd -- MI355X (gfx950) implementation of the adaptive edge-aware JPEG ENCODE hot path.

Same Python surface as the reference for this path (fevzibabaoglu/adaptive-edge-aware-jpeg):
``Jpeg``, ``JpegCompressionSettings``, ``Image``, ``EvaluationMetrics``, ``EdgeDetection``, ``QuadTree``, ``convert``,
``apply_normalization``, ``get_color_spaces``.  The arithmetic runs in hand-written HIP kernels behind the
C ABI of ``libaejpeg_hip.so`` (include/aej.h); there is no CPU fallback.
"""
import os as _os

# HIP maps streams onto at most GPU_MAX_HW_QUEUES hardware queues (default 4); two streams on one queue run one after the other.  The
# library overlaps sub-batches and calls on several streams, so ask for a queue per stream -- effective when this
# package is imported before the process makes its first HIP call (the runtime reads the variable when it initialises; the library
# reads the same variable and falls back to a two-stream schedule when it is absent or smaller than 8).
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

from .color import apply_normalization, convert, get_color_spaces  # noqa: E402
from .edge_detection import EdgeDetection  # noqa: E402
from .evaluation_metrics import EvaluationMetrics  # noqa: E402
from .image import Image  # noqa: E402
from .jpeg import EncodedBatch, Jpeg  # noqa: E402
from .quadtree import QuadNode, QuadTree  # noqa: E402
from .settings import JpegCompressionSettings  # noqa: E402

__all__ = ["Jpeg", "JpegCompressionSettings", "EncodedBatch", "Image", "EvaluationMetrics", "EdgeDetection", "QuadTree", "QuadNode",
           "convert", "apply_normalization", "get_color_spaces"]
