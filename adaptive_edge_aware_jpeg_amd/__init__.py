"""adaptive_edge_aware_jpeg_amd -- MI355X (gfx950) implementation of the adaptive edge-aware JPEG ENCODE hot path.

Same Python surface as the reference for this path (fevzibabaoglu/adaptive-edge-aware-jpeg):
``Jpeg``, ``JpegCompressionSettings``, ``Image``, ``EvaluationMetrics``, ``EdgeDetection``, ``QuadTree``, ``convert``,
``apply_normalization``, ``get_color_spaces``.  The arithmetic runs in hand-written HIP kernels behind the
C ABI of ``libaejpeg_hip.so`` (include/aej.h); there is no CPU fallback.
"""
from .color import apply_normalization, convert, get_color_spaces
from .edge_detection import EdgeDetection
from .evaluation_metrics import EvaluationMetrics
from .image import Image
from .jpeg import EncodedBatch, Jpeg
from .quadtree import QuadNode, QuadTree
from .settings import JpegCompressionSettings

__all__ = ["Jpeg", "JpegCompressionSettings", "EncodedBatch", "Image", "EvaluationMetrics", "EdgeDetection", "QuadTree", "QuadNode",
           "convert", "apply_normalization", "get_color_spaces"]
