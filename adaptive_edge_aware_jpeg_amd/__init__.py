"""adaptive_edge_aware_jpeg_amd -- MI355X (gfx950) implementation of the adaptive edge-aware JPEG ENCODE hot path.

Same Python surface as the reference for this path (fevzibabaoglu/adaptive-edge-aware-jpeg):
``Jpeg``, ``JpegCompressionSettings``, ``Image``, ``EvaluationMetrics``, ``EdgeDetection``, ``QuadTree``, ``convert``,
``apply_normalization``, ``get_color_spaces``.  The arithmetic runs in hand-written HIP kernels behind the
C ABI of ``libaejpeg_hip.so`` (include/aej.h); there is no CPU fallback.
"""
from ._lib import _look_at_hw_queues, configure_hw_queues, hw_queues, set_hw_queues

# Streams beyond the HIP runtime's hardware queues share queues and serialise; the library's overlap of sub-batches and calls wants a
# queue per stream.  Importing this package only LOOKS at GPU_MAX_HW_QUEUES (it never writes the environment); `configure_hw_queues()`
# is the explicit opt-in, `hw_queues()` reports what the library schedules for, `set_hw_queues(n)` states it (policy: _lib.py).
_look_at_hw_queues()

from .color import apply_normalization, convert, get_color_spaces  # noqa: E402
from .edge_detection import EdgeDetection  # noqa: E402
from .evaluation_metrics import EvaluationMetrics  # noqa: E402
from .image import Image  # noqa: E402
from .jpeg import EncodedBatch, Jpeg  # noqa: E402
from .quadtree import QuadNode, QuadTree  # noqa: E402
from .settings import JpegCompressionSettings  # noqa: E402

__all__ = ["Jpeg", "JpegCompressionSettings", "EncodedBatch", "Image", "EvaluationMetrics", "EdgeDetection", "QuadTree", "QuadNode",
           "convert", "apply_normalization", "get_color_spaces", "hw_queues", "set_hw_queues", "configure_hw_queues"]
