"""``EvaluationMetrics`` -- the reference's image-quality scores (src/image/evaluation_metrics.py:31-109) on the GPU.

Same constructor and method names; ``psnr`` / ``ssim`` / ``ms_ssim`` return 0-dim float32 torch tensors as piq does with
``reduction='mean'`` on a batch of one.  ``batch`` scores many pairs in one call (device tensors in, device tensor out):
that is what a parameter sweep (test/analysis/metrics_computation.py:150-190) needs.  ``lpips`` is not offered: its
AlexNet weights are fetched from the network by name (evaluation_metrics.py:34-36) and no copy exists offline.
"""
import ctypes
from typing import Union

import numpy as np

from ._lib import get_context
from .image import Image

PSNR, SSIM, MS_SSIM = 1, 2, 4


def _data(image: Union[Image, np.ndarray]):
    """evaluation_metrics.py:111-141 (_image_to_tensor): Image or ndarray (a torch tensor is accepted too), else TypeError;
    colour (H, W, 3) only."""
    if isinstance(image, Image):
        data = image.data
    elif isinstance(image, np.ndarray) or type(image).__module__.startswith("torch"):
        data = image
    else:
        raise TypeError(f"Expected Image or numpy.ndarray, got {type(image)}")
    if data.ndim != 3 or data.shape[2] != 3:
        raise ValueError(f"Unexpected shape: {tuple(data.shape)}")
    return data


class EvaluationMetrics:
    """A collection of image quality assessment metrics."""

    def __init__(self, original_image: Image, compressed_image: Image, device: int = 0) -> None:
        self.original_image = original_image
        self.compressed_image = compressed_image
        self._device = device
        self._scores = {}

    @staticmethod
    def batch(a, b, which: int = PSNR | SSIM | MS_SSIM, device: int = 0):
        """a, b: float32 [B, H, W, 3] in [0, 1] (torch on the GPU, or numpy -> copied).  Returns a float64 device tensor
        [B, 3] = (psnr, ssim, ms_ssim); columns not requested are NaN."""
        ctx = get_context(device)
        t = ctx.torch
        xa, xb = ctx.to_device(a, t.float32), ctx.to_device(b, t.float32)
        if xa.ndim != 4 or xa.shape[3] != 3 or xa.shape != xb.shape:
            raise ValueError("Input batches must both be [B, H, W, 3].")
        B, H, W, _ = xa.shape
        nbytes = ctx.lib.aej_metrics_workspace_bytes(B, H, W)
        ws = ctx.workspace(nbytes)
        out = ctx.empty((B, 3), t.float64)
        ctx.check(ctx.lib.aej_metrics_batch(ctx.handle, xa.data_ptr(), xb.data_ptr(), B, H, W, which, out.data_ptr(), ws.data_ptr(),
                                            ctypes.c_uint64(nbytes)))
        return out

    def _score(self, which: int, column: int):
        if which not in self._scores:
            a, b = _data(self.original_image), _data(self.compressed_image)
            out = EvaluationMetrics.batch(a[None], b[None], which, self._device)
            self._scores[which] = out[0, column].float().cpu()
        return self._scores[which]

    def psnr(self):
        """Peak Signal-to-Noise Ratio: piq.psnr(x, y, data_range=1.0) (evaluation_metrics.py:50-61)."""
        return self._score(PSNR, 0)

    def ssim(self):
        """SSIM of the 8-bit grey images: piq.ssim(grey(x), grey(y), data_range=255.0) (evaluation_metrics.py:63-76)."""
        return self._score(SSIM, 1)

    def ms_ssim(self):
        """Multi-scale SSIM: piq.multi_scale_ssim(x, y, data_range=1.0) (evaluation_metrics.py:78-89)."""
        return self._score(MS_SSIM, 2)

    def lpips(self) -> float:
        raise NotImplementedError("LPIPS needs the pretrained AlexNet weights lpips.LPIPS(net='alex') downloads; none are available offline")
