"""Device-resident input batches for bench.py and the full-size GPU tests."""
import math
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def _level_lut(torch, device):
    # uint8 levels -> float32 exactly as image.py:80 does (`astype(np.float32) / 255.0`, a true IEEE division): torch divides by a
    # scalar through a reciprocal multiply, which is 1 ulp off for some levels, so the quotients come from a NumPy table
    return torch.from_numpy(np.arange(256, dtype=np.float32) / np.float32(255.0)).to(device)


def synth_batch(torch, B, H, W, seed, device):
    """'mixed' synthetic images of SURVEY.md 8d, generated on the GPU: smooth sinusoidal background, K = ceil(N/32768)
    opaque rectangles, N(0, 1.5^2) noise, rounded to uint8 levels, /255 -> float32 [B, H, W, 3].  Image i uses seed + i."""
    out = torch.empty((B, H, W, 3), dtype=torch.float32, device=device)
    yy = (torch.arange(H, device=device, dtype=torch.float32) / H)[:, None]
    xx = (torch.arange(W, device=device, dtype=torch.float32) / W)[None, :]
    K = -(-(H * W) // 32768)
    for b in range(B):
        rng = np.random.default_rng(seed + b)
        img = out[b]
        for c in range(3):
            fx, fy = rng.integers(1, 4, size=2)
            phi, psi = rng.uniform(0, 2 * np.pi, size=2)
            img[:, :, c] = 127.5 + 80.0 * torch.sin(2 * math.pi * float(fx) * xx + float(phi)) * torch.cos(2 * math.pi * float(fy) * yy + float(psi))
        x0 = rng.integers(0, W, size=K); y0 = rng.integers(0, H, size=K)
        ww = rng.integers(16, 257, size=K); hh = rng.integers(16, 257, size=K)
        col = rng.integers(0, 256, size=(K, 3)).astype(np.float32)
        colt = torch.from_numpy(col).to(device)
        for k in range(K):
            img[y0[k]:y0[k] + hh[k], x0[k]:x0[k] + ww[k], :] = colt[k]
        g = torch.Generator(device=device)
        g.manual_seed(seed + b)
        img += torch.randn(img.shape, generator=g, device=device) * 1.5
        img.round_().clamp_(0, 255)
    lut = _level_lut(torch, device)
    for b in range(B):
        out[b] = lut[out[b].to(torch.int64)]
    return out


def natural_batch(torch, B, H, W, seed, device):
    """Labelled variant (--data natural): the reference's own natural test images (tests/golden/natural/*.png = its test_images/,
    metrics_computation.py:307-324) mirror-tiled to H x W -- reflected copies side by side, so the seams add no artificial edges --
    each batch image from another source image / tile offset; uint8 levels -> float32 by the exact division of image.py:80."""
    from PIL import Image as PILImage
    d = os.path.join(ROOT, "tests", "golden", "natural")
    names = sorted(f for f in os.listdir(d) if f.endswith(".png"))
    srcs = [np.asarray(PILImage.open(os.path.join(d, f)).convert("RGB")) for f in names]
    lut = _level_lut(torch, device)
    out = torch.empty((B, H, W, 3), dtype=torch.float32, device=device)
    for b in range(B):
        k = seed + b
        src = srcs[k % len(srcs)]
        period = np.concatenate([np.concatenate([src, src[:, ::-1]], 1), np.concatenate([src[::-1], src[::-1, ::-1]], 1)], 0)   # 2h x 2w, tiles seamlessly
        ph, pw = period.shape[:2]
        oy, ox = (k * 37) % ph, (k * 53) % pw
        ys = (np.arange(H) + oy) % ph
        xs = (np.arange(W) + ox) % pw
        img = torch.from_numpy(np.ascontiguousarray(period[ys][:, xs])).to(device)
        out[b] = lut[img.to(torch.int64)]
    return out


def to_u8(torch, batch_f32):
    """The uint8 levels of a batch made by the generators above (8-bit ingest variant)."""
    return (batch_f32 * 255.0).round().to(torch.uint8)
