"""CPU restatement of the reference's EvaluationMetrics (src/image/evaluation_metrics.py:50-89) -- TEST INFRASTRUCTURE ONLY.

Only tests/ may import this module; the product (adaptive_edge_aware_jpeg_amd) never does.

The arithmetic lives in two third-party packages that are absent from /root/reference and from this container:
piq==0.8.0 (requirements.txt:22) and opencv-python==4.11.0.86 (requirements.txt:18).  Their published algorithms are
restated here in numpy:
  * piq/psnr.py `psnr`: x, y / data_range; mse = mean((x - y)^2) over (C, H, W); -10 log10(mse + 1e-8)
  * piq/functional/filters.py `gaussian_filter`: coords = arange(k) - (k - 1) / 2; g = exp(-(c_i^2 + c_j^2) / (2 sigma^2)); g /= g.sum()
    (float32, as the tensors are)
  * piq/ssim.py `ssim`: x, y / data_range; f = max(1, round(min(H, W) / 256)); avg_pool2d(f) if f > 1;
    `_ssim_per_channel`: 'valid' correlation with the window of x, y, x^2, y^2, xy; c1 = k1^2, c2 = k2^2 (k1 = .01, k2 = .03);
    cs = (2 s_xy + c2) / (s_xx + s_yy + c2); ss = (2 mu_xy + c1) / (mu_xx + mu_yy + c1) * cs; spatial means; channel mean
  * piq/ms_ssim.py `multi_scale_ssim`: weights (0.0448, 0.2856, 0.3001, 0.2363, 0.1333); between scales
    pad = max(H % 2, W % 2) replicated on the left and top, then avg_pool2d(2); relu; prod(mcs^w) with the last scale's ssim; channel mean
  * cv.cvtColor(COLOR_RGB2GRAY) on uint8: (R * 4899 + G * 9617 + B * 1868 + 8192) >> 14
Call sites and arguments are the reference's own (evaluation_metrics.py:57-61, 70-76, 85-89).  No golden vector exists for
these in the reference's tests and neither package can run here: **parity unpinned** for this row -- the GPU path is checked
against this restatement to a float tolerance, plus independent properties (identical images, known MSE).
"""
import numpy as np


def get_uint8(data):                      # image.py:120-127
    return (data * 255).astype(np.uint8)


def rgb2gray_u8(u8):
    r, g, b = (u8[..., i].astype(np.int64) for i in range(3))
    return ((r * 4899 + g * 9617 + b * 1868 + (1 << 13)) >> 14).astype(np.uint8)


def gaussian_window(k=11, sigma=1.5):
    c = np.arange(k, dtype=np.float32) - np.float32((k - 1) / 2.0)
    g = c ** 2
    g = np.exp(-(g[None, :] + g[:, None]) / np.float32(2 * sigma ** 2)).astype(np.float32)
    return (g / g.sum(dtype=np.float32)).astype(np.float32)


def _valid_corr(x, win):
    """'valid' 2-D correlation of (C, H, W) with a (k, k) window, accumulated in float64."""
    k = win.shape[0]
    C, H, W = x.shape
    out = np.zeros((C, H - k + 1, W - k + 1), np.float64)
    xd = x.astype(np.float64)
    for i in range(k):
        for j in range(k):
            out += float(win[i, j]) * xd[:, i:i + H - k + 1, j:j + W - k + 1]
    return out


def _ssim_per_channel(x, y, win, k1=0.01, k2=0.03):
    if x.shape[-1] < win.shape[-1] or x.shape[-2] < win.shape[-2]:
        raise ValueError(f"Kernel size can't be greater than actual input size. Input size: {x.shape}. Kernel size: {win.shape}")
    c1, c2 = k1 ** 2, k2 ** 2
    mu_x, mu_y = _valid_corr(x, win), _valid_corr(y, win)
    mu_xx, mu_yy, mu_xy = mu_x ** 2, mu_y ** 2, mu_x * mu_y
    s_xx = _valid_corr(x.astype(np.float64) ** 2, win) - mu_xx
    s_yy = _valid_corr(y.astype(np.float64) ** 2, win) - mu_yy
    s_xy = _valid_corr(x.astype(np.float64) * y.astype(np.float64), win) - mu_xy
    cs = (2.0 * s_xy + c2) / (s_xx + s_yy + c2)
    ss = (2.0 * mu_xy + c1) / (mu_xx + mu_yy + c1) * cs
    return ss.mean(axis=(-1, -2)), cs.mean(axis=(-1, -2))


def _avg_pool(x, f):
    C, H, W = x.shape
    h, w = H // f, W // f
    return x[:, :h * f, :w * f].reshape(C, h, f, w, f).astype(np.float64).mean(axis=(2, 4)).astype(np.float32)


def psnr(a, b):
    """a, b: (H, W, 3) float32 in [0, 1]."""
    mse = np.mean((a.astype(np.float64) - b.astype(np.float64)) ** 2)
    return float(-10.0 * np.log10(mse + 1e-8))


def ssim(a, b):
    ga = rgb2gray_u8(get_uint8(a))[None].astype(np.float32) / np.float32(255.0)
    gb = rgb2gray_u8(get_uint8(b))[None].astype(np.float32) / np.float32(255.0)
    f = max(1, round(min(ga.shape[-2:]) / 256))
    if f > 1:
        ga, gb = _avg_pool(ga, f), _avg_pool(gb, f)
    ss, _ = _ssim_per_channel(ga, gb, gaussian_window())
    return float(ss.mean())


def ms_ssim(a, b):
    x = np.ascontiguousarray(a.transpose(2, 0, 1)).astype(np.float32)
    y = np.ascontiguousarray(b.transpose(2, 0, 1)).astype(np.float32)
    weights = np.array([0.0448, 0.2856, 0.3001, 0.2363, 0.1333], dtype=np.float32).astype(np.float64)
    win = gaussian_window()
    levels = len(weights)
    min_size = (win.shape[-1] - 1) * 2 ** (levels - 1) + 1
    if x.shape[-1] < min_size or x.shape[-2] < min_size:
        raise ValueError(f"Invalid size of the input images, expected at least {min_size}x{min_size}.")
    mcs = []
    ss = None
    for it in range(levels):
        if it > 0:
            p = max(x.shape[1] % 2, x.shape[2] % 2)
            x = np.pad(x, ((0, 0), (p, 0), (p, 0)), mode="edge")
            y = np.pad(y, ((0, 0), (p, 0), (p, 0)), mode="edge")
            x, y = _avg_pool(x, 2), _avg_pool(y, 2)
        ss, cs = _ssim_per_channel(x, y, win)
        mcs.append(cs)
    stack = np.maximum(np.stack(mcs[:-1] + [ss], axis=0), 0.0)           # (level, channel)
    return float(np.prod(stack ** weights[:, None], axis=0).mean())
