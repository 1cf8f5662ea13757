"""Run the REFERENCE's own Jpeg.compress (src/jpeg/jpeg.py:240-272) on lena.png with the CPU oracle plugged
in for the OpenCV calls it makes, and store the resulting .ajpg bytes (build container only).

    python tests/golden/make_golden_compress.py

This pins everything the reference does *around* OpenCV -- stage order, layer handling, normalisation, the
per-leaf gather + np.pad(reflect), np.round(f32/int32), zigzag gather, concatenation order, state-bit packing,
JSON header and zlib framing -- against the oracle's own orchestration (oracle.encode_image + write_ajpg).
OpenCV's arithmetic itself (resize / CLAHE / blur / bilateral / Canny / dct) is the oracle's restatement on both
sides, so it stays "parity unpinned" (cv2 is not installed).
"""
import hashlib
import json
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402
from _ref_loader import load_reference  # noqa: E402


def make_cv2():
    cv = types.ModuleType("cv2")
    cv.INTER_AREA, cv.INTER_LINEAR = 3, 1

    def resize(src, dsize, interpolation=1):
        src = np.ascontiguousarray(src, dtype=np.float32)
        wd, hd = int(dsize[0]), int(dsize[1])
        h, w = src.shape
        if interpolation == cv.INTER_AREA:
            assert h % hd == 0 and w % wd == 0
            tmp = np.zeros((h, w, 3), np.float32)
            tmp[:, :, 0] = src
            return O.downsample(tmp, 0, h // hd, w // wd)
        assert h == w and hd == wd
        return O.resize_linear_f32(src, wd)

    class _Clahe:
        def apply(self, img):
            return O.clahe(img)

    def createCLAHE(clipLimit, tileGridSize):
        assert clipLimit == 0.75 and tuple(tileGridSize) == (4, 4)
        return _Clahe()

    def GaussianBlur(img, ksize, sigma):
        assert tuple(ksize) == (3, 3) and sigma == 0
        return O.gauss3(img)

    def bilateralFilter(img, d, sc, ss):
        assert (d, sc, ss) == (5, 75, 75)
        return O.bilateral5(img)

    def Canny(img, lo, hi, apertureSize=3, L2gradient=False):
        assert apertureSize == 3 and L2gradient
        return O.canny(img, float(lo), float(hi))

    def dct(block):
        block = np.ascontiguousarray(block, dtype=np.float32)
        s = block.shape[0]
        ones = {s: np.ones((s, s), np.int32)}
        ident = {s: np.arange(s * s, dtype=np.int32)}
        _, d = O.blocks_encode(block, np.array([[0, 0, s]], np.int32), ones, ident, want_dct=True)
        return d.reshape(s, s)

    cv.resize, cv.createCLAHE, cv.GaussianBlur, cv.bilateralFilter, cv.Canny, cv.dct = resize, createCLAHE, GaussianBlur, bilateralFilter, Canny, dct
    return cv


def main():
    from PIL import Image as PILImage
    ref = load_reference(make_cv2())
    Jpeg, Settings, Image = ref["jpeg"].Jpeg, ref["jpeg"].JpegCompressionSettings, ref["image"].Image
    lena = np.asarray(PILImage.open(os.path.join(HERE, "lena.png")).convert("RGB")).astype(np.float32) / 255.0
    crop = np.ascontiguousarray(lena[100:100 + 150, 200:200 + 212])       # ragged: 150x212 -> overhanging leaves
    cases = {
        "lena_ycbcr_8_8_q50": (lena, "YCbCr", (50, 50), (8, 8)),          # BASELINE config 1
        "lena_default_ycocg_4_64": (lena, "YCoCg", (40, 80), (4, 64)),    # reference defaults (jpeg.py:150-155)
        "crop_ycbcr_4_64": (crop, "YCbCr", (40, 80), (4, 64)),
    }
    # natural images the reference ships (test_images/, converted by make_natural_fixtures.py): textures drive the hysteresis pass
    # count and the leaf-size mix very differently from lena and from the synthetic generator
    def natural(name):
        return np.asarray(PILImage.open(os.path.join(HERE, "natural", name + ".png")).convert("RGB")).astype(np.float32) / 255.0
    cases.update({
        "baboon_ycbcr_4_64": (natural("baboon"), "YCbCr", (40, 80), (4, 64)),
        "peppers_default_ycocg_4_64": (natural("peppers"), "YCoCg", (40, 80), (4, 64)),
        "house_ycocgr_2_32_q30_90": (natural("house"), "YCoCg-R", (30, 90), (2, 32)),
        "bikes_ycbcr_4_128": (natural("bikes"), "YCbCr", (40, 80), (4, 128)),
    })
    meta = {}
    for name, (img, space, qr, br) in cases.items():
        codec = Jpeg(Settings(space, qr, br))
        data = codec.compress(Image(img, img.shape, ".png"))
        with open(os.path.join(HERE, name + ".ajpg"), "wb") as f:
            f.write(data)
        # quantisation matrices as the reference built them (through the cv2.resize stand-in)
        qm = {f"{l}_{s}": codec.quantization_matrix_cache[l][s].tolist() for l in range(3) for s in codec.quantization_matrix_cache[l]}
        meta[name] = {"space": space, "quality_range": qr, "block_size_range": br, "shape": list(img.shape),
                      "sha256": hashlib.sha256(data).hexdigest(), "bytes": len(data), "qm": qm,
                      "crop": [100, 200, 150, 212] if name.startswith("crop") else None,
                      "image": "lena" if name.startswith(("lena", "crop")) else "natural/" + name.split("_")[0]}
        print(name, len(data), meta[name]["sha256"][:16])
    json.dump(meta, open(os.path.join(HERE, "compress_cases.json"), "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
