"""Copy the reference's own natural test images (test_images/*.tiff and two of the LIVE database BMPs its sweep runs on,
test/analysis/metrics_computation.py:307-324) into tests/golden/natural/ as PNG (lossless: the decoded uint8 pixels are identical,
asserted below).  Build container only -- /root/reference does not exist on the GPU box.

    python tests/golden/make_natural_fixtures.py
"""
import os

import numpy as np
from PIL import Image as PILImage

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = "/root/reference/test_images"
PICKS = ["baboon.tiff", "peppers.tiff", "house.tiff", "jelly_beans.tiff",
         "LIVE_image_quality_assessment_database/bikes.bmp", "LIVE_image_quality_assessment_database/buildings.bmp"]


def main():
    out = os.path.join(HERE, "natural")
    os.makedirs(out, exist_ok=True)
    for rel in PICKS:
        px = np.asarray(PILImage.open(os.path.join(SRC, rel)).convert("RGB"))
        dst = os.path.join(out, os.path.splitext(os.path.basename(rel))[0] + ".png")
        PILImage.fromarray(px).save(dst, optimize=True)
        assert np.array_equal(np.asarray(PILImage.open(dst).convert("RGB")), px)
        print(dst, px.shape, os.path.getsize(dst))


if __name__ == "__main__":
    main()
