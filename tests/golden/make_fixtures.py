"""Regenerates the committed data fixtures.  Runs in the BUILD container only (it reads
/root/reference/assets, which does not exist on the GPU box).

* raytracer_amd/assets/earthmap_rgb8.npz — assets/earthmap.jpg decoded to RGB8 with PIL
  (libjpeg).  The reference decodes with the `image` crate / jpeg-decoder 0.1.22
  (texture.rs:29-30); IDCT rounding may differ by +-1 LSB: parity unpinned for the decoder,
  which is outside the hot path.  The file is input DATA of the reference, not source.
* tests/golden/earthmap.jpg — the same asset, byte for byte: the INPUT fixture of the host layer's JPEG decoder
  (tests/test_jpeg.py decodes it and compares with the PIL decode above).  An image, i.e. data the reference's scenes
  load (scenes.rs:126), not source.
* tests/golden/reference_kats.json is written by hand from hittable.rs:547-552 and
  alias.rs:188-198 (the reference's own known-answer vectors).
"""
import os
import sys

import numpy as np
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    src = "/root/reference/assets/earthmap.jpg"
    rgb = np.array(Image.open(src).convert("RGB"), dtype=np.uint8)
    assert rgb.shape == (512, 1024, 3), rgb.shape
    out = os.path.join(ROOT, "raytracer_amd", "assets", "earthmap_rgb8.npz")
    np.savez_compressed(out, rgb8=rgb)
    print("wrote", out, os.path.getsize(out), "bytes")
    import shutil
    shutil.copyfile(src, os.path.join(ROOT, "tests", "golden", "earthmap.jpg"))


if __name__ == "__main__":
    sys.exit(main())
