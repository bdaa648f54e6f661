"""Generates tests/golden/cornell_gt_256_f16.npy: the reference's ground-truth image of cornell-box
(scenes/cornell-box/TungstenRender.exr, 1024x1024 HALF/PIZ, the file main.py:38-41 loads) decoded
with practical_path_guiding_lab_amd/exr.py and box-downsampled 4x4 to 256x256 (float16, 384 KiB).
It is data of the reference, not code; the GPU box has no /root/reference, hence the fixture.

    python tests/golden/make_gt_fixture.py [/root/reference]
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from practical_path_guiding_lab_amd import exr  # noqa: E402


def main():
    ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
    img = exr.read_rgb(os.path.join(ref, "scenes", "cornell-box", "TungstenRender.exr"))
    assert img.shape == (1024, 1024, 3)
    small = img.reshape(256, 4, 256, 4, 3).astype(np.float64).mean(axis=(1, 3))
    np.save(os.path.join(HERE, "cornell_gt_256_f16.npy"), small.astype(np.float16))
    print("cornell-box mean radiance", img.mean(), "->", small.mean())
    # veach-mis: 1280x720 -> 320x180.  The lamps exceed the float16 range: clamp at 60000 (three pixels)
    img = exr.read_rgb(os.path.join(ref, "scenes", "veach-mis", "TungstenRender.exr"))
    assert img.shape == (720, 1280, 3)
    small = img.reshape(180, 4, 320, 4, 3).astype(np.float64).mean(axis=(1, 3))
    np.save(os.path.join(HERE, "veach_mis_gt_320x180_f16.npy"), np.minimum(small, 60000.0).astype(np.float16))
    print("veach-mis mean radiance", img.mean(), "->", small.mean(), "max", small.max())
    # torus: the reference ships an 8-bit sRGB image only (scenes/torus/TungstenRender.png, 1024x768);
    # linearised (inverse sRGB curve), box-downsampled 4x4 to 256x192, stored as float16.  Values that
    # were clipped at 1.0 in the PNG stay clipped: tests skip blocks holding them.
    from PIL import Image
    png = np.asarray(Image.open(os.path.join(ref, "scenes", "torus", "TungstenRender.png")).convert("RGB")).astype(np.float64) / 255.0
    assert png.shape == (768, 1024, 3)
    lin = np.where(png <= 0.04045, png / 12.92, ((png + 0.055) / 1.055) ** 2.4)
    small = lin.reshape(192, 4, 256, 4, 3).mean(axis=(1, 3))
    np.save(os.path.join(HERE, "torus_gt_256x192_f16.npy"), small.astype(np.float16))
    print("torus mean (linearised sRGB)", lin.mean(), "->", small.mean())


if __name__ == "__main__":
    main()
