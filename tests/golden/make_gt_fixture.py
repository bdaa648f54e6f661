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
    print("mean radiance", img.mean(), "->", small.mean())


if __name__ == "__main__":
    main()
