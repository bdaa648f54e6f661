"""Makes the veach-ajar data the GPU box needs (it has no /root/reference):

  practical_path_guiding_lab_amd/data/veach_ajar.npz
      the 15 OBJ meshes scenes/veach-ajar/scene.xml refers to that exist in the reference mount
      (Mesh000.obj and Mesh009.obj, the teapots, are missing blobs), read with mesh.read_obj:
      positions (as parsed: float64), integer faces, per-corner texture coordinates as the files hold them, and
      per-corner normals for the one mesh the scene shades smoothly (Mesh015, the door handle);
      the three JPG textures as the byte strings of their files (jpg_landscape 1920x1280, jpg_table 2000x3008,
      jpg_cherry 1280x1024; 2.6 MB): scene.veach_ajar() decodes them with PIL exactly as scene.load_xml does
      with the files themselves, so the packaged scene has the reference's textures at full resolution
      (round 2 shipped box-downsampled copies).
  tests/golden/veach_ajar_gt_320x180_f16.npy, veach_ajar_gt_640x360_f16.npy
      scenes/veach-ajar/TungstenRender.exr (1280x720 HALF/PIZ, what main.py:38-41 loads) decoded with
      practical_path_guiding_lab_amd/exr.py and box-downsampled 4x4 / 2x2, float16.

It is data of the reference, not code.    python tests/golden/make_ajar_fixture.py [/root/reference]
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from practical_path_guiding_lab_amd import exr  # noqa: E402
from practical_path_guiding_lab_amd.mesh import read_obj  # noqa: E402

MESHES = ["Mesh%03d" % i for i in (1, 2, 3, 4, 5, 6, 7, 8, 10, 11, 12, 13, 14, 15, 16)]
SMOOTH = {"Mesh015"}  # every other shape sets face_normals=true (scenes/veach-ajar/scene.xml:133-250)
TEXTURES = {"landscape": "landscape-with-a-lake.jpg", "table": "Good Textures_005844.jpg", "cherry": "cherry-wood-texture.jpg"}


def main():
    from PIL import Image

    ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
    base = os.path.join(ref, "scenes", "veach-ajar")
    out = {}
    for name in MESHES:
        v, f, uv, n = read_obj(os.path.join(base, "models", name + ".obj"), attributes=True)
        assert uv is not None and n is not None and f.max() < v.shape[0]
        out[name + "_v"] = v  # float64: what parsing the decimal text gives (to_world is applied before rounding to fp32)
        out[name + "_f"] = f.astype(np.uint16 if v.shape[0] <= 65536 else np.uint32)
        out[name + "_uv"] = uv.astype(np.float32)
        if name in SMOOTH:
            out[name + "_n"] = n
        print(name, v.shape[0], "vertices", f.shape[0], "triangles")
    for key, fname in TEXTURES.items():
        raw = open(os.path.join(base, "textures", fname), "rb").read()
        out["jpg_" + key] = np.frombuffer(raw, np.uint8)
        print(key, Image.open(os.path.join(base, "textures", fname)).size, len(raw), "bytes")
    dst = os.path.join(ROOT, "practical_path_guiding_lab_amd", "data", "veach_ajar.npz")
    np.savez_compressed(dst, **out)
    print("wrote", dst, os.path.getsize(dst), "bytes")
    gt = exr.read_rgb(os.path.join(base, "TungstenRender.exr"))
    assert gt.shape == (720, 1280, 3)
    for f in (4, 2):
        small = gt.reshape(720 // f, f, 1280 // f, f, 3).astype(np.float64).mean(axis=(1, 3))
        np.save(os.path.join(HERE, "veach_ajar_gt_%dx%d_f16.npy" % (1280 // f, 720 // f)), np.minimum(small, 60000.0).astype(np.float16))
        print("ground truth", gt.mean(), "->", small.mean(), "max", small.max())


if __name__ == "__main__":
    main()
