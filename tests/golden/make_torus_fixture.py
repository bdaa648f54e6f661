"""Makes practical_path_guiding_lab_amd/data/torus_meshes.npz: the five meshes scenes/torus/scene.xml refers to
(`meshes.serialized`, shape_index 1..5), read with practical_path_guiding_lab_amd.mesh.read_serialized
and stored as plain arrays (float32 positions and normals exactly as in the file, integer faces), so
that the torus scene can be built where the reference's files are absent (the GPU box).

    python tests/golden/make_torus_fixture.py [/root/reference/scenes/torus/meshes.serialized]
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from practical_path_guiding_lab_amd.mesh import read_serialized  # noqa: E402

NAMES = {1: "floor", 2: "donut", 3: "glass", 4: "metal_a", 5: "metal_b"}


def main():
    src = sys.argv[1] if len(sys.argv) > 1 else "/root/reference/scenes/torus/meshes.serialized"
    out = {}
    for idx, name in NAMES.items():
        v, f, n = read_serialized(src, idx)
        assert n is not None and f.max() < v.shape[0]
        out[f"{name}_v"] = v.astype(np.float32)
        out[f"{name}_n"] = n.astype(np.float32)
        out[f"{name}_f"] = f.astype(np.uint16 if v.shape[0] <= 65536 else np.uint32)
        print(name, v.shape[0], "vertices", f.shape[0], "faces")
    dst = os.path.join(os.path.dirname(__file__), "..", "..", "practical_path_guiding_lab_amd", "data", "torus_meshes.npz")
    np.savez_compressed(dst, **out)
    print("wrote", dst, os.path.getsize(dst), "bytes")


if __name__ == "__main__":
    main()
