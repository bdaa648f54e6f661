"""Dev tool (GPU box, repo root): the guided-lifecycle parity check of tests/test_gpu_render.py
(radiance, pixel sums, accumulators and refined trees bit for bit against the CPU oracle) at sizes
larger than the test suite uses, for every scene and feature level, including the long-path
configurations that end in the tail launch.  About 10 s.

    python tools/soak_parity.py
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)

import test_gpu_render as T  # noqa: E402
from practical_path_guiding_lab_amd import scene as S  # noqa: E402

CASES = [
    ("torus 160x120", lambda: S.torus(160, 120)),
    ("mixed 96", lambda: T.mixed_scene(96)),
    ("mixed 64 depth 13", lambda: T.mixed_scene(64, max_depth=13, rr_depth=10)),
    ("veach-mis 160x90 depth 14", lambda: S.veach_mis(160, 90, 14, 10)),
    ("cornell-box 96 depth 12", lambda: S.cornell_box(96, 96, 12, 9)),
]

if __name__ == "__main__":
    for name, make in CASES:
        t = time.time()
        T._guided_lifecycle_bit_exact(make(), True)
        print(f"{name}: bit-exact over the lifecycle, {time.time() - t:.1f} s", flush=True)
