"""Dev tool (GPU box, repo root): the guided-lifecycle parity of tests/test_gpu_render.py (radiance per lane,
pixel sums, accumulators and refined trees bit for bit against the CPU oracle over four iterations) at sizes
the test suite does not afford -- the oracle runs on all host cores.    python tools/soak_parity.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import test_gpu_render as T  # noqa: E402
from oracle import pg_oracle as po  # noqa: E402
from practical_path_guiding_lab_amd import scene as S  # noqa: E402

po.set_threads(0)
cases = [("veach-ajar 640x360", lambda: S.veach_ajar(640, 360)), ("torus 480x360", lambda: S.torus(480, 360)),
         ("veach-mis 640x360 depth 14", lambda: S.veach_mis(640, 360, 14, 10)), ("cornell-box 256 depth 12", lambda: S.cornell_box(256, 256, 12, 9)),
         ("mixed 192 depth 13", lambda: T.mixed_scene(192, max_depth=13, rr_depth=10))]
if len(sys.argv) > 1 and sys.argv[1] == "full":  # the bench's own film: 124 M paths, about two minutes of oracle on 256 threads
    cases = [("veach-ajar 1920x1080 (the bench size)", lambda: S.veach_ajar(1920, 1080)),
             ("cornell-box 512x512 depth 8 (the bench size)", lambda: S.cornell_box(512, 512, 8, 8)),
             ("veach-mis 1280x720 depth 3 (the bench size)", lambda: S.veach_mis(1280, 720, 3, 8)),
             ("torus 1920x1080 depth 32 (the bench size)", lambda: S.torus(1920, 1080, 32, 8))]
    if len(sys.argv) > 2:
        cases = [c for c in cases if sys.argv[2] in c[0]]
for name, make in cases:
    t0 = time.time()
    sc = make()
    T._guided_lifecycle_bit_exact(sc, True)
    print(f"{name}: {sc.camera.width * sc.camera.height * 60 / 1e6:.1f} M paths bit-exact in {time.time() - t0:.1f} s", flush=True)
print("soak ok")
