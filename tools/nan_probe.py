"""Dev probe: how many lanes of a guided cornell-box pass end with a non-finite radiance?"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from practical_path_guiding_lab_amd import scene as S
from practical_path_guiding_lab_amd.integrator import PathGuidingIntegrator
from practical_path_guiding_lab_amd.render import IndependentSampler, WavefrontScene

sc = S.cornell_box(512, 512, 8, 8)
g = PathGuidingIntegrator({"max_depth": 8})
g.setup(512 * 512, sc.bbox_min - 1e-4, sc.bbox_max + 1e-4, 20, 20, True, 0.5)
ws = WavefrontScene(sc)
cumm = 0
for k in range(6):
    g.setIteration(k, False)
    bad = 0
    tot = 0
    for _ in range(2 ** (k + 2) // 4):
        L, valid, _ = g.sample(ws, IndependentSampler(4, cumm))
        cumm += 4
        bad += int((~torch.isfinite(L)).any(dim=0).sum().item())
        tot += L.shape[1]
    g.refineAndPrepareSDTreeForNextIteration()
    st = g.sdTree.stats()
    print(f"iter {k}: non-finite lanes {bad} / {tot}  ({bad / tot:.2e})  kd leaves {st.n_kd_leaves} quad recs {st.n_quad_records}")
