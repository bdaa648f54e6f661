import sys, time, numpy as np, torch
sys.path.insert(0, "/root/repo")
from practical_path_guiding_lab_amd import scene as S
from practical_path_guiding_lab_amd.integrator import PathGuidingIntegrator
from practical_path_guiding_lab_amd.render import IndependentSampler, WavefrontScene, render
sc = S.veach_ajar(1920, 1080, 13, 8)
g = PathGuidingIntegrator({"max_depth": 13, "rr_depth": 8})
g.setup(1920*1080, sc.bbox_min - np.float32(1e-4), sc.bbox_max + np.float32(1e-4), 20, 20, True, 0.5)
ws = WavefrontScene(sc); ws.reserve(g, 16)
cumm = 0
for k in range(6):
    g.setIteration(k, False); g.resetVarianceCounter()
    n = 2 ** (k + 2); c = min(16, n)
    for i in range(n // c):
        g.sample(ws, IndependentSampler(c, cumm + i * c, batched=True))
    cumm += n
    g.refineAndPrepareSDTreeForNextIteration()
tree = g.sdTree
for final in (False, True):
    g.setIteration(6, final)
    tree.enableKernelTiming(True)
    for _ in range(2): g.sample(ws, IndependentSampler(16, cumm)); 
    tree.readKernelTiming(reset=True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(8): g.sample(ws, IndependentSampler(16, cumm + 16 * i))
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 8
    kt = tree.readKernelTiming(reset=True)
    print("final" if final else "train", f"{dt*1e3:.2f} ms/pass", {k: round(getattr(kt, k) / 8, 2) for k in ("trace_ms", "shade_ms", "sort_ms", "splat_ms", "finish_ms", "tail_ms")})
    tree.enableKernelTiming(False)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(8): render(ws, g, spp=16, seed=cumm + 16 * i)
    torch.cuda.synchronize(); print("   with the film (render()):", f"{(time.perf_counter() - t0) / 8 * 1e3:.2f} ms/pass")
