"""Dev tool (GPU box): how many host cores does this process really have?  nproc / affinity / cgroup quota, and the CPU
oracle's pdf rate on S1 with 8 ... 256 threads (bench.py's cpu columns pick their thread count from this)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from oracle import pg_oracle as po  # noqa: E402
from practical_path_guiding_lab_amd import workload as W  # noqa: E402

print("os.cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us"):
    try:
        print(f, open(f).read().strip())
    except OSError as e:
        print(f, "-", e.__class__.__name__)
po.build()
t = po.OracleTree()
t.load(W.s1_balanced_tree())
n = 1 << 22
P = W.s_positions_uniform(n, 3).numpy()
D = W.s_directions_uniform(n, 4).numpy()
for th in (8, 16, 32, 64, 128, 256):
    po.set_threads(th)
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        t.pdf(P, D)
        best = min(best, time.perf_counter() - t0)
    print(f"threads {th:4d}: pdf {n / best / 1e6:8.2f} M/s", flush=True)
