"""Dev tool: per-pass kernel durations from a rocprofv3 kernel trace of bench.py.
A pass = the kernels between two k_finish launches; prints the bounce durations of the first
`n` passes (training: iterations 0 and 1 are unguided) and of the last one."""
import csv
import glob
import sys

d = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 6
f = glob.glob(f"{d}/*/*kernel_trace.csv")[0]
rows = [r for r in csv.DictReader(open(f))]
allk = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0]) for r in rows)
passes, cur = [], []
for s, e, k in allk:
    if "k_bounce" in k or "k_process_and_splat" in k or "k_finish" in k:
        cur.append((k.replace("void pg::", "").replace("pg::", ""), (e - s) / 1e3))
    if "k_finish" in k:
        passes.append(cur)
        cur = []
for i in list(range(min(n, len(passes)))) + [len(passes) - 1]:
    p = passes[i]
    b = " ".join("%6.1f" % t for k, t in p if "k_bounce" in k)
    o = " ".join("%s %.1f" % (k[:9], t) for k, t in p if "k_bounce" not in k)
    print("pass %3d: bounces %s | %s" % (i, b, o))
