"""Dev tool: per-kernel durations and gaps of the last timed pass from a rocprofv3 kernel trace."""
import csv
import glob
import sys

d = sys.argv[1]
f = glob.glob(f"{d}/*/*kernel_trace.csv")[0]
rows = [r for r in csv.DictReader(open(f))]
allk = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:34]) for r in rows)
marks = [i for i, k in enumerate(allk) if "k_process_and_splat" in k[2]]
# the refine at the end adds no splat; last two splats delimit the final timed pass
start, idx = marks[-2], marks[-1]
print("pass span us %.1f" % ((allk[idx][1] - allk[start][1]) / 1e3))
busy = 0.0
for i in range(start + 1, idx + 1):
    dur = (allk[i][1] - allk[i][0]) / 1e3
    busy += dur
    print("  %-36s dur %7.1f gap %6.1f" % (allk[i][2], dur, (allk[i][0] - allk[i - 1][1]) / 1e3))
print("busy us %.1f" % busy)
