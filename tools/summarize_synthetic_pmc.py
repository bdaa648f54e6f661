"""Dev tool: FETCH_SIZE / WRITE_SIZE of the stand-alone entry points on S1 / S2 / S3 (tools/profile_bench.sh, its last two
passes) and their durations in the kernel trace of the same script -> <dst>/synthetic_pmc.json.
    python tools/summarize_synthetic_pmc.py gpurun_out/prof_r04a profiles/r04
Launches of one kernel come in two groups, S1's then S2's (bench.py: synthetic_kernels_leg); within a group the steady
launches are the median."""
import collections
import csv
import glob
import json
import os
import statistics
import sys

src, dst = sys.argv[1], sys.argv[2]
NAMES = ("k_sample", "k_pdf", "k_guide_bounce", "k_leaf_index", "k_splat(")


def short(n):
    for k in NAMES:
        if "pg::" + k in n or n.startswith(k):
            return k.rstrip("(")
    return None


def newest(p):
    fs = sorted(glob.glob(p), key=os.path.getmtime)
    return fs[-1] if fs else None


def counters(sub, counter):
    f = newest(f"{src}/{sub}/*/*counter_collection.csv")
    d = collections.defaultdict(list)
    if f:
        for r in sorted(csv.DictReader(open(f)), key=lambda r: int(r["Dispatch_Id"])):
            k = short(r["Kernel_Name"])
            if k and r["Counter_Name"] == counter:
                d[k].append(float(r["Counter_Value"]))
    return d


def halves(v):
    """S1's launches then S2's (k_splat: S2's growth, then S3's replays = the last 7)."""
    h = len(v) // 2
    return v[:h], v[h:]


fetch, write = counters("pmc_syn_fetch", "FETCH_SIZE"), counters("pmc_syn_write", "WRITE_SIZE")
trace = newest(f"{src}/trace/*/*kernel_trace.csv")
dur = collections.defaultdict(list)
for r in sorted(csv.DictReader(open(trace)), key=lambda r: int(r["Start_Timestamp"])):
    k = short(r["Kernel_Name"])
    if k:
        dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
out = {"note": __doc__.split("\n\n")[0] if False else "FETCH_SIZE / WRITE_SIZE in KiB as rocprofv3 reports them (gfx950: FETCH_SIZE counts half of a wide coalesced read, "
       "MI355X_MICROARCH.md); per kernel the median over its steady launches on S1 and on S2 (k_splat: S3 = the replays of the 2^24-record "
       "stream, the last launches); durations from the kernel trace of the same script's bench run", "kernels": {}}
for k in ("k_leaf_index", "k_pdf", "k_sample", "k_guide_bounce"):
    if k not in fetch:
        continue
    e = {}
    for tag, fv, wv, dv in zip(("S1", "S2"), halves(fetch[k]), halves(write.get(k, [])), halves(dur.get(k, []))):
        fm, wm = statistics.median(fv), (statistics.median(wv) if wv else 0.0)
        e[tag] = {"launches": len(fv), "FETCH_SIZE_KiB": round(fm, 1), "WRITE_SIZE_KiB": round(wm, 1),
                  "hbm_bytes_uncorrected": round((fm + wm) * 1024), "hbm_bytes_x2_fetch": round((2 * fm + wm) * 1024),
                  "trace_median_us": round(statistics.median(dv), 2) if dv else None, "trace_launches": len(dv)}
    out["kernels"][k] = e
if "k_splat" in fetch:
    fv, wv, dv = fetch["k_splat"][-5:], write.get("k_splat", [])[-5:], dur.get("k_splat", [])[-5:]
    fm, wm = statistics.median(fv), (statistics.median(wv) if wv else 0.0)
    out["kernels"]["k_splat"] = {"S3": {"launches": len(fv), "FETCH_SIZE_KiB": round(fm, 1), "WRITE_SIZE_KiB": round(wm, 1),
                                        "hbm_bytes_uncorrected": round((fm + wm) * 1024), "hbm_bytes_x2_fetch": round((2 * fm + wm) * 1024),
                                        "trace_median_us": round(statistics.median(dv), 2) if dv else None}}
os.makedirs(dst, exist_ok=True)
json.dump(out, open(os.path.join(dst, "synthetic_pmc.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
