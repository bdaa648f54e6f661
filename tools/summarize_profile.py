"""Dev tool: condense a tools/profile_bench.sh output directory into profiles/<round>/.

    python tools/summarize_profile.py gpurun_out/prof_r01b profiles/r01 "render res=512 depth=8 spp=8"
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

src, dst, cfg = sys.argv[1], sys.argv[2], sys.argv[3]
BOUNCES = int(sys.argv[4]) if len(sys.argv) > 4 else 8  # k_bounce launches per pass = max_depth
os.makedirs(dst, exist_ok=True)
KERNELS = ("k_wave_trace", "k_wave_shade_a", "k_wave_cast", "k_wave_guide", "k_wave_shade_b", "k_wave_shade", "k_wave_tail", "k_bounce", "k_splat_list",
           "k_process_and_splat", "k_finish", "k_sort_scatter", "k_sort_hist", "k_sort_scan")  # (k_wave_shade after _a and _b: the first name found in a kernel's name counts)  # k_wave_cast = the persistent any-hit kernel of the shadow rays
PER_BOUNCE = ("k_bounce", "k_wave_trace", "k_wave_shade_a", "k_wave_cast", "k_wave_guide", "k_wave_shade_b", "k_wave_shade")


def short(name):
    for k in KERNELS:
        if k in name:
            return k
    return None


def newest(pattern):
    """gpurun merges every call's files into the same directory: take the latest run's."""
    fs = sorted(glob.glob(pattern), key=os.path.getmtime)
    return fs[-1:] if fs else []


stats = newest(f"{src}/trace/*/*kernel_stats.csv")[0]
shutil.copy(stats, os.path.join(dst, "bench_kernel_stats.csv"))
shutil.copy(os.path.join(src, "bench_under_trace.json"), os.path.join(dst, "bench_under_rocprof.json"))

# exact per-launch durations of the timed region from the trace (last 10 passes)
trace = newest(f"{src}/trace/*/*kernel_trace.csv")[0]
dur = collections.defaultdict(list)
for r in csv.DictReader(open(trace)):
    k = short(r["Kernel_Name"])
    if k:
        dur[k].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
# bench.py times two regions: the default form of the pipeline (one shading kernel per bounce, k_wave_shade), then the
# same passes with pg_render_stages(2), k_wave_guide's only launches.  Every kernel but k_wave_guide is summarised over
# the FIRST region: the launches before the first k_wave_guide.
SPLIT_ONLY = ("k_wave_shade_a", "k_wave_cast", "k_wave_guide", "k_wave_shade_b")
second_region = set(SPLIT_ONLY) if "k_wave_shade" in dur else {"k_wave_guide"}  # kernels that run in the second region only
t_split = min((t for k in second_region for t, _ in dur.get(k, [])), default=None)
if t_split is not None:
    for k in list(dur):
        if k not in second_region:
            dur[k] = [(t, d) for t, d in dur[k] if t < t_split]
trace_summary = {}
for k, v in dur.items():
    v.sort()
    per_pass = BOUNCES if k in PER_BOUNCE else 1
    if k == "k_wave_tail":
        per_pass = sum(1 for b in range(BOUNCES) if BOUNCES > 8 and b >= 4 and b + 1 < BOUNCES and (b < 8 or (b < 16 and b % 2 == 0) or b % 4 == 0))
    last = [d for _, d in v[-10 * per_pass:]]
    trace_summary[k] = {"launches_in_timed_region": len(last), "avg_us": round(sum(last) / len(last) / 1e3, 2),
                        "min_us": round(min(last) / 1e3, 2), "max_us": round(max(last) / 1e3, 2)}
    if t_split is not None and k in second_region:
        trace_summary[k]["region"] = "roofline (pg_render_stages 2)"


PMC_STEPS = 3  # tools/profile_bench.sh runs the counter passes with --steps 3


def agg(sub):
    """Counter values of the launches of the TIMED region only (the last PMC_STEPS passes of the run:
    nothing of these kernels runs after it), in dispatch order."""
    fs = newest(f"{src}/{sub}/*/*counter_collection.csv")
    d = collections.defaultdict(lambda: collections.defaultdict(list))
    if fs:
        rows = sorted(csv.DictReader(open(fs[0])), key=lambda r: int(r["Dispatch_Id"]))
        names = {short(r["Kernel_Name"]) for r in rows}
        second = set(SPLIT_ONLY) if "k_wave_shade" in names else {"k_wave_guide"}
        first_second = min((int(r["Dispatch_Id"]) for r in rows if short(r["Kernel_Name"]) in second), default=None)
        if first_second is not None:  # (see above: the other kernels over the first region only)
            rows = [r for r in rows if short(r["Kernel_Name"]) in second or int(r["Dispatch_Id"]) < first_second]
        for r in rows:
            k = short(r["Kernel_Name"])
            if k:
                d[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k in d:
            per_pass = BOUNCES if k in PER_BOUNCE else 1
            if k == "k_wave_tail":
                per_pass = sum(1 for b in range(BOUNCES) if BOUNCES > 8 and b >= 4 and b + 1 < BOUNCES and (b < 8 or (b < 16 and b % 2 == 0) or b % 4 == 0))
            for c in d[k]:
                d[k][c] = d[k][c][-PMC_STEPS * per_pass:]
    return d


f, w, l2, sq = agg("pmc_fetch"), agg("pmc_write"), agg("pmc_l2"), agg("pmc_sq")
out = {"config": cfg,
       "note": "rocprofv3 --pmc, one counter set per pass, `bench.py --steps 3 --warmup 1 --cpu 0`; means per launch over "
               "the launches of the timed region (its last 3 passes; k_wave_guide: of bench.py's roofline region, the other kernels: of the region `value` is quoted on). FETCH_SIZE/WRITE_SIZE are KiB as reported. hbm_bytes_per_launch = "
               "(2*FETCH_SIZE + WRITE_SIZE)*1024 applies the gfx950 x2 FETCH correction of MI355X_MICROARCH.md section HBM "
               "(calibrated for wide coalesced reads only: an upper bound here).",
       "kernels": {}, "trace": trace_summary}
for k in KERNELS:
    if k not in f or not f[k]["FETCH_SIZE"] or not w[k]["WRITE_SIZE"]:
        continue
    m = lambda d, c: (sum(d[k][c]) / len(d[k][c])) if d[k][c] else None
    fs, ws = m(f, "FETCH_SIZE"), m(w, "WRITE_SIZE")
    e = {"launches": len(f[k]["FETCH_SIZE"]), "FETCH_SIZE_KiB": round(fs, 1), "WRITE_SIZE_KiB": round(ws, 1),
         "hbm_bytes_per_launch_uncorrected": round((fs + ws) * 1024), "hbm_bytes_per_launch": round((2 * fs + ws) * 1024)}
    for c in ("TCC_HIT_sum", "TCC_MISS_sum", "TCC_EA0_ATOMIC_sum"):
        if l2[k][c]:
            e[c] = round(m(l2, c))
    for c in ("SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY",
              "SQ_INSTS_VALU", "SQ_INSTS_VMEM_RD"):
        if sq[k][c]:
            e[c] = round(m(sq, c))
    out["kernels"][k] = e
json.dump(out, open(os.path.join(dst, "pmc_summary.json"), "w"), indent=1)
# profiles/pmc_traffic.json: {"configs": {config key: {kernel: {"hbm_bytes_per_launch": ...}}}}, read by bench.py
tpath = os.path.join(os.path.dirname(dst.rstrip("/")), "pmc_traffic.json")
try:
    traffic = json.load(open(tpath))
    if "configs" not in traffic:
        traffic = {"configs": {}}
except Exception:
    traffic = {"configs": {}}
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from practical_path_guiding_lab_amd._native import source_hash  # noqa: E402
# (run this right after the profile, before the sources change: the hash names the code the counters belong to)
traffic["configs"][cfg] = {k: {"hbm_bytes_per_launch": v["hbm_bytes_per_launch"],
                               "hbm_bytes_per_launch_uncorrected": v["hbm_bytes_per_launch_uncorrected"],
                               "atomic_sector_updates_per_launch": v.get("TCC_EA0_ATOMIC_sum")}
                           for k, v in out["kernels"].items()}
traffic["configs"][cfg]["_source_hash"] = source_hash()
try:  # the table resolutions of the forest the counters were taken of (bench.py flags a run whose forest has others)
    bj = json.load(open(os.path.join(dst, "bench_under_rocprof.json")))
    traffic["configs"][cfg]["_jump_bits"] = int(bj["config"]["jump_bits"])
    traffic["configs"][cfg]["_kd_grid_bits"] = int(bj["config"]["kd_grid_bits"])
except Exception:
    pass
json.dump(traffic, open(tpath, "w"), indent=1)
print(json.dumps(out, indent=1))
