"""Dev tool (no GPU): the memory skeleton of one kernel's listing -- every vector load, store, atomic, LDS operation, barrier
and `s_waitcnt vmcnt(N)` in program order with its instruction number -- so that round trips that depend on nothing can be SEEN:
a load followed at once by `vmcnt(0)` and then by a load that does not need its data, loops of load -> wait -> store, loads that
begin behind a barrier.  Round 6 found 27.5 -> 25.0 ms per step of k_wave_shade, 11.4 -> 10.9 of k_wave_trace and the cause of
round 5's "60 % for a never-taken branch" this way; counters do not show any of it.

    python tools/listing_waits.py pg_render_wave.hip "k_wave_shade<2, false>" [--from N] [--to N] [--extra "-DPG_SHADE_PHASES=1"]
    python tools/listing_waits.py pg_kernels_splat.hip k_splat_list --summary

--summary: one line per kernel of the file instead (instructions, loads, full drains, the longest chain of load -> vmcnt(0) pairs)."""
import argparse
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "practical_path_guiding_lab_amd", "csrc")
FLAGS = "-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt --cuda-device-only -S -g0".split()
SHOW = re.compile(r"global_load|global_store|global_atomic|buffer_load|buffer_store|scratch_|s_waitcnt vmcnt|s_barrier|ds_read|ds_write|ds_bpermute|ds_add|Loop Header")


def kernels(listing):
    out, cur = {}, None
    for line in listing.split("\n"):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            cur = m.group(1)
            out[cur] = []
        elif line.startswith(".Lfunc_end"):
            cur = None
        elif cur is not None:
            out[cur].append(line)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("file")
    ap.add_argument("kernel", nargs="?", default="")
    ap.add_argument("--from", dest="lo", type=int, default=0)
    ap.add_argument("--to", dest="hi", type=int, default=1 << 30)
    ap.add_argument("--extra", default="")
    ap.add_argument("--summary", action="store_true")
    ap.add_argument("--lds", action="store_true", help="also LDS operations (off by default: they are many)")
    a = ap.parse_args()
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "k.s")
        r = subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + a.extra.split() + [a.file, "-o", out], cwd=SRC, capture_output=True, text=True)
        if r.returncode != 0:
            sys.exit(r.stderr[-2000:])
        ks = kernels(open(out).read())
    names = subprocess.run(["c++filt"] + list(ks), capture_output=True, text=True).stdout.split("\n")
    for mangled, dem in zip(ks, names):
        short = re.sub(r"\(.*", "", dem.replace("(anonymous namespace)::", "")).replace("void ", "").replace("pg::", "")
        if a.kernel and a.kernel.replace(" ", "") not in short.replace(" ", ""):
            continue
        n, ev, rows = 0, [], []
        for line in ks[mangled]:
            ins = line.startswith("\t") and not line.strip().startswith(";")
            if ins:
                n += 1
            if not SHOW.search(line) or (not a.lds and re.search(r"\bds_", line)):
                continue
            rows.append((n, line.strip()[:110]))
            if re.search(r"(global|buffer)_load", line):
                ev.append((n, "L"))
            elif "vmcnt(0)" in line:
                ev.append((n, "W"))
        if a.summary or not a.kernel:
            chain = best = 0
            for x, y in zip(ev, ev[1:]):
                if x[1] == "L" and y[1] == "W" and y[0] - x[0] <= 6:
                    chain += 1
                    best = max(best, chain)
                elif not (x[1] == "W" and y[1] == "L"):
                    chain = 0
            print("%-64s %6d instructions %4d loads %4d full drains  longest load->vmcnt(0) chain %d" %
                  (short[:64], n, sum(1 for e in ev if e[1] == "L"), sum(1 for e in ev if e[1] == "W"), best))
        else:
            print("# %s: %d instructions" % (short, n))
            for k, text in rows:
                if a.lo <= k <= a.hi:
                    print("%6d  %s" % (k, text))


if __name__ == "__main__":
    main()
