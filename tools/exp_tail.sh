#!/bin/bash
# Dev experiment (GPU box, repo root): the live-path threshold below which k_bounce_tail takes over.
# Rebuilds the library per value, prints the torus bench's kernel times, and for the first value
# the per-launch durations of one pass from a rocprofv3 kernel trace.
# whatever happens, leave the DEFAULT build behind: variant objects are newer than the sources, so a later
# `make` (or __graft_entry__.build()) would otherwise keep shipping the experiment
trap 'touch practical_path_guiding_lab_amd/csrc/*.hip; make -s -C practical_path_guiding_lab_amd/csrc -j8' EXIT
set -e
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/exp_tail
R=$PWD
mkdir -p $OUT
first=1
for t in ${@:-524288 131072 32768 2097152}; do
	touch practical_path_guiding_lab_amd/csrc/pg_render.hip
	make -s -C practical_path_guiding_lab_amd/csrc -j8 EXTRA="-DPG_TAIL_PATHS=${t}u" > $OUT/make.log 2>&1
	python bench.py --scene torus --cpu 0 --steps 10 --detail $OUT/torus.$t.json > $OUT/torus.$t.line
	python - <<EOF
import json
d = json.load(open("$OUT/torus.$t.json"))
print("tail paths %8d  value %7.1f  ms %.3f  bounce step %.1f us  splat %.1f us" % ($t, d["value"], d["ms_per_step"], d["kernels"]["k_bounce"]["avg_us"], d["kernels"]["k_splat_list"]["avg_us"]))
EOF
	if [ $first = 1 ]; then
		first=0
		(cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 $R/bench.py --scene torus --steps 3 --warmup 1 --cpu 0 --detail $OUT/trace.json > $OUT/trace.line 2> $OUT/trace.err)
		python - <<EOF
import csv, glob
f = glob.glob("$OUT/trace/*/*kernel_trace.csv")[0]
rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f)))
names = [("tail" if "k_bounce_tail" in k else "b") + ":%.0f" % ((e - s) / 1e3) for s, e, k in rows if "k_bounce" in k]
print("last pass:", " ".join(names[-41:]))
EOF
	fi
done
