#!/bin/bash
# Dev experiment (GPU box, repo root): where does k_bounce's time go?  Rebuilds the library with
# timing-only ablations (results are wrong on purpose) and prints the bench's kernel times.
# whatever happens, leave the DEFAULT build behind: variant objects are newer than the sources, so a later
# `make` (or __graft_entry__.build()) would otherwise keep shipping the experiment
trap 'touch practical_path_guiding_lab_amd/csrc/*.hip; make -s -C practical_path_guiding_lab_amd/csrc -j8' EXIT
set -e
OUT=gpurun_out/exp_ablate
mkdir -p $OUT
run() {
	python bench.py --cpu 0 --steps 10 --detail $OUT/$1.json > $OUT/$1.line
	python - <<EOF
import json
d = json.load(open("$OUT/$1.json"))
print("%-14s value %7.1f  ms %.3f  bounce %.1f us  splat %.1f us" % ("$1", d["value"], d["ms_per_step"], d["kernels"]["k_bounce"]["avg_us"], d["kernels"]["k_splat_list"]["avg_us"]))
EOF
}
run baseline
for v in SHADOW TRIG "SHADOW -DPG_ABLATE_TRIG"; do
	touch practical_path_guiding_lab_amd/csrc/*.hip
	make -s -C practical_path_guiding_lab_amd/csrc -j8 EXTRA="-DPG_ABLATE_$v" > $OUT/make.log 2>&1
	run "no_$(echo $v | tr -d ' ' | tr -d '-')"
done
