#!/bin/bash
# Dev experiment (GPU box, repo root): workgroups per CU of the grid-stride splat kernels (k_splat_list, k_process_and_splat).
# Rebuilds the library per value and prints the bench's kernel times for the three scenes.
# whatever happens, leave the DEFAULT build behind: variant objects are newer than the sources, so a later
# `make` (or __graft_entry__.build()) would otherwise keep shipping the experiment
trap 'touch practical_path_guiding_lab_amd/csrc/*.hip; make -s -C practical_path_guiding_lab_amd/csrc -j8' EXIT
set -e
OUT=gpurun_out/exp_splat_grid
mkdir -p $OUT
for g in ${@:-8 16 32 64 128}; do
	touch practical_path_guiding_lab_amd/csrc/pg_kernels_splat.hip
	make -s -C practical_path_guiding_lab_amd/csrc -j8 EXTRA="-DPG_SPLAT_GROUPS_PER_CU=$g" > $OUT/make.log 2>&1
	for s in cornell-box veach-mis torus; do
		python bench.py --scene $s --cpu 0 --steps 10 --detail $OUT/$s.$g.json > $OUT/$s.$g.line
		python - <<EOF
import json
d = json.load(open("$OUT/$s.$g.json"))
print("groups/CU %4d  %-12s value %7.1f  ms %.3f  bounce %.1f us  splat %.1f us" % ($g, "$s", d["value"], d["ms_per_step"], d["kernels"]["k_bounce"]["avg_us"], d["kernels"]["k_splat_list"]["avg_us"]))
EOF
	done
done
