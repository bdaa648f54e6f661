#!/bin/bash
# Dev experiment (run on the GPU box from the repo root): bench.py for several spp-per-pass values
# and k_bounce occupancy targets.  Rebuilds pg_render.o on the box for each occupancy variant.
set -e
OUT=gpurun_out/exp_occ
mkdir -p $OUT
for spp in 8 16 32; do
	python bench.py --cpu-res 0 --spp-per-pass $spp --steps 10 > $OUT/spp$spp.json
	python - <<EOF
import json
d = json.load(open("$OUT/spp$spp.json"))
print("spp $spp value", d["value"], "ms", d["ms_per_step"], "bounce", d["kernels"]["k_bounce"]["avg_us"], "splat", d["kernels"]["k_process_and_splat"]["avg_us"])
EOF
done
for w in 5 6; do
	touch practical_path_guiding_lab_amd/csrc/pg_render.hip
	make -s -C practical_path_guiding_lab_amd/csrc EXTRA=-DPG_BOUNCE_WAVES_L2=$w > $OUT/make_w$w.log 2>&1   # level-2 kernels only
	python bench.py --scene torus --cpu-res 0 --steps 10 > $OUT/w$w.json
	python - <<EOF
import json
d = json.load(open("$OUT/w$w.json"))
print("waves $w value", d["value"], "ms", d["ms_per_step"], "bounce", d["kernels"]["k_bounce"]["avg_us"], "splat", d["kernels"]["k_process_and_splat"]["avg_us"])
EOF
done
