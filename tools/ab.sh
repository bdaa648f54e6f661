#!/bin/bash
# Dev tool (GPU box, repo root): the default library and variants built by tools/build_variant.sh, interleaved on one box.
#   bash tools/ab.sh OUTNAME "bench args" default w6 w7 ...
OUT=gpurun_out/$1; shift
ARGS=$1; shift
mkdir -p $OUT
for rep in 1 2; do
for v in "$@"; do
	if [ "$v" = default ]; then unset PGSD_LIBRARY; else export PGSD_LIBRARY=$PWD/practical_path_guiding_lab_amd/libpgsd_$v.so; fi
	python bench.py --cpu 0 --full-schedule 0 --spp1 0 --other-configs 0 $ARGS --detail $OUT/$v.$rep.json > $OUT/$v.$rep.line 2> $OUT/$v.$rep.err || exit 1
	python - <<PY
import json
d = json.load(open("$OUT/$v.$rep.json"))
k = d["kernels"]
print("%-10s rep $rep value %7.1f  ms %.3f  " % ("$v", d["value"], d["ms_per_step"]) + "  ".join("%s %.2f" % (n.replace("k_wave_", "").replace("k_process_and_", ""), x["ms_per_step"]) for n, x in k.items()) + "  a/b %s/%s" % (k.get("k_wave_shade_a+b", {}).get("shade_a_ms_per_step"), k.get("k_wave_shade_a+b", {}).get("shade_b_ms_per_step")))
PY
done
done
