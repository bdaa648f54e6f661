#!/bin/bash
# Dev tool: builds a VARIANT of libpgsd.so next to the default one, from the same sources with extra
# compiler flags, into its own object directory -- the default build is not touched.
#   bash tools/build_variant.sh alt "-DPG_SOME_SWITCH=1"      -> practical_path_guiding_lab_amd/libpgsd_alt.so
# Select it at run time with PGSD_LIBRARY=practical_path_guiding_lab_amd/libpgsd_alt.so (see _native.py).
set -e
NAME=$1; shift
EXTRA="$@"
R=$(cd "$(dirname "$0")/.." && pwd)
SRC=$R/practical_path_guiding_lab_amd/csrc
OBJ=/tmp/pgsd_variant_$NAME
mkdir -p $OBJ
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt"
pids=()
for f in $SRC/*.hip; do
	/opt/rocm/bin/hipcc $FLAGS $EXTRA -c $f -o $OBJ/$(basename $f .hip).o &
	pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/practical_path_guiding_lab_amd/libpgsd_$NAME.so $OBJ/*.o
echo built $R/practical_path_guiding_lab_amd/libpgsd_$NAME.so
