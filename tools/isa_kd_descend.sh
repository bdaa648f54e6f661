#!/bin/bash
# ISA of pg::k_process_and_splat (the kernel round 2's fault was seen in; its list variant <true> was removed later in
# round 3, when the renderer's record list began to name accumulators: the dense variant inlines the same kd_descend_grid) for the excerpts under
# profiles/r03/kd_descend_isa/: the kernel's listing goes to stdout.
#   tools/isa_kd_descend.sh > /tmp/k_process_and_splat_list.s
set -e
SRC="$(cd "$(dirname "$0")/.." && pwd)/practical_path_guiding_lab_amd/csrc"
OUT="$(mktemp -d)"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math \
    -fhip-fp32-correctly-rounded-divide-sqrt --cuda-device-only -S "$SRC/pg_kernels_splat.hip" -o "$OUT/splat.s" 2>/dev/null
s=$(grep -n "^_ZN2pg19k_process_and_splatE" "$OUT/splat.s" | cut -d: -f1)
e=$(awk -v s="$s" 'NR>s && /^\.Lfunc_end/{print NR; exit}' "$OUT/splat.s")
sed -n "${s},${e}p" "$OUT/splat.s"
rm -rf "$OUT"
