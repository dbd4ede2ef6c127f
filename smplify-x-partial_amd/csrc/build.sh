#!/bin/bash
# Build libsfx.so for gfx950 (cross-compiles without a GPU).  Usage: build.sh [outdir]
set -e
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
OUT="${1:-$HERE/..}"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result $SFX_DEFINES"   # SFX_DEFINES: diagnostic -D switches (tools/)
mkdir -p "$HERE/obj"
pids=()
$HIPCC $FLAGS -c "$HERE/api.hip" -o "$HERE/obj/api.o" & pids+=($!)
$HIPCC $FLAGS -c "$HERE/closure.hip" -o "$HERE/obj/closure.o" & pids+=($!)
$HIPCC $FLAGS -c "$HERE/lbs_dense.hip" -o "$HERE/obj/lbs_dense.o" & pids+=($!)
$HIPCC $FLAGS -ffp-contract=off -c "$HERE/lbfgs.hip" -o "$HERE/obj/lbfgs.o" & pids+=($!)
$HIPCC $FLAGS -c "$HERE/fused.hip" -o "$HERE/obj/fused.o" & pids+=($!)
$HIPCC $FLAGS -ffp-contract=off -c "$HERE/collide.hip" -o "$HERE/obj/collide.o" & pids+=($!)
$HIPCC $FLAGS -c "$HERE/lbs_adjoint.hip" -o "$HERE/obj/lbs_adjoint.o" & pids+=($!)
for p in "${pids[@]}"; do wait $p; done
$HIPCC --offload-arch=gfx950 -shared -fPIC -o "$OUT/libsfx.so" "$HERE/obj/api.o" "$HERE/obj/closure.o" \
    "$HERE/obj/lbs_dense.o" "$HERE/obj/lbfgs.o" "$HERE/obj/fused.o" "$HERE/obj/collide.o" "$HERE/obj/lbs_adjoint.o"
echo "built $OUT/libsfx.so"
