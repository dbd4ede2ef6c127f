#!/bin/bash
# Build libsfx.so for gfx950 (cross-compiles without a GPU).  Usage: build.sh [outdir]
#   SFX_LAB=1 build.sh   builds libsfx_lab.so instead: the same sources with -DSFX_LAB -- the A/B forms, environment switches and
#                        phase clocks of include/sfx_lab.h (tools/, tools/run_gpu_suite.sh); the product library has none of them.
set -e
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
OUT="${1:-$HERE/..}"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
NAME=libsfx.so; OBJ="$HERE/obj"
if [ -n "$SFX_LAB" ]; then NAME=libsfx_lab.so; OBJ="$HERE/obj_lab"; SFX_DEFINES="$SFX_DEFINES -DSFX_LAB"; fi
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result $SFX_DEFINES"   # SFX_DEFINES: diagnostic -D switches (tools/)
mkdir -p "$OBJ"
pids=()
$HIPCC $FLAGS -c "$HERE/api.hip" -o "$OBJ/api.o" & pids+=($!)
$HIPCC $FLAGS -c "$HERE/closure.hip" -o "$OBJ/closure.o" & pids+=($!)
$HIPCC $FLAGS -c "$HERE/lbs_dense.hip" -o "$OBJ/lbs_dense.o" & pids+=($!)
$HIPCC $FLAGS -ffp-contract=off -c "$HERE/lbfgs.hip" -o "$OBJ/lbfgs.o" & pids+=($!)
$HIPCC $FLAGS -c "$HERE/fused.hip" -o "$OBJ/fused.o" & pids+=($!)
$HIPCC $FLAGS -ffp-contract=off -c "$HERE/collide.hip" -o "$OBJ/collide.o" & pids+=($!)
$HIPCC $FLAGS -c "$HERE/lbs_adjoint.hip" -o "$OBJ/lbs_adjoint.o" & pids+=($!)
for p in "${pids[@]}"; do wait $p; done
$HIPCC --offload-arch=gfx950 -shared -fPIC -o "$OUT/$NAME" "$OBJ/api.o" "$OBJ/closure.o" \
    "$OBJ/lbs_dense.o" "$OBJ/lbfgs.o" "$OBJ/fused.o" "$OBJ/collide.o" "$OBJ/lbs_adjoint.o"
echo "built $OUT/$NAME"
