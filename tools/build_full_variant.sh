#!/bin/bash
# build_full_variant.sh TAG [hipcc flags...]: libsfx_TAG.so with EVERY csrc file compiled under the extra flags
set -e
HERE=/root/repo/smplify-x-partial_amd/csrc; TAG=$1; shift
mkdir -p /tmp/v_$TAG
pids=()
for f in api closure lbs_dense fused collide lbs_adjoint; do /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result "$@" -c $HERE/$f.hip -o /tmp/v_$TAG/$f.o 2>/dev/null & pids+=($!); done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -ffp-contract=off "$@" -c $HERE/lbfgs.hip -o /tmp/v_$TAG/lbfgs.o 2>/dev/null & pids+=($!)
for p in "${pids[@]}"; do wait $p; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /root/repo/smplify-x-partial_amd/libsfx_$TAG.so /tmp/v_$TAG/*.o
echo built $TAG
