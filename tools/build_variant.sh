#!/bin/bash
# build_variant.sh TAG [hipcc flags...]: libsfx_TAG.so with lbs_dense.hip compiled under extra flags (tuning experiments)
set -e
HERE=/root/repo/smplify-x-partial_amd/csrc; TAG=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC "$@" -c $HERE/lbs_dense.hip -o /tmp/lbs_dense_$TAG.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /root/repo/smplify-x-partial_amd/libsfx_$TAG.so $HERE/obj/api.o $HERE/obj/closure.o /tmp/lbs_dense_$TAG.o $HERE/obj/lbfgs.o $HERE/obj/fused.o
echo built $TAG
