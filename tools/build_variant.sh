#!/bin/bash
# build_variant.sh TAG FILE [hipcc flags...]: libsfx_TAG.so with csrc/FILE.hip compiled under extra flags
# (tuning / diagnostic experiments, e.g. `build_variant.sh count collide -DPEN_COUNT`); load it with SFX_LIB=libsfx_TAG.so
set -e
HERE=/root/repo/smplify-x-partial_amd/csrc; TAG=$1; FILE=$2; shift; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result "$@" -c $HERE/$FILE.hip -o /tmp/${FILE}_$TAG.o
case " api closure lbs_dense lbfgs fused collide lbs_adjoint " in *" $FILE "*) ;; *) echo "unknown csrc file $FILE"; exit 1;; esac
OBJS=""
for f in api closure lbs_dense lbfgs fused collide lbs_adjoint; do
  if [ "$f" == "$FILE" ]; then OBJS="$OBJS /tmp/${FILE}_$TAG.o"; else OBJS="$OBJS $HERE/obj/$f.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /root/repo/smplify-x-partial_amd/libsfx_$TAG.so $OBJS
echo built $TAG
