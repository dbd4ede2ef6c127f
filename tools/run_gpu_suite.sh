#!/bin/bash
# The GPU suite on BOTH builds of the library (GPU box):
#   1. product  libsfx.so      -- what ships; the tests of interchangeable forms / environment switches skip
#   2. lab      libsfx_lab.so  -- the same source with -DSFX_LAB (include/sfx_lab.h): those tests run, everything else runs again
# usage: tools/run_gpu_suite.sh [pytest args]      (both libraries must have been built: python __graft_entry__.py)
set -u
cd "$(dirname "$0")/.."
echo "== product build (libsfx.so)"
python -m pytest tests -m gpu -q "$@"; rc1=$?
echo "== lab build (libsfx_lab.so)"
SFX_LIB="$PWD/smplify-x-partial_amd/libsfx_lab.so" python -m pytest tests -m gpu -q "$@"; rc2=$?
echo "product rc=$rc1 lab rc=$rc2"
exit $(( rc1 != 0 || rc2 != 0 ))
