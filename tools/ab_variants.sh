#!/bin/bash
# A/B of builds or switches inside ONE gpurun call (box-to-box differences are as large as most changes):
#   tools/ab_variants.sh TAG ...     TAG = name of a smplify-x-partial_amd/libsfx_TAG.so (tools/build_variant.sh, or a copy of an
#                                    older build), or env:NAME=VALUE (a measurement switch of the default library)
# Runs the default library and every TAG twice, alternating; ABFLAGS = extra bench.py flags.
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for tag in base "$@"; do
  unset SFX_LIB; EV=""
  case $tag in
    base) ;;
    env:*) EV="${tag#env:}" ;;
    *) export SFX_LIB=$GRAFT_REPO_ROOT/smplify-x-partial_amd/libsfx_$tag.so ;;
  esac
  env $EV timeout 240 python bench.py --steps 4 --warmup 1 --no-cpu --no-parity --no-alt $ABFLAGS > /tmp/ab.json 2> /tmp/ab.err
  python - <<P
import json; d=json.load(open("/tmp/ab.json")); print("$tag", d["value"], d["config"]["final_loss_mean"], d["config"]["closure_evals_per_frame_mean"], "gemm", d["roofline"]["avg_launch_us"], "tick", d["roofline_tick"]["avg_launch_us"])
P
done; done
