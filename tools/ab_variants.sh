#!/bin/bash
# usage: ab.sh TAG...   (runs bench for default lib and each libsfx_TAG.so, twice, alternating)
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for tag in base "$@"; do
  if [ $tag == base ]; then unset SFX_LIB; else export SFX_LIB=$GRAFT_REPO_ROOT/smplify-x-partial_amd/libsfx_$tag.so; fi
  timeout 240 python bench.py --steps 4 --warmup 1 --no-cpu --no-parity --no-alt $ABFLAGS > /tmp/ab_$tag.json 2> /tmp/ab_$tag.err
  python - <<P
import json; d=json.load(open("/tmp/ab_$tag.json")); print("$tag", d["value"], d["config"]["final_loss_mean"], d["config"]["closure_evals_per_frame_mean"], "gemm", d["roofline"]["avg_launch_us"], "tick", d["roofline_tick"]["avg_launch_us"])
P
done; done
