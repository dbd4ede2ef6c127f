mkdir -p gpurun_out/r3
for v in base v5 v6 v7; do
  if [ $v == base ]; then L=libsfx.so; else L=libsfx_$v.so; fi
  SFX_LIB=$PWD/smplify-x-partial_amd/$L python bench.py --workload pen --no-cpu --no-parity --no-side --no-alt > gpurun_out/r3/pen_ab_$v.json 2>/dev/null
  python - <<PY
import json
d=json.loads([l for l in open("gpurun_out/r3/pen_ab_$v.json") if l.startswith("{")][0])
print("$v", round(d["value"],1), round(d["roofline_pen"]["avg_scope_us"],1), d["config"]["closure_evals_per_frame_max"], d["config"]["final_loss_mean"])
PY
done
