cd $GRAFT_REPO_ROOT
for z in 0 1 2; do SFX_HIST_ZEROPAGE=$z python bench.py --steps 3 --warmup 1 --no-configs3 --no-parity > gpurun_out/h2_$z.json 2> gpurun_out/h2_$z.err; python -c "
import json; d=json.load(open('gpurun_out/h2_$z.json')); print($z, d['value'], d['config']['closure_evals_per_frame_mean'], d['config']['final_loss_mean'])"; done
