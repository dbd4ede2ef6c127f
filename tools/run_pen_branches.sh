# concurrent branches of the interpenetration step (SFX_PEN_BRANCHES): tests at the default, then the halpe bench at 1 / 2 / 3 / 4 branches
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_topology.py tests/test_gpu_penetration.py -x -q -m gpu 2>&1 | tail -2
run() { tag=$1; shift; env "$@" timeout 900 python bench.py --workload pen --steps 3 --warmup 1 > gpurun_out/b_$tag.json 2> gpurun_out/b_$tag.err; python -c "
import json; d=json.load(open('gpurun_out/b_$tag.json')); print('$tag', d['value'], d['ms_per_step'], d['config'].get('non_finite'), d['config']['closure_evals_per_frame_mean'], d['config']['final_loss_mean'], 'scope us', d['roofline_pen']['avg_launch_us'], 'loop', d['host']['loop_us_per_round'])"; }
for n in 1 2 3 4 1 2; do run br$n SFX_PEN_BRANCHES=$n; done
