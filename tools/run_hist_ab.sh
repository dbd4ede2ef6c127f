# A/B of the L-BFGS history's dead lanes reading a page of zeros: same bits, a third of the history's bytes gone
cd $GRAFT_REPO_ROOT
run() { tag=$1; shift; env "$@" python bench.py --steps 20 --warmup 2 --no-configs3 > gpurun_out/h_$tag.json 2> gpurun_out/h_$tag.err; python -c "
import json; d=json.load(open('gpurun_out/h_$tag.json')); t=d['roofline_tick']; print('$tag', d['value'], d['ms_per_step'], 'tick us', t['avg_launch_us'], 'traffic', t.get('traffic'), d['host']['loop_us_per_round'], d["config"]["closure_evals_per_frame_mean"], d["config"]["final_loss_mean"])"; }
run a0 SFX_HIST_ZEROPAGE=0
run z0 SFX_HIST_ZEROPAGE=1
run a1 SFX_HIST_ZEROPAGE=0
run z1 SFX_HIST_ZEROPAGE=1
SFX_HIST_ZEROPAGE=1 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -2
