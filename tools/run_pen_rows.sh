# rows of the interpenetration launches over the LIST of wanted columns (default) against a row per active column (SFX_PEN_ROWS_OFF=1)
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_topology.py tests/test_gpu_penetration.py -x -q -m gpu 2>&1 | tail -2
run() { tag=$1; shift; env "$@" timeout 900 python bench.py --workload pen --steps 3 --warmup 1 > gpurun_out/b_$tag.json 2> gpurun_out/b_$tag.err; python -c "
import json; d=json.load(open('gpurun_out/b_$tag.json')); print('$tag', d['value'], d['ms_per_step'], d['config'].get('non_finite'), d['config']['closure_evals_per_frame_mean'], d['config']['final_loss_mean'], 'scope us', d['roofline_pen']['avg_launch_us'], 'loop', d['host']['loop_us_per_round'])"; }
run list A=1
run rows SFX_PEN_ROWS_OFF=1
run list2 A=1
run rows2 SFX_PEN_ROWS_OFF=1
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt_pen -o p -- python bench.py --workload pen --steps 2 --warmup 1 --no-parity > gpurun_out/b_pen_prof.json 2> gpurun_out/kt.log
for f in $(find gpurun_out/kt_pen -name "*kernel_trace.csv"); do python tools/kt_percentiles.py $f > gpurun_out/pen_form0_percentiles.txt; done
find gpurun_out -name "*kernel_trace.csv" -delete; find gpurun_out -name "*.db" -delete; head -16 gpurun_out/pen_form0_percentiles.txt
