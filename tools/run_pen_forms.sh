cd $GRAFT_REPO_ROOT
run() { tag=$1; shift; env "$@" python bench.py --workload pen --steps 3 --warmup 1 > gpurun_out/b_$tag.json 2> gpurun_out/b_$tag.err; python -c "
import json; d=json.load(open('gpurun_out/b_$tag.json')); print('$tag', d['value'], d['ms_per_step'], d['config'].get('non_finite'), d['config']['closure_evals_per_frame_mean'], d['config']['closure_evals_per_frame_max'], d['roofline_pen']['avg_launch_us'], d['host']['loop_us_per_round'])"; }
run form0 SFX_PEN_FORM=0
run form0_norewalk SFX_PEN_FORM=0 SFX_PEN_REWALK_OFF=1
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
SFX_PEN_REWALK_OFF=1 SFX_PEN_FORM=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt_pen -o p -- python bench.py --workload pen --steps 2 --warmup 1 --no-parity > gpurun_out/b_pen_prof.json 2> gpurun_out/kt.log
for f in $(find gpurun_out/kt_pen -name "*kernel_trace.csv"); do python tools/kt_percentiles.py $f > gpurun_out/pen_form0_norewalk_percentiles.txt; done
find gpurun_out -name "*kernel_trace.csv" -delete; find gpurun_out -name "*.db" -delete; head -16 gpurun_out/pen_form0_norewalk_percentiles.txt
