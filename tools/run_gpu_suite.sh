# the whole GPU suite, then the default bench line and the configs[3] single-GPU job (the two-workgroups-per-CU tick kernel)
cd $GRAFT_REPO_ROOT
python -m pytest tests -x -q -m gpu 2>&1 | tail -4
python bench.py --steps 10 --warmup 2 > gpurun_out/s_default.json 2> gpurun_out/s_default.err
python -c "
import json; d=json.load(open('gpurun_out/s_default.json')); print('default', d['value'], d['roofline_tick']['avg_launch_us'], d['roofline']['avg_launch_us'], d['config'].get('configs3_single_gpu_frames_per_s'))"
