cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py -q -m gpu -k "pool_queue or one_gpu_share" 2>&1 | tail -3
run() { tag=$1; shift; python bench.py --no-cpu --no-alt --no-side --no-parity --no-configs3 --steps 3 --warmup 1 "$@" > gpurun_out/b_$tag.json 2> gpurun_out/b_$tag.err; python -c "
import json; d=json.load(open('gpurun_out/b_$tag.json')); print('$tag', d['value'], d['ms_per_step'], d['config']['gemm_columns_per_gpu'], d['config']['closure_evals_per_frame_max'])"; tail -2 gpurun_out/b_$tag.err; }
run f1024_auto --frames 1024
run f1024_resident --frames 1024 --slots 0
run f1024_s256 --frames 1024 --slots 256
run f1024_s384 --frames 1024 --slots 384
run f2048_auto --frames 2048
python bench.py --steps 5 --warmup 1 > gpurun_out/b_default.json 2> gpurun_out/b_default.err; tail -c 1500 gpurun_out/b_default.json
