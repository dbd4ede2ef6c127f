cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_topology.py tests/test_gpu_penetration.py -m gpu -x -q -k "pen_set or reference_lines" 2>&1 | grep -v Warning | tail -25
python tools/pen_orient_probe.py 92 2>&1 | grep -v Warning | tail -20
python bench.py --workload pen --steps 3 --no-cpu > gpurun_out/r06_bench_pen_parity.json 2>/dev/null; python -c "
import json; d=json.loads(open('gpurun_out/r06_bench_pen_parity.json').read().strip().splitlines()[-1]); print(d['value'], d.get('reference_parity'))"
