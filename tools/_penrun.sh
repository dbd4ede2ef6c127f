cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/kt5
timeout 250 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt5 -o p -- python $GRAFT_REPO_ROOT/bench.py --workload pen --steps 2 --warmup 1 --no-cpu --no-parity --no-alt --no-side 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'): d=json.loads(l); print('under rocprof', d['value'])"
cd $GRAFT_REPO_ROOT; python tools/kt_percentiles.py $(find /tmp/kt5 -name "*kernel_trace.csv") | head -${1:-8}
for i in 1 2; do timeout 150 python bench.py --workload pen --no-cpu --no-parity --no-side --no-alt 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'): d=json.loads(l); print('bench', d['value'], d['roofline_pen']['avg_scope_us'], d['config']['final_loss_mean'])"; done
