#!/bin/bash
# Profiling passes of the bench command on the GPU box (run through gpurun).  Raw traces stay in
# /tmp on the box; the summaries land in gpurun_out/r2prof/ and are copied into profiles/ by hand.
set -x
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r2prof; mkdir -p $O
CMD="python bench.py --steps 2 --warmup 1 --no-cpu --no-parity"
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o p -- $CMD > $O/bench_under_rocprof.json 2> $O/kt.log
PM="python bench.py --steps 1 --warmup 0 --no-cpu --no-alt --no-parity"
timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/pmc_f -o p -- $PM > /dev/null 2> $O/pmc_f.log
timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/pmc_w -o p -- $PM > /dev/null 2> $O/pmc_w.log
timeout 400 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d /tmp/pmc_s -o p -- $PM > /dev/null 2> $O/pmc_s.log
python tools/pmc_summary.py $O/pmc_summary.json /tmp/pmc_f /tmp/pmc_w /tmp/pmc_s
# the needed-rows path (persistent kernel): kernel trace only
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_rows -o p -- python bench.py --steps 2 --warmup 1 --no-cpu --no-parity --lbs rows > $O/bench_rows_under_rocprof.json 2> $O/kt_rows.log
rm -f $O/*/p_kernel_trace.csv $O/*/*.db
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err
tail -c 600 $O/kt.log; cat $O/kt/p_kernel_stats.csv | head -5; tail -3 $O/pmc_s.log; tail -c 3000 $O/bench_default.json
