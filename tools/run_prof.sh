#!/bin/bash
# Profiling passes of a bench command on the GPU box (run through gpurun).
#   tools/run_prof.sh TAG [bench.py flags ...]      e.g.  tools/run_prof.sh r03_full --workload full
# Raw traces stay in /tmp on the box; the summaries land in gpurun_out/prof_TAG/ and are copied into profiles/ by hand
# (kernel stats CSV -> profiles/kernel_stats[_full|_pen|_rows].csv, bench line under rocprof, pmc_summary.json -> profiles/pmc_summary[...].json:
# bench.py replays both when the summary's source hash equals the build's).  Counter passes are separate runs (kernel trace + --pmc
# only: gpurun refuses --pmc together with the sys / runtime trace domains).  pmc_summary.json records the hash of the
# csrc sources the counters were taken with (tools/pmc_summary.py csrc_sha): bench.py replays the traffic figures only
# when that hash equals the running build's.
set -x
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
TAG=$1; shift
O=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG; mkdir -p $O
CMD="python bench.py --steps 2 --warmup 1 --no-cpu --no-parity --no-alt --no-configs3 $*"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o p -- $CMD > $O/bench_under_rocprof.json 2> $O/kt.log
PM="python bench.py --steps 1 --warmup 0 --no-cpu --no-alt --no-parity --no-configs3 $*"
rm -rf /tmp/pmc_f /tmp/pmc_w /tmp/pmc_s /tmp/pmc_c
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/pmc_f -o p -- $PM > /dev/null 2> $O/pmc_f.log
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/pmc_w -o p -- $PM > /dev/null 2> $O/pmc_w.log
timeout 600 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d /tmp/pmc_c -o p -- $PM > /dev/null 2> $O/pmc_c.log
timeout 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d /tmp/pmc_s -o p -- $PM > /dev/null 2> $O/pmc_s.log
python tools/pmc_summary.py $O/pmc_summary.json /tmp/pmc_f /tmp/pmc_w /tmp/pmc_c /tmp/pmc_s > $O/pmc_brief.json
for f in $(find $O/kt -name "*kernel_trace.csv"); do python tools/kt_percentiles.py $f > $O/kernel_percentiles.txt; done
find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete
tail -c 400 $O/kt.log; head -12 $O/kt/p_kernel_stats.csv; tail -2 $O/pmc_c.log; cat $O/pmc_brief.json | head -c 3000
