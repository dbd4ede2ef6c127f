#!/bin/bash
# The round's profiles and bench lines of record in one gpurun call:  tools/run_round_profiles.sh r04
# (tools/run_prof.sh per workload, then the un-profiled bench lines; everything lands in gpurun_out/, copied to profiles/ by hand)
R=${1:-rXX}
cd $GRAFT_REPO_ROOT
bash tools/run_prof.sh ${R}_body > gpurun_out/prof_${R}_body.log 2>&1
bash tools/run_prof.sh ${R}_full --workload full > gpurun_out/prof_${R}_full.log 2>&1
bash tools/run_prof.sh ${R}_pen --workload pen > gpurun_out/prof_${R}_pen.log 2>&1
bash tools/run_prof.sh ${R}_rows --lbs rows > gpurun_out/prof_${R}_rows.log 2>&1
# the summaries the bench lines replay (same source hash)
for w in body full pen rows; do
  s=""; [ $w != body ] && s="_$w"
  cp gpurun_out/prof_${R}_$w/pmc_summary.json profiles/pmc_summary$s.json
  cp gpurun_out/prof_${R}_$w/kt/p_kernel_stats.csv profiles/kernel_stats$s.csv
done
line() { tag=$1; shift; timeout 900 python bench.py "$@" > gpurun_out/${R}_bench_$tag.json 2> gpurun_out/${R}_bench_$tag.err; tail -c 300 gpurun_out/${R}_bench_$tag.err; head -c 400 gpurun_out/${R}_bench_$tag.json; echo; }
line default
cp gpurun_out/bench_detail_body.json gpurun_out/${R}_detail_body.json
line full --workload full --steps 3
cp gpurun_out/bench_detail_full.json gpurun_out/${R}_detail_full.json
line pen --workload pen --steps 3
cp gpurun_out/bench_detail_pen.json gpurun_out/${R}_detail_pen.json
line rows --lbs rows --steps 3
cp gpurun_out/bench_detail_body_rows.json gpurun_out/${R}_detail_rows.json
line 1024 --frames 1024 --steps 2 --no-cpu
cp profiles/pmc_summary*.json profiles/kernel_stats*.csv gpurun_out/ 2>/dev/null
