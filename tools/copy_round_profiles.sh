#!/bin/bash
# After `gpurun -- bash tools/run_round_profiles.sh rNN`: copy the merged gpurun_out/ summaries into profiles/ under the round's names
# (kernel stats, percentiles, PMC summaries, bench lines under the profiler and of record, detail files) and refresh the summaries
# bench.py replays (profiles/pmc_summary*.json, kernel_stats*.csv: same csrc hash as the build).   usage: tools/copy_round_profiles.sh r04
R=${1:?round tag}; cd "$(dirname "$0")/.."; G=gpurun_out
for w in body full pen rows; do
  cp $G/prof_${R}_$w/kt/p_kernel_stats.csv profiles/${R}_${w}_kernel_stats.csv
  cp $G/prof_${R}_$w/kernel_percentiles.txt profiles/${R}_${w}_kernel_percentiles.txt
  cp $G/prof_${R}_$w/pmc_summary.json profiles/${R}_pmc_summary_$w.json
  cp $G/prof_${R}_$w/bench_under_rocprof.json profiles/${R}_bench_${w}_under_rocprof.json
  s=""; [ $w != body ] && s="_$w"
  cp $G/prof_${R}_$w/pmc_summary.json profiles/pmc_summary$s.json
  cp $G/prof_${R}_$w/kt/p_kernel_stats.csv profiles/kernel_stats$s.csv
done
for t in default full pen rows 1024; do cp $G/${R}_bench_$t.json profiles/${R}_bench_$t.json; done
for t in body full pen rows; do cp $G/${R}_detail_$t.json profiles/${R}_detail_$t.json; done
