set -x
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r1c; mkdir -p $O
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_dense -o p -- python bench.py --steps 2 --warmup 1 --no-cpu > $O/bench_dense_prof.json 2> $O/kt_dense.log
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_rows -o p -- python bench.py --steps 2 --warmup 1 --no-cpu --lbs rows > $O/bench_rows_prof.json 2> $O/kt_rows.log
timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/pmc_f -o p -- python bench.py --steps 1 --warmup 0 --no-cpu > /dev/null 2> $O/pmc_f.log
timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/pmc_w -o p -- python bench.py --steps 1 --warmup 0 --no-cpu > /dev/null 2> $O/pmc_w.log
python tools/pmc_summary.py $O/pmc_summary.json /tmp/pmc_f /tmp/pmc_w
rm -f $O/*/p_kernel_trace.csv $O/*/*.db
ls -la $O $O/*
