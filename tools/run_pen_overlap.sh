cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for n in 1 2; do
SFX_PEN_BRANCHES=$n timeout 600 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/kt_br$n -o p -- python bench.py --workload pen --steps 1 --warmup 1 --no-parity --no-cpu > gpurun_out/b_ovl$n.json 2> gpurun_out/kt_br$n.log
for f in $(find gpurun_out/kt_br$n -name "*kernel_trace.csv"); do echo branches $n; python tools/kt_overlap.py $f; python tools/kt_percentiles.py $f | head -16; done
find gpurun_out -name "*kernel_trace.csv" -delete; find gpurun_out -name "*.db" -delete
done
