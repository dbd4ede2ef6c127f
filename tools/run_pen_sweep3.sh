# rows of the launches that loop over the wanted columns: sweep inside one gpurun call (builds on the GPU box)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
one() { tag=$1; shift
  SFX_DEFINES="$*" bash smplify-x-partial_amd/csrc/build.sh > /dev/null 2>&1
  (cd /tmp; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/kt_$tag -o p -- python $GRAFT_REPO_ROOT/bench.py --workload pen --steps 2 --warmup 1 --no-parity > $GRAFT_REPO_ROOT/gpurun_out/b_$tag.json 2> $GRAFT_REPO_ROOT/gpurun_out/kt_$tag.log)
  for f in $(find gpurun_out/kt_$tag -name "*kernel_trace.csv"); do python tools/kt_percentiles.py $f > gpurun_out/pct_$tag.txt; done
  find gpurun_out -name "*kernel_trace.csv" -delete; find gpurun_out -name "*.db" -delete
  echo "== $tag ($*)"; python -c "
import json; d=json.load(open('gpurun_out/b_$tag.json')); print(d['value'], d['roofline_pen']['avg_launch_us'])"
  grep -E "k_pen_rank|k_pen_walk |k_pen_facesum|k_pen_gather" gpurun_out/pct_$tag.txt | cut -c1-110
}
one r64
one r32 -DPEN_SEL_ROWS=32
one r128 -DPEN_SEL_ROWS=128
one r64b
