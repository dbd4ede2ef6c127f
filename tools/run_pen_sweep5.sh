# workgroups of the flat kernels k_pen_walk2 / k_pen_eval (PEN_FLAT_BLOCKS) after k_pen_walk / k_pen_rank went flat: sweep in one gpurun call
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
one() { tag=$1; shift
  SFX_DEFINES="$*" bash smplify-x-partial_amd/csrc/build.sh > /dev/null 2>&1
  (cd /tmp; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/kt_$tag -o p -- python $GRAFT_REPO_ROOT/bench.py --workload pen --steps 2 --warmup 1 --no-parity > $GRAFT_REPO_ROOT/gpurun_out/b_$tag.json 2> $GRAFT_REPO_ROOT/gpurun_out/kt_$tag.log)
  for f in $(find gpurun_out/kt_$tag -name "*kernel_trace.csv"); do python tools/kt_percentiles.py $f > gpurun_out/pct_$tag.txt; done
  find gpurun_out -name "*kernel_trace.csv" -delete; find gpurun_out -name "*.db" -delete
  echo "== $tag ($*)"; python -c "
import json; d=json.load(open('gpurun_out/b_$tag.json')); print(d['value'], d['roofline_pen']['avg_launch_us'])"
  grep -E "k_pen_walk2|k_pen_eval|k_pen_rank|k_pen_walk " gpurun_out/pct_$tag.txt | cut -c1-110
}
one base
one f1024 -DPEN_FLAT_BLOCKS=1024
one f512 -DPEN_FLAT_BLOCKS=512
one rw256 -DPEN_RANK_FLAT=256 -DPEN_WALK_FLAT=256
one rw1024 -DPEN_RANK_FLAT=1024 -DPEN_WALK_FLAT=1024
