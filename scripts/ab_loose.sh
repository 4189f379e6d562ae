# Kernel statistics at the reference's default threshold for several builds (sfm_amd/lib/ab/<name>.so)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for v in "$@"; do
  rm -rf gpurun_out/trace_loose_$v
  SFM_AMD_LIB=$PWD/sfm_amd/lib/ab/$v.so timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/trace_loose_$v -- python3 bench.py --images 24 --threshold 0.5 --steps 2 --warmup 1 --no-cpu-baseline --no-ransac --no-pipeline --no-extensions --no-regimes > gpurun_out/trace_loose_$v.json 2>gpurun_out/trace_loose_$v.err || { echo "$v failed"; tail -3 gpurun_out/trace_loose_$v.err; exit 1; }
  echo "== $v"
  python3 -c "
import json;d=json.load(open('gpurun_out/trace_loose_$v.json'));print(d['value'], d['ms_per_step'], d['config']['matches_rank0'], d['config']['candidates_rank0'])"
  python3 scripts/trace_summary.py gpurun_out/trace_loose_$v 6
done
