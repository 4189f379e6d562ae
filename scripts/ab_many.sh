# Headline bench of several builds of the library on one box (sfm_amd/lib/ab/<name>.so), round the list twice:
#   scripts/ab_many.sh <name> <name> ...
cd "$GRAFT_REPO_ROOT"
for round in 1 2; do
for v in "$@"; do
  SFM_AMD_LIB=$PWD/sfm_amd/lib/ab/$v.so timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-ransac --no-pipeline --no-extensions --no-regimes > gpurun_out/ab_$v.json 2>gpurun_out/ab_$v.err || { tail -3 gpurun_out/ab_$v.err; exit 1; }
  python3 -c "
import json;d=json.load(open('gpurun_out/ab_$v.json'));print('$v', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['config']['matches_rank0'])"
done
done
