# Headline bench under several values of one environment variable: scripts/ab_env.sh NAME v1 v2 ... (round the list twice)
cd "$GRAFT_REPO_ROOT"
name=$1; shift
for round in 1 2; do
for v in "$@"; do
  env $name=$v timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-ransac --no-pipeline --no-extensions --no-regimes > gpurun_out/abenv_$v.json 2>gpurun_out/abenv_$v.err || { tail -3 gpurun_out/abenv_$v.err; exit 1; }
  python3 -c "
import json;d=json.load(open('gpurun_out/abenv_$v.json'));print('$name=$v', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['config']['matches_rank0'])"
done
done
