# Similarity kernel: the shipped instance (1) beside its timing-only ablations (4 no test, 5 no test and no staging,
# 6 candidates not stored, 7 hits not queued), one bench run each on one box
cd "$GRAFT_REPO_ROOT"
for v in ${@:-1 6 7 4 1}; do
  SFM_SIM_VARIANT=$v timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-ransac --no-pipeline --no-extensions --no-regimes > gpurun_out/abl_$v.json 2>gpurun_out/abl_$v.err || { tail -3 gpurun_out/abl_$v.err; exit 1; }
  python3 -c "
import json;d=json.load(open('gpurun_out/abl_$v.json'));print('variant $v', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['config']['matches_rank0'])"
done
# and the shipped instance on a threshold nothing can pass (no visit to the candidate path at all)
timeout -k 10 300 python bench.py --threshold 1.5 --steps 3 --warmup 1 --no-cpu-baseline --no-ransac --no-pipeline --no-extensions --no-regimes > gpurun_out/abl_none.json 2>gpurun_out/abl_none.err || exit 1
python3 -c "
import json;d=json.load(open('gpurun_out/abl_none.json'));print('no candidates', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['config']['matches_rank0'])"
