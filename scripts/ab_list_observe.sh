# The headline bench with the match lists allocated for the worst case / sized as the call goes, alternating (one box)
for v in 0 1 0 1 0 1; do
  SFM_MATCH_LIST_OBSERVE=$v timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-ransac --no-pipeline --no-extensions --no-regimes > gpurun_out/lo_$v.json 2>gpurun_out/lo_$v.err || { tail -5 gpurun_out/lo_$v.err; exit 1; }
  python3 -c "
import json;d=json.load(open('gpurun_out/lo_$v.json'));print('list_observe $v', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['config']['matches_rank0'], d['config']['device_bytes_rank0']['matcher_work_buffers_and_lists'])"
done
