# A/B of two builds of the library on one box: old, new, old, new (headline matcher only)
for v in old new old new; do
  cp sfm_amd/lib/ab/$v.so sfm_amd/lib/libsfm_amd.so
  timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-ransac --no-pipeline --no-extensions > gpurun_out/ab_$v.json 2>gpurun_out/ab_$v.err || exit 1
  python3 -c "
import json;d=json.load(open('gpurun_out/ab_$v.json'));print('$v', d['value'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['config']['matches_rank0'])"
done
