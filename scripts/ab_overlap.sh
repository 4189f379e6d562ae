# Matcher tests, then the headline bench with the chunk overlap off / on / off / on (one box)
timeout -k 10 600 python -u -m pytest tests/test_match_gpu.py tests/test_match_fuzz_gpu.py tests/test_match_ratio_gpu.py tests/test_golden_gpu.py tests/test_pipeline_gpu.py tests/test_extensions_fuzz_gpu.py -x -q --timeout 180 --timeout-method thread > gpurun_out/overlap_tests.txt 2>&1
st=$?; echo "tests status $st"; tail -3 gpurun_out/overlap_tests.txt; [ $st -eq 0 ] || exit 1
for v in 0 1 0 1; do
  SFM_MATCH_OVERLAP=$v timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-ransac --no-pipeline --no-extensions > gpurun_out/ov_$v.json 2>gpurun_out/ov_$v.err || { tail -5 gpurun_out/ov_$v.err; exit 1; }
  python3 -c "
import json;d=json.load(open('gpurun_out/ov_$v.json'));print('overlap $v', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['config']['matches_rank0'])"
done
