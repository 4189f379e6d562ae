# A/B of two builds of the library on one box (sfm_amd/lib/ab/old.so, new.so): matcher tests on the new one,
# then the headline bench old, new, old, new
cp sfm_amd/lib/ab/new.so sfm_amd/lib/libsfm_amd.so
timeout -k 10 600 python -u -m pytest tests/test_match_gpu.py tests/test_match_fuzz_gpu.py tests/test_match_ratio_gpu.py tests/test_golden_gpu.py tests/test_pipeline_gpu.py -x -q --timeout 180 --timeout-method thread > gpurun_out/ab_tests.txt 2>&1
st=$?; echo "tests status $st"; tail -3 gpurun_out/ab_tests.txt; [ $st -eq 0 ] || exit 1
for v in old new old new; do
  cp sfm_amd/lib/ab/$v.so sfm_amd/lib/libsfm_amd.so
  timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-ransac --no-pipeline --no-extensions --no-regimes > gpurun_out/ab_$v.json 2>gpurun_out/ab_$v.err || exit 1
  python3 -c "
import json;d=json.load(open('gpurun_out/ab_$v.json'));print('$v', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['config']['matches_rank0'])"
done
