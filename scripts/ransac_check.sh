# RANSAC tests (with a larger randomised sweep) and the RANSAC bench line, twice
SFM_FUZZ_CASES=${SFM_FUZZ_CASES:-300} timeout -k 10 600 python -u -m pytest tests/test_ransac_gpu.py tests/test_ransac_fuzz_gpu.py tests/test_pipeline_gpu.py tests/test_golden_gpu.py tests/test_cpp_frontend_gpu.py -x -q --timeout 180 --timeout-method thread > gpurun_out/ransac_tests.txt 2>&1
st=$?; echo "status $st"; tail -4 gpurun_out/ransac_tests.txt; [ $st -eq 0 ] || exit 1
for k in 1 2; do
  timeout -k 10 200 python bench.py --images 20 --steps 5 --warmup 2 --ransac-pairs 64 --no-cpu-baseline --no-extensions > gpurun_out/bench_rs.json 2>gpurun_out/bench_rs.err || exit 1
  python3 -c "
import json;d=json.load(open('gpurun_out/bench_rs.json'));print(d['ransac']['value'], d['ransac']['score_kernel_ms_per_step'], d['ransac']['mean_best_inliers'], d['pipeline']['ms'])"
done
