set -e
cd /root/repo
L=gpurun_out/defer_sweep.log
echo "== tests" >> $L
timeout -k 10 600 python -u -m pytest tests/test_ransac_gpu.py tests/test_ransac_fuzz_gpu.py tests/test_pipeline_gpu.py tests/test_zz_no_alloc.py -x -q -m gpu --timeout 300 --timeout-method thread 2>&1 | tail -15 >> $L
for r in 0.8 0.5 0.15; do
for cfg in "1 0 4" "1 512 4" "1 512 8" "1 1024 4" "1 1024 2"; do
set -- $cfg
echo "== defer $1 first $2 growth $3 adaptive $r" >> $L
SFM_DEBUG_DEFER=$1 SFM_DEBUG_GROWTH=$3 timeout -k 10 120 python scripts/probe_ransac_regimes.py $r 256 1 1 2 $2 2>&1 | grep -v percentiles | tail -1 >> $L
done
for cfg in "1 512" "1 1024"; do
set -- $cfg
echo "== defer $1 first $2 all rows $r" >> $L
SFM_DEBUG_DEFER=$1 timeout -k 10 120 python scripts/probe_ransac_regimes.py $r 256 1 0 2 $2 2>&1 | grep -v percentiles | tail -1 >> $L
done
done
