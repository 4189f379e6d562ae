# Kernel statistics of the RANSAC stress shape only (bench.py without the matcher's secondary legs).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/trace_ransac -- python3 bench.py --steps 2 --warmup 1 --images 8 --no-cpu-baseline --no-pipeline --no-extensions > gpurun_out/trace_ransac.json 2>gpurun_out/trace_ransac.err
echo "status $?"
python3 scripts/trace_summary.py gpurun_out/trace_ransac 12
