cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/trace_pipeline
timeout -s KILL 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace_pipeline -- python3 bench.py --images 20 --steps 1 --warmup 0 --no-cpu-baseline --no-ransac --no-extensions > gpurun_out/trace_pipeline.json 2>gpurun_out/trace_pipeline.err
echo "status $?"
