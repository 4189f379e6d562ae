cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -s KILL 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d gpurun_out/trace_headline -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-ransac --no-pipeline --no-extensions > gpurun_out/trace_headline.json 2>gpurun_out/trace_headline.err
echo "status $?"
